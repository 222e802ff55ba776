#!/usr/bin/env python3
"""
bench.py -- DoF-updates/s of the implicit reaction-diffusion time step on a 3-D brain-extent mesh, plus the
HBM roofline fraction of the diffusion SpMV and a CPU baseline, as ONE JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c3|c2] [--n CELLS_PER_EDGE]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one backward-Euler step of the concentration equation (Newton + Jacobi-PCG) over the whole mesh.
N > 1: the SAME mesh is partitioned over N GPUs (strong scaling), halo exchange + all-reduce over RCCL.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(budget_s=15.0):
    """
    CPU baseline ("port"): the C/OpenMP restatement oracle/glims_oracle_c.c (Newton + Jacobi-PCG on CSR, fp64, same
    tolerances as the device run) on this job's CPU share of the box (16 OpenMP threads unless GLIMS_ORACLE_THREADS
    says otherwise; the box reports more cores than one job may use), on a bounded sample of the same workload: the
    brain-extent box at n=99 (config C3, 1 000 000 DoF), as many implicit steps as fit in ~budget_s seconds.
    FEniCS itself is not installed here, hence kind = "port", not "reference".
    """
    from oracle.c_port import COracle
    from glimslib_amd import workloads
    w = workloads.config_c3()
    t0 = time.perf_counter()
    o = COracle(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.dt)
    t_setup = time.perf_counter() - t0
    c = o.step(w.c0, 1)                                             # warm-up step (not timed)
    steps = 0
    t0 = time.perf_counter()
    while True:
        c = o.step(c, 1)
        steps += 1
        el = time.perf_counter() - t0
        if (steps >= 3 and el > budget_s) or steps >= 200:
            break
    n = w.mesh.num_vertices()
    st = o.stats()
    return {"value": n * steps / el, "unit": "DoF-updates/s", "cores": COracle.threads(), "kind": "port",
            "sample": "oracle/glims_oracle_c.c (C + OpenMP, CSR Newton/Jacobi-PCG, fp64) on the same brain-extent box "
                      "at n=99 (%d DoF): %d implicit steps in %.1f s on %d threads (setup %.1f s excluded; "
                      "%.1f Newton, %.1f PCG iterations per step)" %
                      (n, steps, el, COracle.threads(), t_setup, st['newton_its'] / (steps + 1.0),
                       st['cg_its'] / (steps + 1.0))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("GLIMS_BENCH_WORKLOAD", "c4"))
    ap.add_argument("--size", "--n", dest="n", type=int, default=0,
                    help="cells per edge (overrides the workload's size); use --size under torchrun, which claims --n*")
    ap.add_argument("--spmv-reps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extrapolate", type=int, default=0)
    ap.add_argument("--cg-rtol", type=float, default=None, help="override glims_options.cg_rtol (tuning runs only)")
    ap.add_argument("--check-every", type=int, default=None)
    ap.add_argument("--newton-rtol", type=float, default=None)
    ap.add_argument("--fp32-jacobian", type=int, default=0,
                    help="1: GLIMS_FLAG_FP32_JACOBIAN (study runs only; the line then says dtype f64/f32-jacobian)")
    ap.add_argument("--warm-start", type=int, default=None, help="override GLIMS_FLAG_WARM_START (tuning runs only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            log("bench.py: --gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
            sys.exit(2)
        args.gpus = world

    import torch
    if not torch.cuda.is_available():
        log("bench.py: no GPU visible -- the HIP backend has no CPU fallback")
        sys.exit(3)
    # GLIMS_FORCE_DEVICE: rehearsal of the N>1 code path on a box with fewer GPUs than ranks (all ranks share one
    # device, torch side on gloo); never set by the driver.
    forced = os.environ.get("GLIMS_FORCE_DEVICE")
    if forced is not None:
        local_rank = int(forced)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if forced is not None:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="cpu:gloo,cuda:nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    # HIP events around the Krylov SpMV launches of the timed steps (read by the library at glims_create)
    os.environ.setdefault("GLIMS_TIME_SPMV", "1")
    from glimslib_amd import workloads
    from glimslib_amd._backend import Handle, GLIMS_OK, FLAG_EXTRAPOLATE_GUESS
    from glimslib_amd.partition import partition_mesh

    t0 = time.perf_counter()
    w = workloads.by_name(args.workload, args.n or None)
    n_global = w.mesh.num_vertices()
    if rank == 0:
        log("[bench] workload %s: %d nodes, %d cells (mesh built in %.1f s)" %
            (w.name, n_global, w.mesh.num_cells(), time.perf_counter() - t0))

    t0 = time.perf_counter()
    if world > 1:
        part = partition_mesh(w.mesh.points, w.mesh.cells, world, rank)
        h = Handle(part.points, part.cells, w.cell_label[part.cell_ids], n_own=part.n_own, device=local_rank)
        from glimslib_amd.parallel import broadcast_unique_id, HostStagedTransport
        if forced is not None:      # rehearsal: RCCL refuses several ranks per device, stage the halos through gloo
            transport = HostStagedTransport(dist)
            h.set_transport(rank, world, transport.halo_cb, transport.allreduce_cb)
        else:
            h.comm_init(rank, world, broadcast_unique_id(dist, rank))
        h.set_halo(part.peer_rank, part.send_ptr, part.send_idx, part.recv_count)
        from glimslib_amd.parallel import setup_node_mailbox
        mailbox = setup_node_mailbox(h, dist, rank)
        if rank == 0:
            log("[bench] scalar all-reduce: %s" % ("node mailbox (shared host memory)" if mailbox else "RCCL"))
        c0 = w.c0[part.global_ids]
    else:
        part = None
        h = Handle(w.mesh.points, w.mesh.cells, w.cell_label, device=local_rank)
        c0 = w.c0
    t = w.tables
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    extra = {}
    if args.cg_rtol is not None:
        extra["cg_rtol"] = args.cg_rtol
    if args.check_every is not None:
        extra["check_every"] = args.check_every
    if args.newton_rtol is not None:
        extra["newton_rtol"] = args.newton_rtol
    flags = FLAG_EXTRAPOLATE_GUESS if args.extrapolate else h.options.flags
    if args.warm_start is not None:
        flags = (flags | 2) if args.warm_start else (flags & ~2)
    if args.fp32_jacobian:
        flags |= 4
    h.set_options(dt=w.dt, flags=flags, **extra)
    # coupled configs (C5): the displacement is solved after EVERY step, as the reference's monolithic solve does; the
    # unknown count is then (d + 1) per node.  (The simulation classes solve it lazily)
    coupled = bool(w.mechanics)
    dim = w.mesh.points.shape[1] if w.mesh is not None else 3
    if coupled and w.dirichlet_nodes is not None:
        if part is None:
            nodes = np.asarray(w.dirichlet_nodes, dtype=np.int64)
        else:
            g2l = np.full(n_global, -1, dtype=np.int64)
            g2l[part.global_ids[:part.n_own]] = np.arange(part.n_own)
            nodes = g2l[np.asarray(w.dirichlet_nodes, dtype=np.int64)]
            nodes = nodes[nodes >= 0]
        dofs = (nodes[:, None] * dim + np.arange(dim)).ravel()
        h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    h.setup(with_mechanics=coupled)
    h.set_state(c0)
    st0 = h.stats()
    if rank == 0:
        log("[bench] setup %.1f s; rows/rank %d, nnz %d (padded %d, +%.1f%%), corners %d" %
            (time.perf_counter() - t0, st0['n_rows'], st0['nnz'], st0['nnz_padded'],
             100.0 * (st0['nnz_padded'] / st0['nnz'] - 1.0), st0['n_corners']))
    w.mesh = None   # free host memory

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up, then EXACTLY K timed steps ---------------------------------------------------------
    def advance(k):
        if not coupled:
            return h.step(k)
        st_ = GLIMS_OK
        for _ in range(k):
            st_ |= h.step(1)
            st_ |= h.solve_mechanics()
        return st_

    status = advance(args.warmup) if args.warmup > 0 else GLIMS_OK
    h.reset_stats()
    steps_before = h.stats()['steps']
    barrier()
    t0 = time.perf_counter()
    status |= advance(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if forced is not None else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st = h.stats()
    steps_done = int(st['steps'] - steps_before)      # < K only if the solver gave up (status != 0)
    if status != GLIMS_OK:
        log("[bench] WARNING: solver status %d after %d of %d steps; throughput counts completed steps only" %
            (status, steps_done, args.steps))

    # ---- roofline of the dominant kernel: SELL-64 SpMV with the RD Jacobian A(c) ------------------------
    # algorithmic bytes per launch = 12*nnz + 20*rows of THIS rank's operator;
    # duration = HIP events on the library's own stream around `reps` back-to-back launches (glims_apply).
    # (a) inside the timed region: HIP events around every Krylov SpMV launch of the steps (GLIMS_TIME_SPMV, single
    #     GPU); (b) after it: `spmv_reps` back-to-back launches of the same operator without the fused dot product.
    x = np.random.default_rng(0).standard_normal(h.n_nodes)
    h.apply(0, x, reps=5)
    _, ms = h.apply(0, x, reps=args.spmv_reps)
    t_spmv = ms * 1e-3 / args.spmv_reps
    b_alg = workloads.b_spmv_bytes(st['nnz'], st['n_rows'])
    t_isolated = t_spmv
    in_step = st.get('n_spmv_steps', 0) > 0
    if in_step:
        t_spmv = st['ms_spmv_steps'] * 1e-3 / st['n_spmv_steps']
    achieved = b_alg / t_spmv / 1e9
    # HBM-side bytes per launch from the PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3 runs,
    # calibrated on kernels with exactly known byte counts): profiles/r01_pmc_c4.json.  Only reported when this
    # run's operator is the one those passes measured.
    traffic = None
    try:
        pmc = json.load(open(os.path.join(HERE, "profiles", "r01_pmc_c4.json")))
        if world == 1 and pmc["n_rows"] == st['n_rows'] and pmc["nnz"] == st['nnz']:
            key = [k for k in pmc["kernels"] if k.startswith("k_spmv<1" if in_step else "k_spmv<0")][0]
            traffic = pmc["kernels"][key]["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": ("k_spmv<1, 4, 1, 1>" if in_step else "k_spmv<0, 4, 1, 1>") + " (SELL-64, fp64 values, "
                "columns streamed as 16-bit window codes; algorithmic bytes still count 4-byte CSR columns)",
                "algorithmic_bytes_per_launch": b_alg, "avg_launch_us": t_spmv * 1e6,
                "launches_timed": int(st['n_spmv_steps']) if in_step else args.spmv_reps,
                "timed": "inside the timed steps (k_spmv<1,..>, fused dot product)" if in_step
                         else "back-to-back launches after the timed steps (k_spmv<0,..>)",
                "isolated_launch_us": t_isolated * 1e6}

    # practical HBM ceiling of THIS device next to the nominal peak: streaming scale kernel y = 2 x over
    # 1 GiB, read + write bytes over the HIP-event time of the fastest of 10 launches (torch is only the allocator / launcher here;
    # its memcpy path goes through the copy engines and is much slower than a kernel)
    try:
        src = torch.empty(1 << 27, dtype=torch.float64, device="cuda")
        dst = torch.empty_like(src)
        src.fill_(1.0)
        torch.mul(src, 2.0, out=dst)
        best = None
        for _ in range(10):                     # one event pair per launch, fastest launch counts
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            torch.mul(src, 2.0, out=dst)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e-3
            best = t if best is None else min(best, t)
        roofline["stream_ceiling"] = 2.0 * src.numel() * 8 / best / 1e9
        del src, dst
    except Exception:   # noqa: BLE001 -- informational only
        roofline["stream_ceiling"] = None

    if rank == 0:
        out = {
            "metric": "DoF-updates/s (implicit RD timestep) on 3D brain mesh; 1/2/4/8 GPU + %HBM roofline",
            "value": n_global * (dim + 1 if coupled else 1) * steps_done / elapsed,
            "unit": "DoF-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(1, steps_done),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if not args.fp32_jacobian else "f64 (Jacobian stored f32 inside the Krylov solves)",
            "data": "synthetic",
            "config": {"workload": w.name, "dofs": n_global * (dim + 1 if coupled else 1), "dt": w.dt,
                       "mech_cg_its_per_step": st['mech_cg_its'] / max(1, steps_done) if coupled else None,
                       "partition": "morton-node x%d" % world if world > 1 else "single GPU",
                       "newton_its_per_step": st['newton_its'] / max(1, steps_done),
                       "cg_its_per_step": st['cg_its'] / max(1, steps_done),
                       "assemblies_per_step": st['rd_assemblies'] / max(1, steps_done),
                       "device_ms_per_step": st['ms_steps'] / max(1, steps_done),
                       "steps_completed": steps_done,
                       "solver_status": int(status)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    h.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
