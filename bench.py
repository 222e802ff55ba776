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

# numpy's BLAS threads busy-wait after a call, one per VISIBLE core (256 on the GPU boxes), while the job's CPU share is 16:
# the cgroup is then throttled for tens of milliseconds, which shows as 2-3x slower time steps at 1 M rows
# (profiles/r05_blas_threads_throttle.txt).  This program needs BLAS for a few norms only: one thread.
for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernels_sha16():
    """Hash of the library's sources: figures measured by another run (profiles/*.json) are only quoted for the same kernels."""
    import glob
    import hashlib
    hs = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(HERE, "glimslib_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(HERE, "glimslib_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(HERE, "glimslib_amd", "csrc", "*.cpp"))):
        hs.update(open(f, "rb").read())
    return hs.hexdigest()[:16]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (torch.distributed.run), relay its one
    JSON line and return its exit code.  Nothing in this process has touched the GPU (torch is not even imported yet)."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] --gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd[1:9])))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return r.returncode


def _time_c_oracle(w, budget_s, min_steps, warm=True):
    """Time the C/OpenMP oracle on workload `w`: (DoF-updates/s, steps, seconds, setup seconds, stats)."""
    from oracle.c_port import COracle
    t0 = time.perf_counter()
    o = COracle(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.dt)
    t_setup = time.perf_counter() - t0
    c = w.c0
    if warm:
        c = o.step(c, 1)                                            # warm-up step (not timed)
    steps = 0
    t0 = time.perf_counter()
    while True:
        c = o.step(c, 1)
        steps += 1
        el = time.perf_counter() - t0
        if (steps >= min_steps and el > budget_s) or steps >= 200:
            break
    st = o.stats()
    n = w.mesh.num_vertices()
    o.close()
    return n * steps / el, steps, el, t_setup, st, n


def cpu_baseline(w_headline=None, budget_s=12.0):
    """
    CPU baseline ("port"): the C/OpenMP restatement oracle/glims_oracle_c.c (Newton + Jacobi-PCG on CSR, fp64, same
    tolerances as the device run) on this job's CPU share of the box (16 OpenMP threads unless GLIMS_ORACLE_THREADS
    says otherwise; the box reports more cores than one job may use).  Two bounded samples, each one warm-up step and
    then as many implicit steps as fit in ~budget_s seconds: (1) the workload of this very line (`w_headline`, e.g. C4
    at its full 10 M DoF) -> `value`; (2) the same brain-extent box at n = 99 (config C3, 1 M DoF) -> `c3_sample`.  FEniCS itself is not installed here, hence kind = "port".
    """
    from oracle.c_port import COracle
    from glimslib_amd import workloads
    out = {"unit": "DoF-updates/s", "cores": COracle.threads(), "kind": "port"}
    v3, steps, el, t_setup, st, n = _time_c_oracle(workloads.config_c3(), budget_s, 3)
    c3 = {"value": v3, "workload": "C3 brain-extent box n=99 (%d DoF)" % n, "steps": steps, "seconds": el,
          "setup_seconds": t_setup, "newton_its_per_step": st['newton_its'] / (steps + 1.0),
          "cg_its_per_step": st['cg_its'] / (steps + 1.0)}
    if w_headline is not None and w_headline.mesh is not None and w_headline.mesh.num_vertices() != n:
        big = False
        vh, hs, hel, hsetup, hst, hn = _time_c_oracle(w_headline, budget_s, 3)
        out.update({"value": vh, "same_workload_as_value": True,
                    "sample": "oracle/glims_oracle_c.c (C + OpenMP, CSR Newton/Jacobi-PCG, fp64) on THIS line's workload "
                              "(%s, %d DoF): %d implicit steps in %.1f s on %d threads (setup %.1f s excluded; %.1f Newton, "
                              "%.1f PCG iterations per step%s)" %
                              (w_headline.name, hn, hs, hel, COracle.threads(), hsetup, hst['newton_its'] / float(hs),
                               hst['cg_its'] / float(hs), "; first steps of the run, no warm-up step" if big else ""),
                    "c3_sample": c3})
    else:
        out.update({"value": v3, "same_workload_as_value": w_headline is not None,
                    "sample": "oracle/glims_oracle_c.c (C + OpenMP, CSR Newton/Jacobi-PCG, fp64) on the brain-extent box "
                              "at n=99 (%d DoF): %d implicit steps in %.1f s on %d threads (setup %.1f s excluded; "
                              "%.1f Newton, %.1f PCG iterations per step)" %
                              (n, steps, el, COracle.threads(), t_setup, c3['newton_its_per_step'], c3['cg_its_per_step'])})
    return out


def cpu_reference_like(n_c2=18):
    """
    The reference-LIKE CPU path (baseline 2a): Newton with a sparse direct solve (scipy SuperLU, one
    core) of the whole system per iteration -- what DOLFIN's default linear_solver='default' = LU does under
    simulation_tumor_growth.py:126-130 -- through the numpy oracle, on BASELINE config C1 (coupled, monolithic (d+1)N
    system, its 10 steps) and on config C2 reduced to n_c2 cells per edge (RD block, 2 steps; at C2's real size one 3-D
    factorisation no longer fits a bench run).  FEniCS itself is not installed; bounded to ~15 s.
    """
    from glimslib_amd import workloads
    from oracle.glims_oracle import OracleTumorGrowth
    out = {"unit": "DoF-updates/s", "cores": 1, "kind": "port",
           "what": "Newton + sparse LU (SuperLU) of the whole system per iteration, oracle/glims_oracle.py"}
    for key, w, steps in (("c1", workloads.config_c1(), 10), ("c2_reduced", workloads.config_c2(n_c2), 2)):
        per = {k: w.per_cell(k) for k in ('D', 'rho', 'gamma', 'E', 'nu')}
        dim = w.mesh.points.shape[1]
        kw = {}
        if w.dirichlet_nodes is not None:
            dofs = (w.dirichlet_nodes[:, None] * dim + np.arange(dim)).ravel()
            kw['dirichlet_u'] = (dofs, np.zeros(len(dofs)))
        o = OracleTumorGrowth(w.mesh.points, w.mesh.cells, per['D'], per['rho'], per['gamma'], per['E'], per['nu'], w.dt, **kw)
        n = w.mesh.num_vertices()
        t0 = time.perf_counter()
        if w.mechanics:
            o.run(w.c0, steps * w.dt, mechanics=True, monolithic=True)
            unknowns = (dim + 1) * n
        else:
            o.run(w.c0, steps * w.dt, mechanics=False, linear='lu')
            unknowns = n
        el = time.perf_counter() - t0
        out[key] = {"workload": w.name, "unknowns": unknowns, "steps": steps, "seconds": el, "value": unknowns * steps / el}
    return out


def alt_c2(Handle, device, steps=20, warmup=2):
    """BASELINE config C2 (unit cube n=46, D = rho = 0.1, dt = 1: a STIFF step, dt D / h^2 = 211) under the driver's
    clock: ms per step and Krylov iterations with the preconditioner `auto` picks, and with Jacobi for comparison."""
    from glimslib_amd import workloads, _backend
    w = workloads.config_c2()
    h = Handle(w.mesh.points, w.mesh.cells, w.cell_label, device=device)
    t = w.tables
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    out = {"workload": w.name, "dofs": w.mesh.num_vertices(), "steps": steps, "warmup": warmup}
    for name, pre in (("auto", _backend.RD_PRECOND_AUTO), ("jacobi", _backend.RD_PRECOND_JACOBI)):
        h.set_options(dt=w.dt, rd_precond=pre)
        h.setup(with_mechanics=False)
        h.set_state(w.c0)
        st = h.step(warmup)
        h.reset_stats()
        t0 = time.perf_counter()
        st |= h.step(steps)
        el = time.perf_counter() - t0
        s = h.stats()
        out[name] = {"ms_per_step": 1e3 * el / steps, "value": w.mesh.num_vertices() * steps / el, "solver_status": int(st),
                     "newton_its_per_step": s['newton_its'] / steps, "pcg_its_per_step": s['cg_its'] / steps,
                     "pcg_its_per_newton_solve": s['cg_its'] / max(1, s['newton_its']),
                     "preconditioner": {1: "jacobi", 2: "multigrid"}.get(int(s['rd_precond_used']), "?"),
                     "stiffness_ratio_q": s['rd_stiffness_ratio'], "mg_levels": int(s['rd_mg_levels']),
                     "mg_setup_ms": s['ms_rd_mg_setup']}
    out.update({k: out["auto"][k] for k in ("ms_per_step", "value", "solver_status", "pcg_its_per_step", "preconditioner")})
    h.close()
    return out


def alt_c5(Handle, device, steps=10, warmup=10):
    """BASELINE config C5 (coupled model, c + u = 4 DoF per node) on the brain-extent box at n = 99 (1 M nodes) under the
    driver's clock: one RD step + one elasticity solve per step, as the reference's monolithic solve does (10 warm-up
    steps: the elasticity solver's initial guess is a fit over the last 8 solves, a run of the reference's length -- 50
    steps -- spends its time in that steady state); then a short pass with HIP events around the two dominant kernels of the
    elasticity solve for their roofline entries."""
    from glimslib_amd import workloads
    w = workloads.config_c5()
    n = w.mesh.num_vertices()
    h = Handle(w.mesh.points, w.mesh.cells, w.cell_label, device=device)
    t = w.tables
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    h.set_options(dt=w.dt)
    nodes = np.asarray(w.dirichlet_nodes, dtype=np.int64)
    dofs = (nodes[:, None] * 3 + np.arange(3)).ravel()
    h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    h.setup(with_mechanics=True)
    h.set_state(w.c0)
    st = h.solve_mechanics()                      # builds the hierarchy (set-up reported, not timed)
    for _ in range(warmup):
        st |= h.step(1) | h.solve_mechanics()
    h.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        st |= h.step(1) | h.solve_mechanics()
    el = time.perf_counter() - t0
    s = h.stats()
    out = {"workload": w.name, "dofs": 4 * n, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * el / steps,
           "value": 4 * n * steps / el, "solver_status": int(st),
           "mech_pcg_its_per_solve": s['mech_cg_its'] / max(1, s['mech_solves']), "mech_ms_per_solve": s['ms_mech'] / steps,
           "rd_device_ms_per_step": s['ms_steps'] / steps, "mg_levels": int(s['mg_levels']),
           "mg_complexity": s['mg_complexity'], "mg_setup_ms": s['ms_mg_setup'], "last_mech_true_residual": s['last_mech_res']}
    # per-kernel figures: eager launches with event pairs, 3 more coupled steps
    h.set_options(time_kernels=3)
    h.reset_stats()
    for _ in range(3):
        st |= h.step(1) | h.solve_mechanics()
    k = h.stats()
    h.set_options(time_kernels=0)
    kernels = []
    ent, rows = k['nnz_padded'], k['n_rows']
    for name, what, alg, ms, cnt, med in (
            ("k_mg_fine<3, 1, 2, _Float16, 1, float>", "level-0 pass of the V-cycle: half-precision copy of the scaled K_el "
             "with 16-bit column codes + the Chebyshev step's vector update (algorithmic bytes: 20 per stored 3x3 block + "
             "~100 per node of vectors and inverse diagonal blocks)", 20 * ent + 100 * rows,
             k['ms_mgfine_mech'], k['n_mgfine_mech'], k['us_mgfine_median']),
            ("k_spmv_block2<3, 1, 2, double>", "w = K_el u of the Krylov iteration, fp64 3x3 blocks, fused w.u partials "
             "(algorithmic bytes: 76 per stored block + 48 per node)", 76 * ent + 48 * rows,
             k['ms_spmvb_mech'], k['n_spmvb_mech'], k['us_spmvb_median'])):
        if cnt > 0:
            mean_us = 1e3 * ms / cnt
            kernels.append({"name": name, "does": what, "algorithmic_bytes_per_launch": alg, "median_us": med,
                            "mean_us": mean_us, "launches_per_solve": cnt / 3.0,
                            "achieved_GBps": alg / (mean_us * 1e-6) / 1e9,
                            "frac": alg / (mean_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                            "timed": "HIP events, eager launches, 3 coupled steps after the timed ones"})
    out["kernels"] = kernels
    out["solver_status"] = int(st)
    h.close()
    return out


def cheb_bytes(nnz, n_rows):
    """Algorithmic bytes of one pass of the dot-free Krylov iteration (k_cheb): the CSR SpMV's 12 nnz + 4 rows + 8 rows of
    gathered iterate, plus the recurrence's vector work: right-hand side, Dinv, direction read + written, new iterate
    (40 B per row) = 12 nnz + 52 rows."""
    return 12 * int(nnz) + 52 * int(n_rows)


def load_pmc(st, candidates):
    """Committed PMC summary (tools/pmc_summary.py) of the same operator, or (None, None)."""
    for cand in candidates:
        try:
            q = json.load(open(os.path.join(HERE, "profiles", cand)))
            if q["n_rows"] == st['n_rows'] and q["nnz"] == st['nnz']:
                return q, "profiles/" + cand
        except Exception:   # noqa: BLE001
            continue
    return None, None


def pmc_lookup(pmc, prefix):
    if pmc is None:
        return None
    keys = [k for k in pmc["kernels"] if k.startswith(prefix)]
    if not keys:
        return None
    if prefix in ("k_rd_assemble", "k_rd_quad"):      # one launch per slice class: a sweep is the sum of them
        return sum(pmc["kernels"][k]["hbm_bytes_per_launch"] for k in keys)
    return pmc["kernels"][keys[0]]["hbm_bytes_per_launch"]


def kernel_list(s, k, steps, step_ms, k_steps, unr, pmc=None):
    """Per-kernel roofline entries: s = stats of the timed steps (time_kernels = 1: operator passes), k = stats of the short
    pass with event pairs around every hot kernel (time_kernels = 2), or None."""
    out = []

    def entry(name, what, alg, ms, cnt, med, n_steps, ms_step, where, pmc_prefix):
        if cnt <= 0:
            return
        mean_us = 1e3 * ms / cnt
        tb = pmc_lookup(pmc, pmc_prefix)
        out.append({"name": name, "does": what, "algorithmic_bytes_per_launch": alg, "median_us": med,
                    "mean_us": mean_us, "launches_per_step": cnt / float(n_steps),
                    "achieved_GBps": alg / (mean_us * 1e-6) / 1e9, "frac": alg / (mean_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "share_of_step": ms / (ms_step * n_steps), "traffic_bytes_per_launch": tb,
                    "achieved_real_GBps": None if tb is None else tb / (mean_us * 1e-6) / 1e9, "timed": where})

    where = "HIP events inside the %d timed steps" % steps
    entry("k_cheb<%d, NT, 1, double>" % unr, "one pass of the dot-free Krylov iteration: t = b - A y (SELL-64, 16-bit column codes) "
          "with the Chebyshev recurrence in the epilogue (algorithmic bytes: 12 nnz + 52 rows)",
          cheb_bytes(s['nnz'], s['n_rows']), s['ms_cheb_steps'], s['n_cheb_steps'], s['us_cheb_median'], steps, step_ms,
          where, "k_cheb")
    entry("k_spmv<1, %d, NT, 1, double>" % unr, "y = A(c) x, SELL-64 with 16-bit column codes, fused w.u partials (PCG solves of "
          "the learning steps; algorithmic bytes: CSR with 4-byte columns, 12 nnz + 20 rows)",
          12 * s['nnz'] + 20 * s['n_rows'], s['ms_spmv_steps'], s['n_spmv_steps'], s['us_spmv_median'], steps, step_ms, where,
          "k_spmv<1")
    if k is not None:
        kms = k['ms_steps'] / k_steps
        where = "HIP events in a separate pass of %d steps right after the timed ones" % k_steps
        entry("k_rd_assemble_s<4, CAP, RB, 1, double> (one launch per slice class)", "Jacobian + Newton residual(s) in one "
              "sweep over the (row, cell) incidences (algorithmic bytes: 12 per incidence + 20 per stored entry + 32 per row)",
              12 * s['n_corners'] + 20 * s['nnz'] + 32 * s['n_rows'], k['ms_sweep_steps'], k['n_sweep_steps'],
              k['us_sweep_median'], k_steps, kms, where, "k_rd_assemble")
        entry("k_rd_quad_s<4, CAP, RB, 1> (one launch per slice class)", "Newton residual from the quadratic structure "
              "(algorithmic bytes: 8 per incidence + 4 per stored entry + 24 per row)",
              8 * s['n_corners'] + 4 * s['nnz'] + 24 * s['n_rows'], k['ms_quad_steps'], k['n_quad_steps'],
              k['us_quad_median'], k_steps, kms, where, "k_rd_quad")
        entry("k_cg_update<1>", "PCG recurrence + vector update (96 B per row; learning steps only)", 96 * s['n_rows'],
              k['ms_update_steps'], k['n_update_steps'], k['us_update_median'], k_steps, kms, where, "k_cg_update<1>")
    return out


def solver_counts(s, steps):
    return {"newton_its_per_step": s['newton_its'] / steps, "krylov_passes_per_step": s['cg_its'] / steps,
            "chebyshev_solves_per_step": s['cheb_solves'] / steps, "chebyshev_passes_per_step": s['cheb_its'] / steps,
            "chebyshev_fallbacks": int(s['cheb_fallbacks']), "pcg_learning_solves": int(s['cheb_learn_solves']),
            "spectral_interval": [s['cheb_lmin'], s['cheb_lmax']],
            "assemblies_per_step": s['rd_assemblies'] / steps,
            "quadratic_residual_updates_per_step": s['rd_quad_updates'] / steps,
            "stream_policy": "non-temporal" if s['stream_nontemporal'] else "cached",
            "krylov_working_set_bytes": int(s['krylov_working_set'])}


def alt_rank_sized(Handle, device, headline_ns_per_row, steps=20, warmup=5, n=107, octant=False):
    """The per-rank share of an 8-GPU run of config C4 on ONE GPU: the brain-extent box at n = 107 (1.26 M rows = C4 / 8), same
    parameters -- the regime that decides the 8-GPU number (small-problem efficiency, not communication): ms per step, the
    per-row rate relative to the headline's, and the per-kernel list.  octant: instead the central octant of C4 at C4's own mesh
    width (workloads.config_c4_octant) -- the same rows per rank AND the same problem (spectrum, iteration counts) as a rank of
    the partitioned headline run; the box at n = 107 is a coarser mesh with more Krylov passes per step."""
    from glimslib_amd import workloads
    w = workloads.config_c4_octant(n) if octant else workloads.by_name("c4", n)
    rows = w.mesh.num_vertices()
    h = Handle(w.mesh.points, w.mesh.cells, w.cell_label, device=device)
    t = w.tables
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    # (no event pairs in the timed steps: at this size the ~17 pairs per step around the operator passes cost 10-15 % of the step
    #  -- 1.22 against 1.05 ms; the per-kernel timings come from two passes after them)
    h.set_options(dt=w.dt, time_kernels=0)
    h.setup(with_mechanics=False)
    h.set_state(w.c0)
    st = h.step(warmup)
    h.reset_stats()
    t0 = time.perf_counter()
    st |= h.step(steps)
    el = time.perf_counter() - t0
    s0 = h.stats()
    h.set_options(time_kernels=1)
    h.reset_stats()
    st |= h.step(steps)
    s = h.stats()
    k_steps = 5
    h.set_options(time_kernels=2)
    h.reset_stats()
    st |= h.step(k_steps)
    k = h.stats()
    ns_row = 1e9 * el / steps / rows
    out = {"workload": w.name + ("" if octant else " (1/8 of config C4's rows)"), "dofs": rows, "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * el / steps, "value": rows * steps / el, "solver_status": int(st),
           "ns_per_row_and_step": ns_row, "headline_ns_per_row_and_step": headline_ns_per_row,
           "per_row_rate_relative_to_headline": (headline_ns_per_row / ns_row) if headline_ns_per_row else None,
           "projected_8gpu_speedup_without_communication": (8.0 * headline_ns_per_row / ns_row) if headline_ns_per_row else None}
    out.update(solver_counts(s0, steps))
    out["ms_per_step_with_event_pairs"] = s['ms_steps'] / steps
    pmc, pmc_file = load_pmc(s, ("r05_c_pmc_c4_107.json", "r05_a_pmc_c4_107.json"))
    out["kernels"] = kernel_list(s, k, steps, s['ms_steps'] / steps, k_steps, 8, pmc)
    out["traffic_source"] = None if pmc is None else "lookup, not measured in this run: " + pmc_file
    h.close()
    return out


def alt_unstructured(Handle, device, steps=20, warmup=5, n_points=1000000):
    """The brain-like unstructured mesh (workloads.config_brain_like: ~1 M nodes, quality-controlled Delaunay tetrahedra,
    curved two-tissue interface, config C3's parameters) under the driver's clock -- the stand-in for the CGAL atlas
    meshes the reference's 3-D cases load (test_case_comparison_3D_atlas.py:84-121): ms per step, iterations, and the
    per-kernel roofline list (Krylov SpMV timed inside the timed steps, the other kernels in a short pass after them)."""
    from glimslib_amd import workloads
    tm = time.perf_counter()
    w = workloads.config_brain_like(n_points, isolate=True)
    t_mesh = time.perf_counter() - tm
    n = w.mesh.num_vertices()
    ts = time.perf_counter()
    h = Handle(w.mesh.points, w.mesh.cells, w.cell_label, device=device)
    t = w.tables
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    h.set_options(dt=w.dt, time_kernels=0)   # (timed without event pairs, like alt_rank_sized)
    h.setup(with_mechanics=False)
    h.set_state(w.c0)
    t_setup = time.perf_counter() - ts
    st = h.step(warmup)
    h.reset_stats()
    t0 = time.perf_counter()
    st |= h.step(steps)
    el = time.perf_counter() - t0
    s = h.stats()
    out = {"workload": w.name, "dofs": n, "cells": w.mesh.num_cells(), "steps": steps, "warmup": warmup,
           "ms_per_step": 1e3 * el / steps, "value": n * steps / el, "solver_status": int(st),
           "pcg_its_per_step": s['cg_its'] / steps,
           "preconditioner": {1: "jacobi", 2: "multigrid"}.get(int(s['rd_precond_used']), "?"),
           "nnz": int(s['nnz']), "nnz_padded": int(s['nnz_padded']), "n_corners": int(s['n_corners']),
           "mesh_seconds": t_mesh, "setup_seconds": t_setup}
    out.update(solver_counts(s, steps))
    h.set_options(time_kernels=1)
    h.reset_stats()
    st |= h.step(steps)
    s = h.stats()
    out["ms_per_step_with_event_pairs"] = s['ms_steps'] / steps
    k_steps = 5
    h.set_options(time_kernels=2)
    h.reset_stats()
    st |= h.step(k_steps)
    k = h.stats()
    pmc, pmc_file = load_pmc(s, ("r05_c_pmc_bl.json", "r05_a_pmc_bl.json"))
    kernels = kernel_list(s, k, steps, s['ms_steps'] / steps, k_steps, 16, pmc)
    out["traffic_source"] = None if pmc is None else "lookup, not measured in this run: " + pmc_file
    out["kernels"] = kernels
    out["solver_status"] = int(st)
    h.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("GLIMS_BENCH_WORKLOAD", "c4"))
    ap.add_argument("--size", "--n", dest="n", type=int, default=0,
                    help="cells per edge (overrides the workload's size); use --size under torchrun, which claims --n*")
    ap.add_argument("--spmv-reps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the informational fp32-Jacobian-storage pass")
    ap.add_argument("--extrapolate", type=int, default=0)
    ap.add_argument("--cg-rtol", type=float, default=None, help="override glims_options.cg_rtol (tuning runs only)")
    ap.add_argument("--check-every", type=int, default=None)
    ap.add_argument("--newton-rtol", type=float, default=None)
    ap.add_argument("--fp32-jacobian", type=int, default=0,
                    help="1: GLIMS_FLAG_FP32_JACOBIAN (study runs only; the line then says dtype f64/f32-jacobian)")
    ap.add_argument("--warm-start", type=int, default=None, help="override GLIMS_FLAG_WARM_START (tuning runs only)")
    ap.add_argument("--stream-policy", type=int, default=None, help="glims_options.stream_policy: 1 non-temporal, 2 cached (A/B runs)")
    ap.add_argument("--rd-linear", type=int, default=None, help="glims_options.rd_linear: 1 PCG, 2 Chebyshev (A/B runs)")
    ap.add_argument("--fixed-forcing", type=int, default=0,
                    help="1: GLIMS_FLAG_FIXED_FORCING, cg_rtol for every linear solve (A/B against the default forcing that "
                         "follows the quadratic remainder from a step's second solve on)")
    ap.add_argument("--full-newton", type=int, default=0,
                    help="1: GLIMS_FLAG_FULL_NEWTON, a sweep after every linear solve (A/B against the default, which takes "
                         "the residual after a solve from the quadratic structure)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    # stdout carries exactly ONE line, the JSON result of rank 0.  Libraries below us write there too (gloo announces
    # "[Gloo] Rank r is connected to n peer ranks" on stdout when a process group forms): everything written to fd 1
    # from here on goes to stderr, the result line is written to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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

    from glimslib_amd import workloads
    from glimslib_amd._backend import Handle, GLIMS_OK, FLAG_EXTRAPOLATE_GUESS
    from glimslib_amd.partition import partition_mesh

    t0 = time.perf_counter()
    # N > 1 on the box configs: every rank builds only its own share of the mesh (same partition, array by array, as
    # cutting the whole mesh: tests/test_partition_box.py)
    lw = workloads.local_by_name(args.workload, args.n or None, world, rank) if world > 1 else None
    if lw is None and world > 1 and args.workload.lower() in ("bl", "brain_like", "brain-like"):
        # the unstructured mesh is built ONCE (rank 0, worker processes of a child interpreter) and handed to the other ranks
        # through the mesh cache in shared memory; every rank then cuts its own part out of it
        cache = os.path.join("/dev/shm", "glims_mesh_%s" % os.environ.get("MASTER_PORT", "0"))
        os.environ["GLIMS_MESH_CACHE"] = cache
        if rank == 0:
            workloads.by_name(args.workload, args.n or None)
        dist.barrier()
    w = lw if lw is not None else workloads.by_name(args.workload, args.n or None)
    if lw is None and world > 1 and "GLIMS_MESH_CACHE" in os.environ and os.environ["GLIMS_MESH_CACHE"].startswith("/dev/shm/glims_mesh_"):
        dist.barrier()
        if rank == 0:
            import shutil
            shutil.rmtree(os.environ["GLIMS_MESH_CACHE"], ignore_errors=True)
    n_global = lw.n_nodes if lw is not None else w.mesh.num_vertices()
    if rank == 0:
        log("[bench] workload %s: %d nodes, %d cells (%s in %.1f s)" %
            (w.name, n_global, lw.n_cells if lw is not None else w.mesh.num_cells(),
             "this rank's share built" if lw is not None else "mesh built", time.perf_counter() - t0))

    t0 = time.perf_counter()
    if world > 1:
        part = lw.part if lw is not None else partition_mesh(w.mesh.points, w.mesh.cells, world, rank)
        labels = lw.cell_label if lw is not None else w.cell_label[part.cell_ids]
        h = Handle(part.points, part.cells, labels, n_own=part.n_own, device=local_rank)
        from glimslib_amd.parallel import broadcast_unique_id, HostStagedTransport
        if forced is not None:      # rehearsal: RCCL refuses several ranks per device, stage the halos through gloo
            transport = HostStagedTransport(dist)
            h.set_transport(rank, world, transport.halo_cb, transport.allreduce_cb)
        else:
            h.comm_init(rank, world, broadcast_unique_id(dist, rank))
        h.set_halo(part.peer_rank, part.send_ptr, part.send_idx, part.recv_count)
        if lw is not None:
            h.set_mg_frame(lw.frame[0], lw.frame[1])
        else:
            h.set_mg_frame(w.mesh.points.min(axis=0), w.mesh.points.max(axis=0))
        from glimslib_amd.parallel import setup_node_mailbox
        mailbox = setup_node_mailbox(h, dist, rank)
        if rank == 0:
            log("[bench] scalar all-reduce: %s" % ("node mailbox (shared host memory)" if mailbox else "RCCL"))
        c0 = lw.c0 if lw is not None else w.c0[part.global_ids]
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
    if args.stream_policy is not None:
        extra["stream_policy"] = args.stream_policy
    if args.rd_linear is not None:
        extra["rd_linear"] = args.rd_linear
    flags = FLAG_EXTRAPOLATE_GUESS if args.extrapolate else h.options.flags
    if args.warm_start is not None:
        flags = (flags | 2) if args.warm_start else (flags & ~2)
    if args.fp32_jacobian:
        flags |= 4
    if args.full_newton:
        flags |= 128
    if args.fixed_forcing:
        flags |= 256
    # HIP events around the Krylov SpMV launches of the timed steps (the dominant kernel's in-step roofline figure)
    h.set_options(dt=w.dt, flags=flags, time_kernels=1, **extra)
    # coupled configs (C5): the displacement is solved after EVERY step, as the reference's monolithic solve does; the
    # unknown count is then (d + 1) per node.  (The simulation classes solve it lazily)
    coupled = bool(w.mechanics)
    dim = 3 if lw is not None else w.mesh.points.shape[1]
    if coupled and lw is not None:
        dofs = (lw.dirichlet_local[:, None] * dim + np.arange(dim)).ravel()
        h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    elif coupled and w.dirichlet_nodes is not None:
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
    if coupled:
        h.solve_mechanics()    # builds the multigrid hierarchy of K_el (one-off set-up, reported in config) before any timed step
    st0 = h.stats()
    if rank == 0:
        log("[bench] setup %.1f s; rows/rank %d, nnz %d (padded %d, +%.1f%%), corners %d" %
            (time.perf_counter() - t0, st0['n_rows'], st0['nnz'], st0['nnz_padded'],
             100.0 * (st0['nnz_padded'] / st0['nnz'] - 1.0), st0['n_corners']))

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
    # the other two hot kernels (assembly sweep, PCG vector update): event pairs around every launch of theirs cost
    # ~0.1 ms per step at 1 M rows, so they are timed in a short pass of their own right after the timed region
    st_k = None
    if world == 1 and status == GLIMS_OK and not coupled:
        k_steps = max(2, min(5, args.steps))
        h.set_options(time_kernels=2)
        h.reset_stats()
        if h.step(k_steps) == GLIMS_OK:
            st_k = h.stats()
            st_k['_steps'] = k_steps
        h.set_options(time_kernels=1)
    if status != GLIMS_OK:
        log("[bench] WARNING: solver status %d after %d of %d steps; throughput counts completed steps only" %
            (status, steps_done, args.steps))

    # ---- roofline of the dominant kernel --------------------------------------------------------------------------------
    # The step's dominant kernel is the operator pass of the Krylov iteration: k_cheb (dot-free Chebyshev iteration, the default
    # wherever the RD solves are Jacobi-preconditioned) or k_spmv<1,..> (PCG: learning steps, multigrid-preconditioned solves,
    # rd_linear = PCG).  Algorithmic bytes per launch = SURVEY 8(d)'s CSR figure 12 nnz + 20 rows for the SpMV, plus the
    # recurrence's 32 B per row of extra vector traffic for k_cheb (cheb_bytes);
    # duration (a) inside the timed region: HIP events attached to every such launch of the steps (glims_options.time_kernels),
    # (b) after it: `spmv_reps` back-to-back launches of the plain SpMV (glims_apply).
    x = np.random.default_rng(0).standard_normal(h.n_nodes)
    h.apply(0, x, reps=5)
    _, ms = h.apply(0, x, reps=args.spmv_reps)
    t_isolated = ms * 1e-3 / args.spmv_reps
    cheb_dominant = st.get('n_cheb_steps', 0) > 0 and st['ms_cheb_steps'] >= st['ms_spmv_steps']
    in_step = cheb_dominant or st.get('n_spmv_steps', 0) > 0
    if cheb_dominant:
        t_op = st['ms_cheb_steps'] * 1e-3 / st['n_cheb_steps']
        b_alg = cheb_bytes(st['nnz'], st['n_rows'])
        n_timed, med_us = int(st['n_cheb_steps']), st['us_cheb_median']
    elif in_step:
        t_op = st['ms_spmv_steps'] * 1e-3 / st['n_spmv_steps']
        b_alg = workloads.b_spmv_bytes(st['nnz'], st['n_rows'])
        n_timed, med_us = int(st['n_spmv_steps']), st['us_spmv_median']
    else:
        t_op = t_isolated
        b_alg = workloads.b_spmv_bytes(st['nnz'], st['n_rows'])
        n_timed, med_us = args.spmv_reps, None
    achieved = b_alg / t_op / 1e9
    # HBM-side bytes per launch from the PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3 runs,
    # calibrated on kernels with exactly known byte counts).  A LOOKUP into the committed summary of those passes, not
    # a measurement of this run: only reported when this run's operator is the one the passes measured, and
    # `traffic_source` names the file.
    pmc, pmc_file = load_pmc(st, ("r05_c_pmc_c4.json", "r05_a_pmc_c4.json")) if world == 1 else (None, None)
    unr = 16 if args.workload.lower() in ("bl", "brain_like", "brain-like", "u", "unstructured") else 8
    traffic = pmc_lookup(pmc, "k_cheb" if cheb_dominant else "k_spmv<1" if in_step else "k_spmv<0")
    steps_n = max(1, steps_done)
    ms_step = 1e3 * elapsed / steps_n
    kernels = kernel_list(st, st_k, steps_n, ms_step, st_k['_steps'] if st_k is not None else 1, unr, pmc) if world == 1 else []
    dom_name = ("k_cheb<%d, %d, 1, double>" % (unr, st['stream_nontemporal'])) if cheb_dominant else \
               ("k_spmv<%d, %d, %d, 1, double>" % (1 if in_step else 0, unr, st['stream_nontemporal']))
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": None if traffic is None else
                "lookup, not measured in this run: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same "
                "operator, tools/pmc_summary.py)" % pmc_file,
                "achieved_real": None if traffic is None else traffic / t_op / 1e9,
                "frac_real": None if traffic is None else traffic / t_op / 1e9 / HBM_PEAK_GBS,
                "kernel": dom_name + (" (one pass of the dot-free Krylov iteration: SELL-64 operator pass, fp64 values, 16-bit "
                                      "column codes, Chebyshev recurrence in the epilogue; algorithmic bytes 12 nnz + 52 rows)"
                                      if cheb_dominant else
                                      " (SELL-64, fp64 values, columns streamed as 16-bit window codes; algorithmic bytes "
                                      "still count 4-byte CSR columns: 12 nnz + 20 rows)"),
                "algorithmic_bytes_per_launch": b_alg, "avg_launch_us": t_op * 1e6,
                "median_launch_us": med_us, "launches_timed": n_timed,
                "timed": "HIP events attached to the launches inside the timed steps" if in_step
                         else "back-to-back launches after the timed steps (k_spmv<0,..>)",
                "isolated_spmv_launch_us": t_isolated * 1e6,
                "kernels": kernels}

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

    # ---- informational: the same K steps with GLIMS_FLAG_FP32_JACOBIAN (A(c) stored / streamed in single precision
    # inside the Krylov solves; Newton residual, sums and every vector stay fp64).  OFF by default and NOT the headline:
    # reported next to it, with the distance between the two concentration fields after the same number of steps, so
    # that the decision about the default can be taken from measured numbers.
    alt = None
    if world == 1 and not coupled and status == GLIMS_OK and not args.fp32_jacobian and not args.no_alt:
        try:
            n_post = st_k['_steps'] if st_k is not None else 0
            c_ref = h.get_state(want_u=False)[0].copy()          # default build after W + K + n_post steps
            h.set_options(flags=flags | 4, time_kernels=0)
            h.setup(with_mechanics=False)
            h.set_state(c0)
            a_status = h.step(args.warmup) if args.warmup > 0 else GLIMS_OK
            h.reset_stats()
            barrier()
            ta = time.perf_counter()
            a_status |= h.step(args.steps)
            barrier()
            ta = time.perf_counter() - ta
            sa = h.stats()
            if n_post:
                a_status |= h.step(n_post)
            c_alt = h.get_state(want_u=False)[0]
            log("[bench] alt fp32 Jacobian storage: sum(c) %r vs default %r, max |diff| %.3e" %
                (float(c_alt.sum()), float(c_ref.sum()), float(np.abs(c_alt - c_ref).max())))
            alt = {"fp32_jacobian_storage": {
                "ms_per_step": 1e3 * ta / args.steps, "value": n_global * args.steps / ta, "solver_status": int(a_status),
                "newton_its_per_step": sa['newton_its'] / args.steps, "cg_its_per_step": sa['cg_its'] / args.steps,
                "rel_l2_vs_default": float(np.linalg.norm(c_alt - c_ref) / np.linalg.norm(c_ref)),
                "compared_after_steps": args.warmup + args.steps + n_post,
                "note": "GLIMS_FLAG_FP32_JACOBIAN, off by default; not the headline value"}}
            del c_ref, c_alt
        except Exception as e:   # noqa: BLE001 -- informational only, never in the way of the line
            log("[bench] alt (fp32 Jacobian storage) pass skipped: %r" % (e,))

    # ---- informational, after the headline region: BASELINE configs C2 (a stiff step -> multigrid-preconditioned RD
    # solves) and C5 (coupled, 1 M nodes) under the same clock, each a few seconds; the headline handle is released first
    if world == 1 and not args.no_alt and args.workload.lower() == "c4" and not args.n:
        h.close()
        head_ns = 1e9 * elapsed / max(1, steps_done) / n_global if status == GLIMS_OK else None
        for key, fn in (("rank_sized", lambda H, d: alt_rank_sized(H, d, head_ns)),
                        ("rank_octant", lambda H, d: alt_rank_sized(H, d, head_ns, octant=True)), ("unstructured", alt_unstructured),
                        ("c2", alt_c2), ("c5_coupled", alt_c5)):
            try:
                ta = time.perf_counter()
                res = fn(Handle, local_rank)
                res["wall_seconds_incl_setup"] = time.perf_counter() - ta
                alt = alt or {}
                alt[key] = res
                log("[bench] alt.%s: %.2f ms/step, status %d (%.1f s incl. mesh and set-up)" %
                    (key, res["ms_per_step"], res["solver_status"], res["wall_seconds_incl_setup"]))
            except Exception as e:   # noqa: BLE001 -- informational only
                log("[bench] alt.%s skipped: %r" % (key, e))

    # ---- partitioned runs: what every rank did (so that the first run on real hardware says where its time went) ---------
    ranks_info = None
    if world > 1:
        mine = {"rank": rank, "rows": int(st['n_rows']), "nnz": int(st['nnz']),
                "ghost_nodes": int(h.n_nodes - st['n_rows']), "peers": int(len(part.peer_rank)),
                "halo_exchanges_per_step": st['halo_exchanges'] / max(1, steps_done),
                "halo_bytes_per_exchange": st['halo_bytes'] / max(1, st['halo_exchanges']),
                "halo_bytes_per_step": st['halo_bytes'] / max(1, steps_done),
                # HIP events on the communication stream (pack -> last receive) and around the compute stream's wait for it;
                # 0 under the host-staged rehearsal transport (GLIMS_FORCE_DEVICE), whose callback is synchronous
                "exchange_us_per_exchange": 1e3 * st['ms_exchange'] / max(1, st['halo_exchanges_timed']),
                "halo_exchanges_timed": int(st['halo_exchanges_timed']),
                "exchange_ms_per_step": st['ms_exchange'] / max(1, steps_done),
                "exchange_exposed_ms_per_step": st['ms_exchange_exposed'] / max(1, steps_done),
                "exchange_hidden_share": (1.0 - st['ms_exchange_exposed'] / st['ms_exchange']) if st['ms_exchange'] > 0 else None,
                "allreduces_per_step": st['allreduces'] / max(1, steps_done),
                "reduce_transport": {0: "none", 1: "node mailbox (inside the reduction kernel)", 2: "ncclAllReduce",
                                     3: "host callback"}.get(int(st['reduce_transport']), "?"),
                "device_ms_per_step": st['ms_steps'] / max(1, steps_done),
                # the Krylov SpMV's launch over the INTERIOR slices (the part that hides the halo exchange), HIP events
                "spmv_interior_us_in_step": (1e3 * st['ms_spmv_steps'] / st['n_spmv_steps']) if st['n_spmv_steps'] > 0 else None,
                "cheb_interior_us_in_step": (1e3 * st['ms_cheb_steps'] / st['n_cheb_steps']) if st['n_cheb_steps'] > 0 else None,
                "krylov_passes_per_step": st['cg_its'] / max(1, steps_done),
                "chebyshev_passes_per_step": st['cheb_its'] / max(1, steps_done),
                "chebyshev_fallbacks": int(st['cheb_fallbacks']),
                "mg_complexity": st['mg_complexity'] if coupled else None,
                "mg_first_grid_operator_bytes": int(st['mg_grid1_bytes']) if coupled else None,
                "mech_ms_per_step": st['ms_mech'] / max(1, steps_done) if coupled else None}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_info = gathered

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
            "steps_requested": args.steps,
            "steps_completed": steps_done,
            "solver_status": int(status),
            "config": {"workload": w.name, "dofs": n_global * (dim + 1 if coupled else 1), "dt": w.dt,
                       "mech_cg_its_per_step": st['mech_cg_its'] / max(1, steps_done) if coupled else None,
                       "mech_preconditioner": (("multigrid V-cycle, %d levels, operator complexity %.2f, set-up %.0f ms "
                                                "(outside the timed steps)" % (st['mg_levels'], st['mg_complexity'],
                                                                               st['ms_mg_setup']))
                                               if st['mg_levels'] else "block-Jacobi") if coupled else None,
                       "mech_ms_per_step": st['ms_mech'] / max(1, steps_done) if coupled else None,
                       "partition": "recursive coordinate bisection (equal work) x%d" % world if world > 1 else "single GPU",
                       "newton_its_per_step": st['newton_its'] / max(1, steps_done),
                       "cg_its_per_step": st['cg_its'] / max(1, steps_done),
                       "chebyshev_passes_per_step": st['cheb_its'] / max(1, steps_done),
                       "chebyshev_solves_per_step": st['cheb_solves'] / max(1, steps_done),
                       "chebyshev_fallbacks": int(st['cheb_fallbacks']),
                       "pcg_learning_solves": int(st['cheb_learn_solves']),
                       "spectral_interval_lo": st['cheb_lmin'], "spectral_interval_hi": st['cheb_lmax'],
                       "stream_policy": "non-temporal" if st['stream_nontemporal'] else "cached",
                       "assemblies_per_step": st['rd_assemblies'] / max(1, steps_done),
                       "quadratic_residual_updates_per_step": st['rd_quad_updates'] / max(1, steps_done),
                       "device_ms_per_step": st['ms_steps'] / max(1, steps_done),
                       "steps_completed": steps_done,
                       "solver_status": int(status),
                       },
            "roofline": roofline,
        }
        # The config's FULL length (test_case_comparison_3D_atlas.py:84: 500 steps) as measured on this path by tools/run_long.py:
        # scalar keys read from the committed summary, quoted only while the library's sources are the ones that run measured
        if args.workload.lower() == "c4" and not args.n and not coupled:
            try:
                fr = json.load(open(os.path.join(HERE, "profiles", "long_c4_run.json")))
                if fr.get("kernels_sha16") == kernels_sha16():
                    out["config"].update({"full_run_steps_requested": fr["steps_requested"],
                                          "full_run_steps_completed": fr["steps_completed"],
                                          "full_run_status": fr["final_status"],
                                          "full_run_ms_per_step_mean": fr["ms_per_step_mean"],
                                          "full_run_ms_per_step_min_window": fr["ms_per_step_min_window"],
                                          "full_run_ms_per_step_max_window": fr["ms_per_step_max_window"],
                                          "full_run_source": "profiles/long_c4_run.json (tools/run_long.py, kernels %s)" % fr["kernels_sha16"]})
                else:
                    out["config"]["full_run_source"] = "profiles/long_c4_run.json is of other kernels (%s): not quoted" % fr.get("kernels_sha16")
            except Exception as e:   # noqa: BLE001 -- informational only
                log("[bench] full-run summary not quoted: %r" % (e,))
        if ranks_info is not None:
            out["ranks"] = ranks_info
        if alt is not None:
            out["alt"] = alt
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(None if coupled else w)
            try:
                out["cpu_baseline"]["reference_like"] = cpu_reference_like()
            except Exception as e:   # noqa: BLE001 -- informational only
                log("[bench] reference-like CPU baseline skipped: %r" % (e,))
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    h.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
