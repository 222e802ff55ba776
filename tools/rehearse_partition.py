#!/usr/bin/env python3
"""
Rehearsal of a partitioned run with MANY ranks on one GPU: the ranks are threads of this process (parallel.ThreadedTransport),
each with its own handle, sub-mesh and halo plan -- the product path except for RCCL.  Reports, per rank, rows / ghosts / peers,
the elasticity multigrid's work-box fraction and first-grid operator bytes, and compares fields and iteration counts with the
single-rank run of the same problem.

    python tools/rehearse_partition.py bl:1000000 8 [rcb|morton] [rd_steps=3] [mech=1]
    python tools/rehearse_partition.py c5:99 8

Reference counterpart: DOLFIN's ParMETIS / SCOTCH partition + PETSc under mpirun (README.md:142-183).
"""
import os
import sys
import time

for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads, _backend as B                         # noqa: E402
from glimslib_amd.parallel import run_threaded_ranks                      # noqa: E402
from glimslib_amd.partition import node_owners, build_local_part          # noqa: E402


def rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(1e-300, np.linalg.norm(b)))


def h_factor_for(world):
    """First-grid spacing a framed partitioned run picks (glims_options.mg_h_factor = 0): 2 / 3 / 4 h for <= 2 / <= 6 / more."""
    return 2.0 if world <= 2 else 3.0 if world <= 6 else 4.0


def run_single(w, rd_steps, mech, world_for_h=1, **opts):
    h = B.Handle(w.mesh.points, w.mesh.cells, w.cell_label)
    t = w.tables
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    h.set_options(dt=w.dt, mech_history=0, mg_h_factor=h_factor_for(world_for_h) if mech else 0.0, **opts)
    d = w.mesh.points.shape[1]
    if mech:
        dofs = (np.asarray(w.dirichlet_nodes)[:, None] * d + np.arange(d)).ravel()
        h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    h.setup(bool(mech))
    h.set_state(w.c0)
    st = h.step(rd_steps)
    if mech:
        st |= h.solve_mechanics()
    c, u = h.get_state(want_u=bool(mech))
    s = h.stats()
    h.close()
    return st, c, u, s


def run_partitioned(w, world, rd_steps, mech, method='rcb', box_min_nodes=None, **opts):
    """Returns (status, c, u, per-rank stats dicts).  Fields are assembled from the ranks' owned values."""
    pts, cells = w.mesh.points, w.mesh.cells
    d = pts.shape[1]
    owner = node_owners(pts, world, cells, method=method)
    parts = [build_local_part(pts, cells, owner, r, world) for r in range(world)]
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    if box_min_nodes is not None:
        os.environ["GLIMS_MG_BOX_MIN_NODES"] = str(box_min_nodes)

    def rank_body(rank, tr):
        part = parts[rank]
        h = B.Handle(part.points, part.cells, w.cell_label[part.cell_ids], n_own=part.n_own, device=0)
        h.set_transport(rank, world, tr.halo_cb, tr.allreduce_cb)
        h.set_halo(part.peer_rank, part.send_ptr, part.send_idx, part.recv_count)
        h.set_mg_frame(lo, hi)
        t = w.tables
        h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
        h.set_options(dt=w.dt, mech_history=0, **opts)
        if mech:
            g2l = np.full(len(pts), -1, dtype=np.int64)
            g2l[part.global_ids[:part.n_own]] = np.arange(part.n_own)
            nodes = g2l[np.asarray(w.dirichlet_nodes)]
            nodes = nodes[nodes >= 0]
            dofs = (nodes[:, None] * d + np.arange(d)).ravel()
            h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
        h.setup(bool(mech))
        h.set_state(w.c0[part.global_ids])
        st = h.step(rd_steps)
        if mech:
            st |= h.solve_mechanics()
        c, u = h.get_state(want_u=bool(mech))
        s = h.stats()
        s.update(rank=rank, ghosts=int(part.n_local - part.n_own), peers=int(len(part.peer_rank)), status=int(st))
        h.close()
        if tr.failed is not None:
            raise tr.failed
        return c[:part.n_own], (u.reshape(-1, d)[:part.n_own] if mech else None), s

    res = run_threaded_ranks(world, rank_body)
    c = np.full(len(pts), np.nan)
    u = np.full((len(pts), d), np.nan) if mech else None
    status = 0
    for r, (cr, ur, s) in enumerate(res):
        own = parts[r].global_ids[:parts[r].n_own]
        c[own] = cr
        if mech:
            u[own] = ur
        status |= s['status']
    return status, c, (u.reshape(-1) if mech else None), [s for _, _, s in res]


def main():
    spec = sys.argv[1] if len(sys.argv) > 1 else "bl:200000"
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    method = sys.argv[3] if len(sys.argv) > 3 else "rcb"
    rd_steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    mech = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    name, _, size = spec.partition(":")
    t0 = time.perf_counter()
    if name in ("bl", "brain_like"):
        w = workloads.config_brain_like(int(size or 200000), mechanics=bool(mech), isolate=True)
    elif name == "c5" or mech:
        w = workloads.config_c5(int(size or 99)) if name in ("c5", "c3", "c4") else workloads.by_name(name, int(size) if size else None)
    else:
        w = workloads.by_name(name, int(size) if size else None)
    print("%s: %d nodes, %d cells (%.1f s)" % (w.name, w.mesh.num_vertices(), w.mesh.num_cells(), time.perf_counter() - t0), flush=True)
    t0 = time.perf_counter()
    st1, c1, u1, s1 = run_single(w, rd_steps, mech, world_for_h=world)
    print("single rank: status %d, Newton %d, Krylov passes %d (Chebyshev %d), elasticity PCG %d  (%.1f s)" %
          (st1, s1['newton_its'], s1['cg_its'], s1['cheb_its'], s1['mech_cg_its'], time.perf_counter() - t0), flush=True)
    t0 = time.perf_counter()
    st, c, u, ss = run_partitioned(w, world, rd_steps, mech, method)
    print("%d ranks (%s): status %d  (%.1f s)" % (world, method, st, time.perf_counter() - t0))
    for s in ss:
        print("  rank %d: rows %8d  ghosts %7d  peers %d  halo %.2f MB/exchange  work box %.0f %% of the first grid, operator %.1f MB, "
              "multigrid complexity %.2f; Newton %d, Krylov %d (Chebyshev %d, fallbacks %d), elasticity PCG %d" %
              (s['rank'], s['n_rows'], s['ghosts'], s['peers'], s['halo_bytes'] / max(1, s['halo_exchanges']) / 1e6,
               100.0 * s['mg_box_fraction'], s['mg_grid1_bytes'] / 1e6, s['mg_complexity'], s['newton_its'], s['cg_its'],
               s['cheb_its'], s['cheb_fallbacks'], s['mech_cg_its']))
    print("partitioned vs single rank: concentration %.2e%s" %
          (rel_l2(c, c1), "" if not mech else ", displacement %.2e" % rel_l2(u, u1)))
    print("largest work box: %.0f %% of the first grid; rows max / mean %.3f" %
          (100.0 * max(s['mg_box_fraction'] for s in ss),
           max(s['n_rows'] for s in ss) / (sum(s['n_rows'] for s in ss) / float(world))))


if __name__ == "__main__":
    main()
