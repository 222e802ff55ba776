#!/usr/bin/env python3
"""Jacobi- against multigrid-preconditioned RD solves on a stiff step (BASELINE config C2's problem: unit cube, D = rho =
0.1, dt = 1; or a 2-D square): ms per step, Krylov iterations per Newton solve, agreement of the fields.

    python tools/run_rd_precond.py 46 99 215          # 3-D, cells per edge
    DIM=2 python tools/run_rd_precond.py 1000
    DEG=2 ... (Chebyshev degree of the RD hierarchy)   RATIO=10 (mg_cheb_ratio)   STEPS=10
    MESH=delaunay DSCALE=100 python tools/run_rd_precond.py 200000     # general mesh, diffusivities x DSCALE
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from glimslib_amd import workloads, _backend  # noqa: E402
from glimslib_amd.mesh import RectangleMesh  # noqa: E402


def problem(dim, n):
    if os.environ.get("MESH") in ("delaunay", "bl"):   # n random points (or the brain-like mesh of ~n nodes) in the brain-extent box, D scaled by DSCALE
        w = workloads.config_unstructured(n) if os.environ["MESH"] == "delaunay" else workloads.config_brain_like(n)
        f = float(os.environ.get("DSCALE", "100"))
        w.tables = dict(w.tables)
        w.tables['D'] = [f * d for d in w.tables['D']]
        return w
    if dim == 3:
        return workloads.config_c2(n)
    mesh = RectangleMesh((0.0, 0.0), (1.0, 1.0), n, n)
    label = np.ones(mesh.num_cells(), dtype=np.int32)
    tables = dict(D=[0.0, 0.1], rho=[0.0, 0.1], gamma=[0.0, 0.1], E=[1.0, 3e-3], nu=[0.3, 0.45])
    c0 = np.exp(-1.0 * ((mesh.points - 0.5) ** 2).sum(axis=1))
    return workloads.Workload("unit square n=%d" % n, mesh, label, tables, c0, 1.0, 20, False)


def main():
    dim = int(os.environ.get("DIM", "3"))
    steps = int(os.environ.get("STEPS", "10"))
    deg = int(os.environ.get("DEG", "0"))
    ratio = float(os.environ.get("RATIO", "0"))
    out = []
    for n in [int(a) for a in sys.argv[1:]] or [46]:
        w = problem(dim, n)
        t0 = time.perf_counter()
        h = _backend.Handle(w.mesh.points, w.mesh.cells, w.cell_label)
        t = w.tables
        h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
        row = {"dim": dim, "n": n, "dofs": w.mesh.num_vertices(), "create_s": time.perf_counter() - t0}
        fields = {}
        only = os.environ.get("ONLY")          # e.g. ONLY=multigrid under a profiler
        for name, pre in (("jacobi", _backend.RD_PRECOND_JACOBI), ("multigrid", _backend.RD_PRECOND_MULTIGRID),
                          ("auto", _backend.RD_PRECOND_AUTO)):
            if only and name != only:
                continue
            h.set_options(dt=w.dt, rd_precond=pre, rd_mg_smooth=deg, mg_cheb_ratio=ratio)
            h.setup(False)
            h.set_state(w.c0)
            st = h.step(2)                      # warm-up (builds the hierarchy)
            h.reset_stats()
            t0 = time.perf_counter()
            st |= h.step(steps)
            el = time.perf_counter() - t0
            s = h.stats()
            fields[name] = h.get_state(want_u=False)[0]
            row[name] = {"status": int(st), "ms_per_step": 1e3 * el / steps, "device_ms_per_step": s['ms_steps'] / steps,
                         "newton_per_step": s['newton_its'] / steps, "pcg_per_solve": s['cg_its'] / max(1, s['newton_its']),
                         "pcg_per_step": s['cg_its'] / steps, "used": int(s['rd_precond_used']),
                         "q": s['rd_stiffness_ratio'], "levels": int(s['rd_mg_levels']),
                         "complexity": s['rd_mg_complexity'], "setup_ms": s['ms_rd_mg_setup'],
                         "dof_updates_per_s": w.mesh.num_vertices() * steps / el}
        h.close()
        print(json.dumps(row), flush=True)
        if only:
            continue
        row["rel_l2_multigrid_vs_jacobi"] = float(np.linalg.norm(fields["multigrid"] - fields["jacobi"]) /
                                                  np.linalg.norm(fields["jacobi"]))
        out.append(row)
    for r in out:
        print("dim %d n %4d (%9d DoF), q %8.1f: Jacobi %8.2f ms/step (%6.1f its/solve) | multigrid %8.2f ms/step "
              "(%5.1f its/solve, %d levels, set-up %.0f ms) | auto -> %s %8.2f ms/step | fields %.1e apart" %
              (r["dim"], r["n"], r["dofs"], r["jacobi"]["q"], r["jacobi"]["ms_per_step"], r["jacobi"]["pcg_per_solve"],
               r["multigrid"]["ms_per_step"], r["multigrid"]["pcg_per_solve"], r["multigrid"]["levels"],
               r["multigrid"]["setup_ms"], {1: "Jacobi", 2: "multigrid"}[r["auto"]["used"]], r["auto"]["ms_per_step"],
               r["rel_l2_multigrid_vs_jacobi"]), file=sys.stderr)


if __name__ == "__main__":
    main()
