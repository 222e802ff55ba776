#!/usr/bin/env python3
"""Elasticity multigrid on a general (Delaunay) mesh: one mesh, one handle, a sweep over the hierarchy's options -- PCG
iterations from a zero guess and ms per solve for each setting.

    python tools/sweep_mg_general.py [n_points=300000]
    env: HFACS="1.6,2,2.4"  SMOOTHS="3,4"  RATIOS="0"  COARSE="216"  JITTER=0.3 (bounded-quality mesh: jittered lattice)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads, _backend  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
    t0 = time.perf_counter()
    jit = os.environ.get("JITTER")
    w = workloads.config_unstructured(n, mechanics=True, jitter=float(jit) if jit else None)
    print("mesh: %d nodes, %d cells (%.1f s)" % (w.mesh.num_vertices(), w.mesh.num_cells(), time.perf_counter() - t0),
          flush=True)
    h = _backend.Handle(w.mesh.points, w.mesh.cells, w.cell_label)
    t = w.tables
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    dofs = (w.dirichlet_nodes[:, None] * 3 + np.arange(3)).ravel()
    h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    hfacs = [float(x) for x in os.environ.get("HFACS", "1.6,2,2.4").split(",")]
    smooths = [int(x) for x in os.environ.get("SMOOTHS", "3,4").split(",")]
    ratios = [float(x) for x in os.environ.get("RATIOS", "0").split(",")]
    rows = []
    for hf in hfacs:
        for sm in smooths:
            for ra in ratios:
                h.set_options(dt=w.dt, mech_history=0, mg_h_factor=hf, mg_smooth=sm, mg_cheb_ratio=ra,
                              mg_coarse_nodes=int(os.environ.get("COARSE", "216")))
                h.setup(True)
                h.set_state(w.c0)
                st = h.solve_mechanics()            # builds the hierarchy
                h.reset_stats()
                reps = 3
                t0 = time.perf_counter()
                for _ in range(reps):
                    h.set_state(w.c0)               # zero displacement again: every solve starts from the zero guess
                    st |= h.solve_mechanics()
                el = (time.perf_counter() - t0) / reps
                s = h.stats()
                row = dict(h_factor=hf, smooth=sm, ratio=ra, status=int(st), its=s['mech_cg_its'] / reps,
                           ms_per_solve=1e3 * el, levels=int(s['mg_levels']), complexity=s['mg_complexity'],
                           setup_ms=s['ms_mg_setup'], res=s['last_mech_res'])
                rows.append(row)
                print(json.dumps(row), flush=True)
    h.close()


if __name__ == "__main__":
    main()
