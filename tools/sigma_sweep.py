#!/usr/bin/env python3
"""SELL-C-sigma window sweep on an unstructured (Delaunay) mesh: padding vs gather locality."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads
from glimslib_amd._backend import Handle
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
w = workloads.config_unstructured(npts)
x = np.random.default_rng(0).standard_normal(w.mesh.num_vertices())
t = w.tables
for sigma in [int(v) for v in os.environ.get('SIGMAS', '64,128,256,512,1024,4096').split(',')]:
    os.environ["GLIMS_SIGMA"] = str(sigma)
    h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    h.set_options(dt=w.dt)
    h.setup(False)
    st = h.stats()
    h.apply(0, x, reps=5)
    ms = min(h.apply(0, x, reps=30)[1] / 30 for _ in range(3))
    b = workloads.b_spmv_bytes(st['nnz'], st['n_rows'])
    h.set_state(w.c0)
    h.step(3)
    h.reset_stats()
    t0 = time.perf_counter(); h.step(15); el = (time.perf_counter() - t0) / 15
    s2 = h.stats()
    print("sigma %5d: padding +%.1f%%  spmv %.1f us = %.0f GB/s  | step %.2f ms (cg %.1f/step)" %
          (sigma, 100.0 * (st['nnz_padded'] / st['nnz'] - 1), ms * 1e3, b / ms / 1e6, el * 1e3, s2['cg_its'] / 15.0), flush=True)
    h.close()
