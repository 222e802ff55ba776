#!/usr/bin/env python3
"""Large 2-D check (triangles): device vs C oracle on an n x n rectangle, plus step time and SpMV rate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd.mesh import RectangleMesh
from glimslib_amd._backend import Handle
from glimslib_amd import workloads
from oracle.c_port import COracle
from oracle.glims_oracle import rel_l2

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
mesh = RectangleMesh((-50.0, -50.0), (50.0, 50.0), n, n)
lab = np.where(mesh.cell_midpoints()[:, 0] > 0.0, 1, 2).astype(np.int32)
D, rho = np.array([0.0, 0.1, 0.02]), np.array([0.0, 0.1, 0.05])
c0 = np.exp(-0.02 * (mesh.points ** 2).sum(1))
h = Handle(mesh.points, mesh.cells, lab)
h.set_materials(D, rho, [0, 0, 0], [1, 1, 1], [0.3, 0.3, 0.3])
h.set_options(dt=1.0)
h.setup(False)
h.set_state(c0)
assert h.step(3) == 0
c = h.get_state(want_u=False)[0]
co = COracle(mesh.points, mesh.cells, D[lab], rho[lab], 1.0)
ref = co.step(c0, 3, rtol=1e-11, cg_rtol=1e-4)
print("2-D %d x %d: %d nodes, %d triangles; 3 steps rel-L2 vs C oracle %.2e" % (n, n, mesh.num_vertices(), mesh.num_cells(), rel_l2(c, ref)))
h.reset_stats()
assert h.step(20) == 0
st = h.stats()
x = np.random.default_rng(0).standard_normal(h.n_nodes)
_, ms = h.apply(0, x, reps=30)
b = workloads.b_spmv_bytes(st['nnz'], st['n_rows'])
print("%.2f ms/step (%.1f Newton, %.1f PCG its per step); SpMV %.1f us = %.0f GB/s algorithmic; 16-bit codes on %d of %d entries"
      % (st['ms_steps'] / 20, st['newton_its'] / 20, st['cg_its'] / 20, ms / 30 * 1e3, b / (ms / 30 * 1e-3) / 1e9,
         st['nnz_idx16'], st['nnz_padded']))
