#!/usr/bin/env python3
"""A/B of the assembled SELL-64 product A(c) x against the matrix-free one ((S + 2 dt N(c)) x rebuilt from the (row, cell)
incidence lists, glims_apply which = 7) -- the measurement SURVEY.md 7.1 step 5 asks for.   tools/ab_matfree.py [c3|c4]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads
from glimslib_amd._backend import Handle

w = workloads.by_name(sys.argv[1] if len(sys.argv) > 1 else "c4")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
t = w.tables
h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
h.set_options(dt=w.dt)
h.setup(False)
h.set_state(w.c0)
assert h.step(3) == 0
c = h.get_state(want_u=False)[0]
h.set_state(c)
h.rd_residual(c, c)                       # assembles A(c) for exactly this state
x = np.random.default_rng(0).standard_normal(h.n_nodes)
ya, _ = h.apply(0, x)
ym, _ = h.apply(7, x)
st = h.stats()
print("%s: %d rows, nnz %d, incidences %d; |y_mf - y_asm| / |y_asm| = %.2e" %
      (w.name, st['n_rows'], st['nnz'], st['n_corners'], np.linalg.norm(ym - ya) / np.linalg.norm(ya)))
for rnd in range(3):                      # interleaved
    _, ta = h.apply(0, x, reps=reps)
    _, tm = h.apply(7, x, reps=reps)
    print("round %d: assembled %.1f us / launch, matrix-free %.1f us / launch, ratio %.2f" %
          (rnd, 1e3 * ta / reps, 1e3 * tm / reps, tm / ta))
alg_a = 8 * st['nnz_padded'] + 2 * st['nnz_padded'] + 16 * st['n_rows']
alg_m = 12 * st['n_corners'] + 8 * st['nnz_padded'] + 4 * st['nnz_padded'] + 24 * st['n_rows']
print("bytes streamed per launch by design: assembled %.3f GB (values + 16-bit codes + x, y), matrix-free %.3f GB "
      "(incidence records + S + int32 columns + c, x, y)" % (alg_a / 1e9, alg_m / 1e9))
h.close()
