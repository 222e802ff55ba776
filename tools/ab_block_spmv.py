#!/usr/bin/env python3
"""Block (elasticity) SpMV: time and achieved bandwidth of K_el x for the coupled config, variants interleaved."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads
from glimslib_amd._backend import Handle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 99
w = workloads.config_c5(n)
h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
t = w.tables
h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
h.set_options(dt=w.dt)
h.setup(True)
st = h.stats()
d = 3
b_alg = (d * d * 8 + 4) * st['nnz'] + (4 + 2 * d * 8) * st['n_rows']      # block CSR: 9 fp64 + one column per block
x = np.random.default_rng(0).standard_normal(h.n_nodes * d)
variants = [(v, dict(GLIMS_BLK_VARIANT=v)) for v in (sys.argv[2:] or ["0", "1"])]
res = {k: [] for k, _ in variants}
y0 = None
for rnd in range(7):
    for name, env in variants:
        os.environ.update(env)
        y, ms = h.apply(3, x, reps=20)
        if y0 is None:
            y0 = y
        assert np.array_equal(y, y0), "variants disagree"
        if rnd > 0:
            res[name].append(ms / 20 * 1e3)
for name, v in res.items():
    v = np.array(v)
    print("variant %-4s median %8.1f us  min %8.1f us  -> %6.0f GB/s algorithmic (%.3f GB per launch)" %
          (name, np.median(v), v.min(), b_alg / np.median(v) / 1e3, b_alg / 1e9))
