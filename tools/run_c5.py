#!/usr/bin/env python3
"""Config C5 (coupled model on the brain-extent box): time the displacement solve at recorded steps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads
from glimslib_amd._backend import Handle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 99
w = workloads.config_c5(n)
t0 = time.perf_counter()
h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
t = w.tables
h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
h.set_options(dt=w.dt, mech_rtol=float(os.environ.get("MECH_RTOL", "1e-8")))
dofs = (w.dirichlet_nodes[:, None] * 3 + np.arange(3)).ravel()
h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
h.setup(True)
h.set_state(w.c0)
print("setup %.1f s, %d nodes" % (time.perf_counter() - t0, h.n_nodes), flush=True)
for rec in range(3):
    t0 = time.perf_counter(); st = h.step(10); t1 = time.perf_counter()
    sm = h.solve_mechanics(); t2 = time.perf_counter()
    s = h.stats()
    print("record %d: 10 RD steps %.3f s (status %d), mechanics %.3f s (status %d), mech its so far %d, res %.2e" %
          (rec, t1 - t0, st, t2 - t1, sm, s['mech_cg_its'], s['last_mech_res']), flush=True)
c, u = h.get_state()
print("max |u| = %.3e, max c = %.3f" % (np.abs(u).max(), c.max()))
