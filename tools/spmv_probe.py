#!/usr/bin/env python3
"""Why does the SpMV time differ between a fresh handle and after time stepping?  Probe order / state effects."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads
from glimslib_amd._backend import Handle
w = workloads.by_name('c4')
h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
t = w.tables
h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
h.set_options(dt=w.dt)
h.setup(False)
h.set_state(w.c0)
rng = np.random.default_rng(0)
x = rng.standard_normal(h.n_nodes)
def t_apply(which, x, tag):
    h.apply(which, x, reps=5)
    v = [h.apply(which, x, reps=30)[1] / 30 * 1e3 for _ in range(4)]
    print("%-44s %s us" % (tag, " ".join("%.0f" % a for a in v)), flush=True)
t_apply(0, x, "fresh: A (=S copy), random x")
t_apply(1, x, "fresh: S, random x")
t_apply(2, x, "fresh: M, random x")
h.step(6)
t_apply(0, x, "after 6 steps: A(c), random x")
t_apply(1, x, "after 6 steps: S, random x")
t_apply(2, x, "after 6 steps: M, random x")
time.sleep(3)
t_apply(0, x, "after 3 s idle: A(c), random x")
c, _ = h.get_state(want_u=False)
t_apply(0, c, "A(c), x = c (smooth, mostly ~0)")
t_apply(0, np.ones(h.n_nodes), "A(c), x = 1")
