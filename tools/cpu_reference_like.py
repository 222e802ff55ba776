#!/usr/bin/env python3
"""
Times the reference-LIKE CPU path (baseline (2a)): Newton with a sparse direct solve of the monolithic
(d+1)N system per iteration -- what DOLFIN's default `linear_solver='default'` (LU) does under
simulation_tumor_growth.py:126-130 -- restated with scipy (SuperLU, one core).  FEniCS itself is not installed.
Prints DoF-updates/s (all (d+1) N unknowns count) for C1 (coupled, 2-D) and a reduced C2 (RD only, 3-D).

usage: tools/cpu_reference_like.py [n_c2=24]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads
from oracle.glims_oracle import OracleTumorGrowth

n2 = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for w, steps in ((workloads.config_c1(), 10), (workloads.config_c2(n2), 3)):
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
    print("%-40s %8d unknowns, %2d steps, %.2f s  ->  %.3e DoF-updates/s (1 core, SuperLU)" %
          (w.name, unknowns, steps, el, unknowns * steps / el), flush=True)
