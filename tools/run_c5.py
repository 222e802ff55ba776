#!/usr/bin/env python3
"""Config C5 (coupled model on the brain-extent box): elasticity solve after every RD step, per preconditioner.
    python tools/run_c5.py [n] [steps]      env: MESH=bl|jitter PRECOND=mg|bj  SMOOTH=k  RATIO  MIXED=0|1|2  HIST=k  HFAC  COARSE  FP32SM=1  X64=1  MECH_RTOL  NU=0.49  EWM=1e4  EGM=1e4"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads
from glimslib_amd import _backend
from glimslib_amd._backend import Handle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 99
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if os.environ.get("MESH") == "bl":      # the brain-like unstructured mesh of ~n nodes
    w = workloads.config_brain_like(n, mechanics=True)
elif os.environ.get("MESH") == "jitter":   # lattice nodes jittered by 0.3 h, one-piece Delaunay
    w = workloads.config_unstructured(n, mechanics=True, jitter=0.3)
else:
    w = workloads.config_c5(n) if n < 100000 else workloads.config_unstructured(n, mechanics=True)
t0 = time.perf_counter()
h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
t = dict(w.tables)
if os.environ.get("NU"):   # Poisson ratio of every tissue with nu > 0.4 (near-incompressible study)
    t['nu'] = [float(os.environ["NU"]) if v > 0.4 else v for v in t['nu']]
    print("nu table:", t['nu'])
if os.environ.get("EWM"):   # stiffness of the white-matter ellipsoid relative to the table's (coefficient-jump study)
    t['E'] = list(t['E']); t['E'][3] = t['E'][3] * float(os.environ["EWM"])
    print("E table:", t['E'])
if os.environ.get("EGM"):   # the same for the grey-matter shell (touches the clamped hull)
    t['E'] = list(t['E']); t['E'][2] = t['E'][2] * float(os.environ["EGM"])
    print("E table:", t['E'])
h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
pre = _backend.PRECOND_BLOCK_JACOBI if os.environ.get("PRECOND", "mg") == "bj" else _backend.PRECOND_MULTIGRID
h.set_options(dt=w.dt, mech_rtol=float(os.environ.get("MECH_RTOL", "1e-10")), mech_precond=pre,
              mg_smooth=int(os.environ.get("SMOOTH", "3")), mg_cheb_ratio=float(os.environ.get("RATIO", "0")), mech_mixed=int(os.environ.get("MIXED", "1")),
              mech_history=int(os.environ.get("HIST", "6")), mg_h_factor=float(os.environ.get("HFAC", "0")),
              mg_coarse_nodes=int(os.environ.get("COARSE", "216")),
              flags=h.options.flags | (_backend.FLAG_MG_FP32_SMOOTHER if os.environ.get("FP32SM") else 0) |
              (_backend.FLAG_MG_FP64_VECTORS if os.environ.get("X64") else 0))
dofs = (w.dirichlet_nodes[:, None] * 3 + np.arange(3)).ravel()
h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
h.setup(True)
h.set_state(w.c0)
print("setup %.1f s, %d nodes, precond %s" % (time.perf_counter() - t0, h.n_nodes, "mg" if pre else "bj"), flush=True)
prev = 0
for rec in range(steps):
    t0 = time.perf_counter(); st = h.step(1); t1 = time.perf_counter()
    sm = h.solve_mechanics(); t2 = time.perf_counter()
    s = h.stats()
    print("step %d: RD %.2f ms (status %d), mechanics %.2f ms (status %d), %d PCG its, res %.2e%s" %
          (rec, 1e3 * (t1 - t0), st, 1e3 * (t2 - t1), sm, s['mech_cg_its'] - prev, s['last_mech_res'],
           "  [mg set-up %.0f ms, %d levels, complexity %.2f]" % (s['ms_mg_setup'], s['mg_levels'], s['mg_complexity'])
           if rec == 0 and pre else ""), flush=True)
    prev = s['mech_cg_its']
c, u = h.get_state()
print("max |u| = %.3e, max c = %.3f" % (np.abs(u).max(), c.max()))
