#!/usr/bin/env python3
"""Config C5 through the public API: coupled model, 1 M nodes, every step recorded; eager vs device-resident results."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import fenics_local as fenics, workloads
from glimslib_amd.simulation import TumorGrowthBrain

n = int(sys.argv[1]) if len(sys.argv) > 1 else 99
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w = workloads.config_c5(n)
f = w.mesh.facets()
bmask = f['exterior']                                     # boolean facet mask = the whole hull
for lazy in (False, True):
    sim = TumorGrowthBrain(w.mesh)
    sim.setup_global_parameters(subdomains=w.cell_label, domain_names={1: 'CSF', 3: 'WM', 2: 'GM', 4: 'Ventricles'},
                                boundaries={'boundary_all': bmask},
                                dirichlet_bcs={'clamped_0': {'bc_value': fenics.Constant((0.0, 0.0, 0.0)),
                                                             'named_boundary': 'boundary_all', 'subspace_id': 0}})
    sim.setup_model_parameters(iv_expression={0: fenics.Constant((0., 0., 0.)), 1: w.c0}, sim_time=steps,
                               sim_time_step=1, E_GM=3000E-6, E_WM=3000E-6, E_CSF=1000E-6, E_VENT=1000E-6, nu_GM=0.45,
                               nu_WM=0.45, nu_CSF=0.45, nu_VENT=0.3, D_GM=0.01, D_WM=0.05, rho_GM=0.05, rho_WM=0.05,
                               coupling=0.1)
    t0 = time.perf_counter()
    sol = sim.run(keep_nth=1, save_method=None, plot=False, results_on_device=lazy)
    t_run = time.perf_counter() - t0
    st = sim.solver_statistics()
    t0 = time.perf_counter()
    u10 = sim.results.get_solution_function(subspace_name='displacement', recording_step=steps // 2).values()
    t_get = time.perf_counter() - t0
    print("results_on_device=%s: run() %.2f s for %d recorded steps (%d nodes, %d unknowns): %.2e DoF-updates/s; "
          "device step time %.1f ms; mechanics solves %d (%d PCG its); displacement of step %d on demand: %.2f s, max |u| %.3e"
          % (lazy, t_run, steps, w.mesh.num_vertices(), 4 * w.mesh.num_vertices(),
             4 * w.mesh.num_vertices() * steps / t_run, st['ms_steps'] / steps, st['mech_solves'], st['mech_cg_its'],
             steps // 2, t_get, np.abs(u10).max()), flush=True)
    sim.close()
