#!/usr/bin/env python3
"""Where does the wall time of a C3-sized run through the public API go?  (cProfile, cumulative top list)"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import fenics_local as fenics, workloads
from glimslib_amd.simulation import TumorGrowthBrain

n = int(sys.argv[1]) if len(sys.argv) > 1 else 99
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
t0 = time.perf_counter()
w = workloads.config_c5(n)
print("workload (mesh, labels, hull nodes): %.2f s" % (time.perf_counter() - t0), flush=True)


class Hull(fenics.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary


def go():
    t = [time.perf_counter()]
    sim = TumorGrowthBrain(w.mesh)
    sim.setup_global_parameters(subdomains=w.cell_label, domain_names={1: 'CSF', 3: 'WM', 2: 'GM', 4: 'Ventricles'},
                                boundaries={'boundary_all': Hull()},
                                dirichlet_bcs={'clamped_0': {'bc_value': fenics.Constant((0.0, 0.0, 0.0)),
                                                             'named_boundary': 'boundary_all', 'subspace_id': 0}})
    t.append(time.perf_counter())
    sim.setup_model_parameters(iv_expression={0: fenics.Constant((0., 0., 0.)), 1: w.c0}, sim_time=steps,
                               sim_time_step=1, E_GM=3000E-6, E_WM=3000E-6, E_CSF=1000E-6, E_VENT=1000E-6, nu_GM=0.45,
                               nu_WM=0.45, nu_CSF=0.45, nu_VENT=0.3, D_GM=0.01, D_WM=0.05, rho_GM=0.05, rho_WM=0.05,
                               coupling=0.1)
    t.append(time.perf_counter())
    sim.run(keep_nth=1, save_method=None, plot=False)
    t.append(time.perf_counter())
    st = sim.solver_statistics()
    sim.close()
    print("setup_global_parameters %.2f s, setup_model_parameters %.2f s, run() %.2f s of which device stepping %.3f s"
          % (t[1] - t[0], t[2] - t[1], t[3] - t[2], st['ms_steps'] / 1e3), flush=True)


pr = cProfile.Profile()
pr.enable()
go()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
