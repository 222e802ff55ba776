"""test_case_comparison_3D_atlas.py with `glimslib` -> `glimslib_amd` (synthetic 4-tissue mesh, 20 steps)."""
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from glimslib_amd.simulation import TumorGrowthBrain, TumorGrowth
from glimslib_amd.simulation_helpers import Comparison
from glimslib_amd import fenics_local as fenics
from _brain_like import brain_like_mesh

logging.basicConfig(format='%(levelname)s:%(message)s', level=logging.WARNING)


class Boundary(fenics.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary


mesh, subdomains = brain_like_mesh(int(sys.argv[1]) if len(sys.argv) > 1 else 24)
tissue_id_name_map = {1: 'CSF', 3: 'WM', 2: 'GM', 4: 'Ventricles'}
boundary_dict = {'boundary_all': Boundary()}
# NB: the reference script writes 'boundary_name' here, which its own parser ignores (no BC at all);
# 'named_boundary' is the spelling that works in both code bases
dirichlet_bcs = {'clamped_0': {'bc_value': fenics.Constant((0.0, 0.0, 0.0)), 'named_boundary': 'boundary_all',
                               'subspace_id': 0}}
u_0_conc_expr = fenics.Expression('exp(-a*pow(x[0]-x0, 2) - a*pow(x[1]-y0, 2) - a*pow(x[2]-z0,2))', degree=1,
                                  a=0.005, x0=118, y0=-109, z0=72)
ivs = {0: fenics.Expression(('0.0', '0.0', '0.0'), degree=1), 1: u_0_conc_expr}
sim_time, sim_time_step = 20, 1
E_GM = E_WM = 3000E-6
E_CSF = E_VENT = 1000E-6
nu_GM = nu_WM = nu_CSF = 0.45
nu_VENT = 0.3
D_GM, D_WM, rho_GM, rho_WM, coupling = 0.01, 0.05, 0.05, 0.05, 0.1

sim_TG = TumorGrowth(mesh)
sim_TG.setup_global_parameters(subdomains=subdomains, domain_names=tissue_id_name_map, boundaries=boundary_dict,
                               dirichlet_bcs=dirichlet_bcs)
sim_TG.setup_model_parameters(iv_expression=ivs,
                              diffusion={'CSF': 0.0, 'WM': D_WM, 'GM': D_GM, 'Ventricles': 0.0}, coupling=coupling,
                              proliferation={'CSF': 0.0, 'WM': rho_WM, 'GM': rho_GM, 'Ventricles': 0.0},
                              E={'CSF': E_CSF, 'WM': E_WM, 'GM': E_GM, 'Ventricles': E_VENT},
                              poisson={'CSF': nu_CSF, 'WM': nu_WM, 'GM': nu_GM, 'Ventricles': nu_VENT},
                              sim_time=sim_time, sim_time_step=sim_time_step)
sim_TG.run(keep_nth=5, save_method=None, plot=False)

sim_TGB = TumorGrowthBrain(mesh)
sim_TGB.setup_global_parameters(subdomains=subdomains, domain_names=tissue_id_name_map, boundaries=boundary_dict,
                                dirichlet_bcs=dirichlet_bcs)
sim_TGB.setup_model_parameters(iv_expression=ivs, sim_time=sim_time, sim_time_step=sim_time_step,
                               E_GM=E_GM, E_WM=E_WM, E_CSF=E_CSF, E_VENT=E_VENT, nu_GM=nu_GM, nu_WM=nu_WM,
                               nu_CSF=nu_CSF, nu_VENT=nu_VENT, D_GM=D_GM, D_WM=D_WM, rho_GM=rho_GM, rho_WM=rho_WM,
                               coupling=coupling)
sim_TGB.run(keep_nth=5, save_method=None, plot=False)

print(Comparison(sim_TG, sim_TGB).compare())
print("TumorGrowth     :", sim_TG.solver_statistics())
print("TumorGrowthBrain:", sim_TGB.solver_statistics())
