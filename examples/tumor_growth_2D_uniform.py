"""test_case_simulation_tumor_growth_2D_uniform.py with `glimslib` -> `glimslib_amd`."""
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd.simulation import TumorGrowth
from glimslib_amd import fenics_local as fenics
import glimslib_amd.utils.data_io as dio

logging.basicConfig(format='%(levelname)s:%(message)s', level=logging.INFO)


class Boundary(fenics.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary


nx = ny = 50
mesh = fenics.RectangleMesh(fenics.Point(-5, -5), fenics.Point(5, 5), nx, ny)
boundary_dict = {'boundary_all': Boundary()}
dirichlet_bcs = {'clamped_boundary': {'bc_value': fenics.Constant((0.0, 0.0)), 'named_boundary': 'boundary_all',
                                      'subspace_id': 0}}
von_neuman_bcs = {}
u_0_conc_expr = fenics.Expression('exp(-a*pow(x[0]-x0, 2) - a*pow(x[1]-y0, 2))', degree=1, a=1, x0=0.0, y0=0.0)
u_0_disp_expr = fenics.Constant((0.0, 0.0))

sim = TumorGrowth(mesh)
sim.setup_global_parameters(boundaries=boundary_dict, dirichlet_bcs=dirichlet_bcs, von_neumann_bcs=von_neuman_bcs)
sim.setup_model_parameters(iv_expression={0: u_0_disp_expr, 1: u_0_conc_expr}, diffusion=0.1, coupling=1,
                           proliferation=0.1, E=0.001, poisson=0.45, sim_time=5, sim_time_step=1)

output_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), 'output', '2D_uniform')
sim.run(save_method='vtk', plot=True, output_dir=output_path, clear_all=True)
dio.merge_VTUs(output_path, 1, 5)

sim.init_postprocess(os.path.join(output_path, 'postprocess'))
sim.postprocess.save_all()
c = sim.solution.components[1]
print("t = 5: max concentration %.4f, max |u| %.4e, max pressure %.3e" %
      (c.max(), abs(sim.solution.components[0]).max(), sim.postprocess.get_pressure().values().max()))
print(sim.solver_statistics())
