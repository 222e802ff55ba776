"""test_case_simulation_tumor_growth_2D_subdomains.py (BASELINE config C1) with `glimslib` -> `glimslib_amd`."""
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd.simulation import TumorGrowth
from glimslib_amd import fenics_local as fenics

logging.basicConfig(format='%(levelname)s:%(message)s', level=logging.INFO)


class Boundary(fenics.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary


nx = ny = 50
mesh = fenics.RectangleMesh(fenics.Point(-5, -5), fenics.Point(5, 5), nx, ny)
# the reference projects this expression on DG1; the label rule int(label(midpoint)) is applied to its cell-vertex values
labels = fenics.project(fenics.Expression('(x[0]>=0.0) ? (1.0) : (2.0)', degree=1), fenics.FunctionSpace(mesh, "DG", 1))
tissue_map = {0: 'outside', 1: 'A', 2: 'B'}
dirichlet_bcs = {'clamped_outside': {'bc_value': fenics.Constant((0.0, 0.0)), 'named_boundary': 'boundary_all',
                                     'subspace_id': 0}}
u_0_conc_expr = fenics.Expression('sqrt(pow(x[0]-x0,2)+pow(x[1]-y0,2)) < 0.4 ? (1.0) : (0.0)', degree=1, x0=2.5, y0=2.5)

sim = TumorGrowth(mesh)
sim.setup_global_parameters(label_function=labels, domain_names=tissue_map, boundaries={'boundary_all': Boundary()},
                            dirichlet_bcs=dirichlet_bcs, von_neumann_bcs={})
sim.setup_model_parameters(iv_expression={0: fenics.Constant((0.0, 0.0)), 1: u_0_conc_expr},
                           diffusion={'outside': 0.0, 'A': 0.1, 'B': 0.0},
                           coupling={'outside': 0.0, 'A': 0.2, 'B': 0.0},
                           proliferation={'outside': 0.0, 'A': 0.1, 'B': 0.0},
                           E={'outside': 10E6, 'A': 0.001, 'B': 0.001},
                           poisson={'outside': 0.49, 'A': 0.40, 'B': 0.10},
                           sim_time=10, sim_time_step=1)
output_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), 'output', '2D_subdomains')
sim.run(save_method='vtk', plot=True, output_dir=output_path, clear_all=True)
print("recorded steps:", sim.results.get_recording_steps())
print("t = 10: max concentration %.4f in tissue A, %.2e beyond x < -3 (tissue B is inert)" %
      (sim.solution.components[1].max(), abs(sim.solution.components[1][mesh.points[:, 0] < -3]).max()))
