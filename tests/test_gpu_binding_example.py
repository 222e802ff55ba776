"""
INTEGRATION.md section B: the reference-side binding shipped as examples/hip_solver_binding.py -- ``HipSolver`` (ctypes
only) behind the four array getters of an adapter -- run on the GPU through the fenics_local adapter and checked against
the oracle and against the package's own run().
"""
import importlib.util
import os

import numpy as np
import pytest

from glimslib_amd import fenics_local as fenics
from glimslib_amd.simulation import TumorGrowth
from oracle.glims_oracle import OracleTumorGrowth, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Boundary(fenics.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary


def _sim():
    mesh = fenics.RectangleMesh(fenics.Point(-5, -5), fenics.Point(5, 5), 30, 26)
    labels = fenics.project(fenics.Expression('(x[0]>=0.0) ? (1.0) : (2.0)', degree=1), fenics.FunctionSpace(mesh, "DG", 1))
    sim = TumorGrowth(mesh)
    sim.setup_global_parameters(label_function=labels, domain_names={0: 'outside', 1: 'A', 2: 'B'},
                                boundaries={'all': _Boundary()},
                                dirichlet_bcs={'clamp': {'bc_value': fenics.Constant((0.0, 0.0)), 'named_boundary': 'all',
                                                         'subspace_id': 0}}, von_neumann_bcs={})
    iv = fenics.Expression('exp(-0.5*(pow(x[0]-1.5, 2) + pow(x[1]-1.0, 2)))', degree=1)
    sim.setup_model_parameters(iv_expression={0: fenics.Constant((0.0, 0.0)), 1: iv},
                               diffusion={'outside': 0.0, 'A': 0.1, 'B': 0.02}, coupling={'outside': 0.0, 'A': 0.2, 'B': 0.1},
                               proliferation={'outside': 0.0, 'A': 0.1, 'B': 0.05},
                               E={'outside': 1.0, 'A': 0.001, 'B': 0.003}, poisson={'outside': 0.3, 'A': 0.40, 'B': 0.45},
                               sim_time=3, sim_time_step=1)
    return sim


def test_reference_side_binding_example_runs_and_matches():
    spec = importlib.util.spec_from_file_location("hip_solver_binding", os.path.join(ROOT, "examples", "hip_solver_binding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sim = _sim()
    u_previous = sim.params.create_initial_value_function()
    solution = sim.functionspace.new_function(name='solution_function')
    solver = mod.HipSolver(mod.ShimAdapter(sim), u_previous, solution, dt=sim.params.sim_time_step)
    for _ in range(3):                       # the body of run()'s loop: solver.solve(); u_previous.assign(solution)
        solver.solve()
        u_previous.assign(solution)
    solver.close()
    c, u = solution.components[1], solution.components[0]
    ref = sim.run(save_method=None, plot=False)          # the package's own path on the same set-up
    assert rel_l2(c, ref.components[1]) < 1e-12 and rel_l2(u, ref.components[0]) < 1e-9
    lab = np.asarray(sim.subdomains.subdomains.array())
    t = lambda a, b: np.array([0.0, a, b])[lab]
    f = sim.mesh.facets()
    bn = np.unique(f['vertices'][f['exterior']])
    dofs = (bn[:, None] * 2 + np.arange(2)).ravel()
    o = OracleTumorGrowth(sim.mesh.points, sim.mesh.cells, t(.1, .02), t(.1, .05), t(.2, .1), np.array([1.0, .001, .003])[lab],
                          np.array([.3, .4, .45])[lab], 1.0, dirichlet_u=(dofs, np.zeros(len(dofs))))
    uo, co = o.run(sim.params.create_initial_value_function().components[1], 3.0)
    assert rel_l2(c, co) < 1e-9 and rel_l2(u.reshape(-1), uo) < 1e-8
    sim.close()
