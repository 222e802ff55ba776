"""
The public drop-in surface (TumorGrowth / TumorGrowthBrain) on the GPU: BASELINE config C1 written the way the
reference's script writes it (test_case_simulation_tumor_growth_2D_subdomains.py:31-107), checked against the
oracle and the committed golden fixture; TG == TGB (the reference's claimed equivalence, K7); run() semantics.
"""
import logging
import os

import numpy as np
import pytest

from glimslib_amd import fenics_local as fenics
from glimslib_amd.simulation import TumorGrowth, TumorGrowthBrain
from oracle.glims_oracle import OracleTumorGrowth, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


class Boundary(fenics.SubDomain):
    def inside(self, x, on_boundary):
        return on_boundary


def _c1_sim(sim_time=10, **kw):
    nx = ny = 50
    mesh = fenics.RectangleMesh(fenics.Point(-5, -5), fenics.Point(5, 5), nx, ny)
    labels = fenics.Expression('(x[0]>=0.0) ? (1.0) : (2.0)', degree=1)
    tissue_map = {0: 'outside', 1: 'A', 2: 'B'}
    dirichlet_bcs = {'clamped_outside': {'bc_value': fenics.Constant((0.0, 0.0)), 'named_boundary': 'boundary_all',
                                         'subspace_id': 0}}
    u_0_conc_expr = fenics.Expression('sqrt(pow(x[0]-x0,2)+pow(x[1]-y0,2)) < 0.4 ? (1.0) : (0.0)', degree=1,
                                      x0=2.5, y0=2.5)
    sim = TumorGrowth(mesh, **kw)
    sim.setup_global_parameters(label_function=labels, domain_names=tissue_map, boundaries={'boundary_all': Boundary()},
                                dirichlet_bcs=dirichlet_bcs, von_neumann_bcs={})
    sim.setup_model_parameters(iv_expression={0: fenics.Constant((0.0, 0.0)), 1: u_0_conc_expr},
                               diffusion={'outside': 0.0, 'A': 0.1, 'B': 0.0},
                               coupling={'outside': 0.0, 'A': 0.2, 'B': 0.0},
                               proliferation={'outside': 0.0, 'A': 0.1, 'B': 0.0},
                               E={'outside': 10E6, 'A': 0.001, 'B': 0.001},
                               poisson={'outside': 0.49, 'A': 0.40, 'B': 0.10},
                               sim_time=sim_time, sim_time_step=1)
    return sim


def test_config_c1_through_the_reference_style_script(tmp_path):
    sim = _c1_sim()
    sol = sim.run(save_method=None, plot=False, output_dir=str(tmp_path))
    g = np.load(os.path.join(GOLD, "oracle_c1.npz"))
    c, u = sol.components[1], sol.components[0].reshape(-1)
    assert rel_l2(c, g['c']) < 1e-9 and rel_l2(u, g['u']) < 1e-8
    # recorded series: t = 0..10, deep copies, subspace access by name (helper_classes.py:1346)
    assert sim.results.get_recording_steps() == list(range(11))
    f5 = sim.results.get_solution_function(subspace_name='concentration', recording_step=5)
    assert f5.values().shape == (2601,) and 0 < f5.values().max() <= 1.0
    assert np.array_equal(sim.results.get_solution_function(subspace_id=1, recording_step=10).values(), c)
    # tissue B (x < -0.2) has D = rho = 0: only the consistent mass matrix couples its first node layers to A
    assert np.abs(c[sim.mesh.points[:, 0] < -3.0]).max() < 1e-6
    lab = sim.subdomains.subdomains.array()
    assert set(np.unique(lab)) == {1, 2}
    st = sim.solver_statistics()
    assert st['steps'] == 10 and st['mech_solves'] == 10
    sim.close()


def test_keep_nth_vtk_output_and_rerun_with_new_parameters(tmp_path):
    sim = _c1_sim(sim_time=6)
    sim.run(keep_nth=3, save_method='vtk', plot=False, output_dir=str(tmp_path))
    assert sim.results.get_recording_steps() == [0, 1, 2]                 # t = 0, 3, 6
    assert sim.solver_statistics()['mech_solves'] == 2                    # displacement only at recorded steps
    assert sorted(f for f in os.listdir(str(tmp_path)) if f.endswith('.vtu')) == \
        ['solution_%05d.vtu' % k for k in range(3)]
    assert os.path.exists(os.path.join(str(tmp_path), 'solution_timeseries.npz'))
    c_first = sim.solution.components[1].copy()
    # run_for_adjoint: same mesh / space / device discretisation, new scalar parameters (stg:142-155)
    sol = sim.run_for_adjoint([0.05, 0.2, 0.1])
    o = OracleTumorGrowth(sim.mesh.points, sim.mesh.cells, 0.05, 0.2, 0.1,
                          np.array([10E6, 0.001, 0.001])[sim.subdomains.subdomains.array()],
                          np.array([0.49, 0.40, 0.10])[sim.subdomains.subdomains.array()], 1.0)
    _, co = o.run(sim.params.create_initial_value_function().components[1], 6.0, mechanics=False)
    assert rel_l2(sol.components[1], co) < 1e-9
    assert rel_l2(sol.components[1], c_first) > 1e-2
    sim.reload_from_hdf5(os.path.join(str(tmp_path), 'solution_timeseries.h5'))
    assert sim.results.get_recording_steps() == [0, 1, 2]
    sim.close()


def _atlas_like(cls, tissue_map, **model):
    mesh = fenics.BoxMesh(fenics.Point(0, 0, 0), fenics.Point(20, 18, 16), 10, 9, 8)
    mid = mesh.cell_midpoints()
    r = np.linalg.norm((mid - np.array([10, 9, 8])) / np.array([10, 9, 8]), axis=1)
    lab = np.where(r < 0.25, 4, np.where(r < 0.6, 3, np.where(r < 0.85, 2, 1)))    # Ventricles < WM < GM < CSF
    sim = cls(mesh)
    dirichlet = {'clamped_0': {'bc_value': fenics.Constant((0.0, 0.0, 0.0)), 'named_boundary': 'boundary_all',
                               'subspace_id': 0}}
    sim.setup_global_parameters(subdomains=lab, domain_names=tissue_map, boundaries={'boundary_all': Boundary()},
                                dirichlet_bcs=dirichlet, von_neumann_bcs={})
    iv = fenics.Expression('exp(-a*pow(x[0]-x0, 2) - a*pow(x[1]-y0, 2) - a*pow(x[2]-z0,2))', degree=1, a=0.05,
                           x0=14, y0=9, z0=8)
    sim.setup_model_parameters(iv_expression={0: fenics.Expression(('0.0', '0.0', '0.0'), degree=1), 1: iv},
                               sim_time=4, sim_time_step=1, **model)
    return sim


def test_K7_tumor_growth_equals_tumor_growth_brain():
    """test_case_comparison_3D_atlas.py:46-49,87-166: same tissue map literal, same parameter values, two classes."""
    tissue_map = {1: 'CSF', 3: 'WM', 2: 'GM', 4: 'Ventricles'}
    tg = _atlas_like(TumorGrowth, tissue_map,
                     diffusion={'CSF': 0.0, 'WM': 0.05, 'GM': 0.01, 'Ventricles': 0.0},
                     proliferation={'CSF': 0.0, 'WM': 0.05, 'GM': 0.05, 'Ventricles': 0.0}, coupling=0.1,
                     E={'CSF': 1000E-6, 'WM': 3000E-6, 'GM': 3000E-6, 'Ventricles': 1000E-6},
                     poisson={'CSF': 0.45, 'WM': 0.45, 'GM': 0.45, 'Ventricles': 0.3})
    tgb = _atlas_like(TumorGrowthBrain, tissue_map, E_GM=3000E-6, E_WM=3000E-6, E_CSF=1000E-6, E_VENT=1000E-6,
                      nu_GM=0.45, nu_WM=0.45, nu_CSF=0.45, nu_VENT=0.3, D_GM=0.01, D_WM=0.05, rho_GM=0.05,
                      rho_WM=0.05, coupling=0.1)
    a = tg.run(save_method=None, plot=False)
    b = tgb.run(save_method=None, plot=False)
    assert rel_l2(a.components[1], b.components[1]) < 1e-12
    assert rel_l2(a.components[0], b.components[0]) < 1e-9
    assert np.abs(a.components[0]).max() > 0
    # and both equal the oracle with per-cell values looked up by tissue id
    lab = tg.subdomains.subdomains.array()
    t = lambda d: np.array([0.0, d['CSF'], d['GM'], d['WM'], d['Ventricles']])[lab]
    f = tg.mesh.facets()
    bn = np.unique(f['vertices'][f['exterior']])
    dofs = (bn[:, None] * 3 + np.arange(3)).ravel()
    o = OracleTumorGrowth(tg.mesh.points, tg.mesh.cells, t({'CSF': 0.0, 'WM': 0.05, 'GM': 0.01, 'Ventricles': 0.0}),
                          t({'CSF': 0.0, 'WM': 0.05, 'GM': 0.05, 'Ventricles': 0.0}), 0.1,
                          t({'CSF': 1e-3, 'WM': 3e-3, 'GM': 3e-3, 'Ventricles': 1e-3}),
                          t({'CSF': 0.45, 'WM': 0.45, 'GM': 0.45, 'Ventricles': 0.3}), 1.0,
                          dirichlet_u=(dofs, np.zeros(len(dofs))))
    uo, co = o.run(tg.params.create_initial_value_function().components[1], 4.0)
    assert rel_l2(a.components[1], co) < 1e-9 and rel_l2(a.components[0].reshape(-1), uo) < 1e-8
    tg.close()
    tgb.close()


def test_brain_outside_domain_and_missing_tissue():
    tm = {0: 'outside', 1: 'CSF', 3: 'WM', 2: 'GM', 4: 'Ventricles'}
    sim = _atlas_like(TumorGrowthBrain, tm, E_GM=3e-3, E_WM=3e-3, E_CSF=1e-3, E_VENT=1e-3, nu_GM=0.45, nu_WM=0.45,
                      nu_CSF=0.45, nu_VENT=0.3, D_GM=0.01, D_WM=0.05, rho_GM=0.05, rho_WM=0.05, coupling=0.1)
    lab = sim.subdomains.subdomains.array()
    lab[sim.mesh.cell_midpoints()[:, 2] > 14] = 0          # a slab of 'outside' (q1: the reference raises here)
    sol = sim.run(save_method=None, plot=False)
    assert np.isfinite(sol.components[0]).all() and np.isfinite(sol.components[1]).all()
    sim.close()
    bad = _atlas_like(TumorGrowthBrain, {1: 'CSF', 3: 'WM', 2: 'GM'}, E_GM=3e-3, E_WM=3e-3, E_CSF=1e-3, E_VENT=1e-3,
                      nu_GM=0.45, nu_WM=0.45, nu_CSF=0.45, nu_VENT=0.3, D_GM=0.01, D_WM=0.05, rho_GM=0.05,
                      rho_WM=0.05, coupling=0.1)
    with pytest.raises(ValueError):
        bad.run(save_method=None, plot=False)              # cells labelled 4 have no material
    bad.close()


def test_solver_failure_warns_stops_and_returns_last_solution(caplog):
    sim = _c1_sim(sim_time=5, solver_options={'newton_maxit': 0})
    with caplog.at_level(logging.WARNING):
        sol = sim.run(save_method=None, plot=False)
    assert any('did not converge' in r.message for r in caplog.records)    # simulation_base.py:303-305
    assert sim.results.get_recording_steps() == [0]
    assert np.isfinite(sol.components[1]).all()
    sim.close()


def test_time_dependent_source_term_steps_one_by_one():
    sim = _c1_sim(sim_time=3, solver_options={'mechanics': False})
    sim.source_term = fenics.Expression('t < 1.5 ? 0.001 : 0.0', degree=1, t=0.0)
    sol = sim.run(save_method=None, plot=False)
    lab = sim.subdomains.subdomains.array()
    o = OracleTumorGrowth(sim.mesh.points, sim.mesh.cells, np.array([0, .1, 0.])[lab], np.array([0, .1, 0.])[lab], 0.0,
                          1.0, 0.3, 1.0)
    c = sim.params.create_initial_value_function().components[1]
    lumped = np.asarray(o.M.sum(axis=1)).ravel()
    for t in (1.0, 2.0, 3.0):
        o.rd_load = 1.0 * (0.001 if t < 1.5 else 0.0) * lumped
        c, _ = o.rd_step(c)
    assert rel_l2(sol.components[1], c) < 1e-9
    sim.close()
