"""
GPU parity tests: the HIP path (through the C-ABI, ctypes) against the CPU oracle on the same inputs, against the
committed golden fixtures, and through size-independent properties at BASELINE.json's full sizes.
Tolerance stated by north_star: concentration within 1e-6 rel-L2; these tests hold the HIP path to <= 1e-8.
"""
import os

import numpy as np
import pytest

from glimslib_amd import workloads
from glimslib_amd.mesh import BoxMesh, RectangleMesh
from oracle.glims_oracle import OracleTumorGrowth, rel_l2, boundary_facets

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")

TABS = dict(D=[0.0, 0.1, 0.02, 0.0], rho=[0.0, 0.1, 0.05, 0.0], gamma=[0.0, 0.2, 0.1, 0.3],
            E=[1.0, 1e-3, 3e-3, 2e-3], nu=[0.3, 0.40, 0.45, 0.1])


def _case(dim, ragged=False):
    if dim == 3:
        mesh = BoxMesh((0, 0, 0), (1.0, 1.2, 0.9), 7, 6, 5) if not ragged else BoxMesh((0, 0, 0), (1, 1, 1), 3, 2, 1)
    else:
        mesh = RectangleMesh((-5, -5), (5, 5), 20, 17) if not ragged else RectangleMesh((0, 0), (1, 1), 2, 1)
    mid = mesh.cell_midpoints()
    lab = (1 + (mid[:, 0] > mesh.points[:, 0].mean()) + (mid[:, 1] > np.percentile(mesh.points[:, 1], 80))).astype(np.int32)
    return mesh, lab


def _handle(backend, mesh, lab, dt, tabs=TABS, mechanics=True, **opts):
    h = backend.Handle(mesh.points, mesh.cells, lab)
    h.set_materials(tabs['D'], tabs['rho'], tabs['gamma'], tabs['E'], tabs['nu'])
    h.set_options(dt=dt, **opts)
    h.setup(mechanics)
    return h


def _oracle(mesh, lab, dt, tabs=TABS, **kw):
    per = {k: np.asarray(v)[lab] for k, v in tabs.items()}
    return OracleTumorGrowth(mesh.points, mesh.cells, per['D'], per['rho'], per['gamma'], per['E'], per['nu'], dt, **kw)


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("ragged", [False, True])
def test_operators_match_oracle(backend, dim, ragged):
    """M, S, A(c), K_el, G and the Newton residual, incl. tiny meshes whose only slice is mostly padding."""
    rng = np.random.default_rng(dim)
    mesh, lab = _case(dim, ragged)
    n = mesh.num_vertices()
    dt = 0.7
    h = _handle(backend, mesh, lab, dt)
    o = _oracle(mesh, lab, dt)
    x, xu = rng.standard_normal(n), rng.standard_normal(n * dim)
    Kel, G = o._mech_setup()
    assert rel_l2(h.apply(2, x)[0], o.M @ x) < 1e-14
    assert rel_l2(h.apply(1, x)[0], o.S @ x) < 1e-14
    assert rel_l2(h.apply(3, xu)[0], Kel @ xu) < 1e-14
    assert rel_l2(h.apply(4, x)[0], G @ x) < 1e-14
    c, cp = rng.random(n), rng.random(n)
    assert rel_l2(h.rd_residual(c, cp), o.rd_residual(c, cp)) < 1e-13
    assert rel_l2(h.apply(0, x)[0], o.rd_jacobian(c) @ x) < 1e-14
    st = h.stats()
    assert st['n_rows'] == n and st['nnz'] == o.S.nnz and st['n_corners'] == mesh.cells.size
    h.close()


def test_timing_hooks_leave_the_operator_of_the_state_in_place(backend):
    """glims_apply(which = 8 | 9) are timing hooks that run the assembly sweep / the quadratic-term pass at c = x; once a state
    is set, A(c) of the STATE must be back in place when they return: a following which = 0 / 5 product and the next step see
    the same operator as without the hook."""
    rng = np.random.default_rng(5)
    mesh, lab = _case(3)
    n = mesh.num_vertices()
    h = _handle(backend, mesh, lab, 0.7, mechanics=False)
    o = _oracle(mesh, lab, 0.7)
    c = rng.random(n)
    h.set_state(c)
    assert h.step(1) == 0
    c1 = h.get_state(want_u=False)[0]
    h.rd_residual(c1, c)                       # assembles A(c1)
    x = rng.standard_normal(n)
    ref = h.apply(0, x)[0]
    assert rel_l2(ref, o.rd_jacobian(c1) @ x) < 1e-13
    y8 = h.apply(8, x, reps=2)[0]
    A_x = o.rd_jacobian(x)
    assert rel_l2(y8, -0.5 * ((A_x + o.S) @ x)) < 1e-12
    h.apply(9, x, reps=2)
    # (the state is c1: the hook re-assembled A(c1), not A(x))
    assert np.array_equal(h.apply(0, x)[0], ref) or rel_l2(h.apply(0, x)[0], ref) < 1e-15
    assert rel_l2(h.apply(5, x)[0], ref) < 1e-14
    h.close()


@pytest.mark.parametrize("dim", [2, 3])
def test_time_stepping_matches_oracle_split_and_monolithic(backend, dim):
    mesh, lab = _case(dim)
    n = mesh.num_vertices()
    dt = 0.7
    bf, _ = boundary_facets(mesh.cells)
    bn = np.unique(bf)
    dofs = (bn[:, None] * dim + np.arange(dim)).ravel()
    vals = 0.01 * np.sin(np.arange(len(dofs)))                    # inhomogeneous Dirichlet data
    ctr = mesh.points.mean(0)
    c0 = np.exp(-8 * ((mesh.points - ctr) ** 2).sum(1) / np.ptp(mesh.points[:, 0]) ** 2)
    o = _oracle(mesh, lab, dt, dirichlet_u=(dofs, vals))
    uo, co = o.run(c0, 5 * dt)
    um, cm = o.run(c0, 5 * dt, monolithic=True)                   # what the reference's SNES+LU iterates on
    h = _handle(backend, mesh, lab, dt)
    h.set_dirichlet_u(dofs, vals)
    h.set_state(c0)
    assert h.step(5) == backend.GLIMS_OK and h.solve_mechanics() == backend.GLIMS_OK
    c, u = h.get_state()
    assert rel_l2(c, co) < 1e-9 and rel_l2(u, uo) < 1e-8
    assert rel_l2(c, cm) < 1e-9 and rel_l2(u, um) < 1e-8
    st = h.stats()
    assert st['steps'] == 5 and st['newton_its'] >= 5 and st['mech_solves'] == 1
    h.close()


def test_golden_fixture_box3d(backend):
    g = np.load(os.path.join(GOLD, "oracle_box3d.npz"))
    mesh = BoxMesh((0.0, 0.0, 0.0), (8.0, 9.0, 7.0), 9, 8, 7)
    tabs = dict(D=[0.0, 0.1, 0.02], rho=[0.0, 0.1, 0.05], gamma=[0.0, 0.2, 0.1], E=[1.0, 1e-3, 3e-3],
                nu=[0.3, 0.40, 0.45])
    f = mesh.facets()
    bn = np.unique(f['vertices'][f['exterior']])
    dofs = (bn[:, None] * 3 + np.arange(3)).ravel()
    h = _handle(backend, mesh, g['label'].astype(np.int32), 1.0, tabs)
    h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    h.set_state(g['c0'])
    assert h.step(int(g['n_steps'])) == 0 and h.solve_mechanics() == 0
    c, u = h.get_state()
    assert rel_l2(c, g['c']) < 1e-9 and rel_l2(u, g['u']) < 1e-8
    h.close()


def test_K1_uniform_field_recurrence_and_K2_mass_conservation(backend):
    mesh, lab = _case(3)
    n = mesh.num_vertices()
    tabs = dict(D=[0, .3, .3, .3], rho=[0, .1, .1, .1], gamma=[0] * 4, E=[1] * 4, nu=[.3] * 4)
    h = _handle(backend, mesh, lab, 1.0, tabs, mechanics=False)
    h.set_state(np.full(n, 0.3))
    cn = 0.3
    for _ in range(4):
        assert h.step(1) == 0
        cn = (-(1 - .1) + np.sqrt((1 - .1) ** 2 + 4 * .1 * cn)) / (2 * .1)
        c, _ = h.get_state(want_u=False)
        assert np.abs(c - cn).max() < 1e-11
    h.close()
    tabs = dict(D=[0, .05, .2, 0.0], rho=[0] * 4, gamma=[0] * 4, E=[1] * 4, nu=[.3] * 4)
    h = _handle(backend, mesh, lab, 0.5, tabs, mechanics=False)
    c0 = np.exp(-4 * ((mesh.points - mesh.points.mean(0)) ** 2).sum(1))
    h.set_state(c0)
    m0 = h.apply(2, c0)[0].sum()
    assert h.step(6) == 0
    c, _ = h.get_state(want_u=False)
    # exact with a direct solve; with the iterative one the defect is bounded by the linear residuals, which stop at
    # half the Newton target (1e-10 relative)
    assert abs(h.apply(2, c)[0].sum() - m0) < 1e-10 * abs(m0)
    h.close()


def test_dirichlet_concentration_source_and_neumann_loads(backend):
    mesh, lab = _case(2)
    n = mesh.num_vertices()
    rng = np.random.default_rng(5)
    dt = 0.5
    left = np.flatnonzero(mesh.points[:, 0] == mesh.points[:, 0].min())
    cD = 0.4 + 0.1 * rng.random(len(left))
    load = dt * 0.01 * rng.random(n)
    mload = 1e-5 * rng.standard_normal(n * 2)
    bottom = np.flatnonzero(mesh.points[:, 1] == mesh.points[:, 1].min())
    dofs = (bottom[:, None] * 2 + np.arange(2)).ravel()
    o = _oracle(mesh, lab, dt, dirichlet_c=(left, cD), dirichlet_u=(dofs, np.zeros(len(dofs))), rd_load=load,
                mech_load=mload)
    c0 = 0.2 * rng.random(n)
    c0[left] = cD
    uo, co = o.run(c0, 4 * dt)
    h = _handle(backend, mesh, lab, dt)
    h.set_dirichlet_c(left, cD)
    h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    h.set_rd_load(load)
    h.set_mech_load(mload)
    h.set_state(c0)
    assert h.step(4) == 0 and h.solve_mechanics() == 0
    c, u = h.get_state()
    assert np.array_equal(c[left], cD)
    assert rel_l2(c, co) < 1e-9 and rel_l2(u, uo) < 1e-8
    h.close()


def test_bitwise_reproducible_and_guess_option(backend):
    mesh, lab = _case(3)
    c0 = np.exp(-4 * ((mesh.points - mesh.points.mean(0)) ** 2).sum(1))
    outs = []
    for flags in (backend.FLAG_WARM_START, backend.FLAG_WARM_START, backend.FLAG_EXTRAPOLATE_GUESS, 0,
                  backend.FLAG_EXTRAPOLATE_GUESS | backend.FLAG_WARM_START):
        h = _handle(backend, mesh, lab, 1.0, mechanics=False, flags=flags)
        h.set_state(c0)
        assert h.step(6) == 0
        outs.append(h.get_state(want_u=False)[0])
        h.close()
    assert np.array_equal(outs[0], outs[1])                      # no atomics anywhere: same bits every run
    assert rel_l2(outs[2], outs[0]) < 1e-9                       # a different Newton guess, the same fixed point
    assert rel_l2(outs[3], outs[0]) < 1e-9                       # no warm start of the linear solve, ditto
    assert np.array_equal(outs[4], outs[2])                      # warm start is ignored when extrapolating


def test_failure_semantics_and_usage_errors(backend):
    mesh, lab = _case(2)
    h = backend.Handle(mesh.points, mesh.cells, lab)
    with pytest.raises(backend.BackendError):
        h.setup(True)                                             # materials not set
    h.set_materials(TABS['D'], TABS['rho'], TABS['gamma'], TABS['E'], TABS['nu'])
    h.set_options(dt=1.0, newton_maxit=0)
    h.setup(False)
    with pytest.raises(backend.BackendError):
        h.step(1)                                                 # no state
    h.set_state(np.random.default_rng(0).random(mesh.num_vertices()))
    assert h.step(3) == backend.GLIMS_NOT_CONVERGED               # iteration cap -> status, not an exception
    st = h.stats()
    assert st["steps"] == 0 and st["failed_steps"] == 1             # stops at the failing step, which is not counted as done
    with pytest.raises(backend.BackendError):
        h.solve_mechanics()                                       # mechanics operators were not assembled
    with pytest.raises(backend.BackendError):
        backend.Handle(mesh.points, mesh.cells, lab + 300)        # label outside [0, 256)
    bad = mesh.cells.copy()
    bad[0, 0] = 10 ** 6
    with pytest.raises(backend.BackendError):
        backend.Handle(mesh.points, bad, lab)
    for bad in (dict(mech_precond=7), dict(mg_smooth=0), dict(mg_cheb_ratio=0.5), dict(mech_mixed=3), dict(mg_h_factor=-1.0)):
        with pytest.raises(backend.BackendError):
            h.set_options(**bad)                                  # elasticity-solver options are validated
        for k in bad:
            setattr(h.options, k, getattr(backend.Options(), k))
    opt = backend.Options()
    backend.load_library().glims_options_default(opt)
    for k, _ in backend.Options._fields_:
        setattr(h.options, k, getattr(opt, k))
    h.set_options(newton_maxit=0)
    with pytest.raises(backend.BackendError):
        h.set_mg_frame([0.0, 0.0], [-1.0, 1.0])                  # hi < lo
    with pytest.raises(backend.BackendError, match="Lame"):
        h.set_materials(TABS['D'], TABS['rho'], TABS['gamma'], TABS['E'], [0.3, 0.5, 0.45, 0.1])   # incompressible
    h.close()


# ---- full-size properties (BASELINE configs C2 / C3) ---------------------------------------------------------------
def test_full_size_c2_properties(backend):
    """C2 (103 823 DoF): oracle-free checks -- K1 recurrence on a uniform field, mass conservation for rho = 0,
    linearity of the operator hook, symmetry x.Ay == y.Ax."""
    w = workloads.config_c2()
    n = w.mesh.num_vertices()
    rng = np.random.default_rng(2)
    h = _handle(backend, w.mesh, w.cell_label, w.dt, w.tables, mechanics=False)
    h.set_state(np.full(n, 0.25))
    assert h.step(2) == 0
    cn = 0.25
    for _ in range(2):
        cn = (-(1 - .1) + np.sqrt((1 - .1) ** 2 + 4 * .1 * cn)) / (2 * .1)
    assert np.abs(h.get_state(want_u=False)[0] - cn).max() < 1e-11
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    Ax, Ay = h.apply(0, x)[0], h.apply(0, y)[0]
    assert abs(x @ Ay - y @ Ax) < 1e-11 * abs(x @ Ay)
    assert rel_l2(h.apply(0, 2 * x - 3 * y)[0], 2 * Ax - 3 * Ay) < 1e-14
    h.close()
    tabs = dict(w.tables)
    tabs['rho'] = [0.0, 0.0]
    h = _handle(backend, w.mesh, w.cell_label, w.dt, tabs, mechanics=False)
    h.set_state(w.c0)
    m0 = h.apply(2, w.c0)[0].sum()
    assert h.step(3) == 0
    assert abs(h.apply(2, h.get_state(want_u=False)[0])[0].sum() - m0) < 1e-11 * m0
    h.close()


def test_c2_sized_step_matches_oracle_cg(backend):
    """C2 mesh at n=24 (15 625 DoF): oracle with its own CG at 1e-13 finishes in seconds."""
    w = workloads.config_c2(24)
    o = OracleTumorGrowth(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), 0.0, 1.0, 0.3, w.dt)
    c = w.c0.copy()
    for _ in range(3):
        c, _ = o.rd_step(c, linear='cg')
    h = _handle(backend, w.mesh, w.cell_label, w.dt, w.tables, mechanics=False)
    h.set_state(w.c0)
    assert h.step(3) == 0
    assert rel_l2(h.get_state(want_u=False)[0], c) < 1e-9
    h.close()


@pytest.mark.parametrize("name,steps", [("c2", 3), ("c3", 2)])
def test_full_size_baseline_configs_match_the_c_oracle(backend, name, steps):
    """BASELINE configs C2 (103 823 DoF) and C3 (1 000 000 DoF, two tissues) at FULL size against the C/OpenMP
    oracle (a third independent implementation; finishes in seconds at these sizes)."""
    from oracle.c_port import COracle
    w = workloads.by_name(name)
    co = COracle(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.dt)
    ref = co.step(w.c0, steps, rtol=1e-11, cg_rtol=1e-4)
    h = _handle(backend, w.mesh, w.cell_label, w.dt, w.tables, mechanics=False)
    h.set_state(w.c0)
    assert h.step(steps) == 0
    c = h.get_state(want_u=False)[0]
    assert rel_l2(c, ref) < 1e-9
    x = np.random.default_rng(0).standard_normal(len(ref))
    assert rel_l2(h.apply(2, x)[0], co.apply(2, x)) < 1e-13 and rel_l2(h.apply(1, x)[0], co.apply(1, x)) < 1e-13
    h.close()
    co.close()


@pytest.mark.parametrize("dim", [2, 3])
def test_unstructured_numbering_and_ragged_rows(backend, dim):
    """An 'unstructured' input: interior nodes jittered, node AND cell numbering randomly permuted, local vertex
    order of every cell shuffled (orientation flips), 3 tissues.  Exercises the Morton / sigma renumbering, ragged
    SELL slices and the slot maps; compared with the oracle on the permuted arrays and with the unpermuted run."""
    rng = np.random.default_rng(10 + dim)
    mesh, lab = _case(dim)
    pts = mesh.points.copy()
    f = mesh.facets()
    interior = np.setdiff1d(np.arange(len(pts)), np.unique(f['vertices'][f['exterior']]))
    h_min = np.min(np.ptp(pts, axis=0) / np.array([20, 17] if dim == 2 else [7, 6, 5]))
    pts[interior] += 0.2 * h_min * (rng.random((len(interior), dim)) - 0.5)
    n, m = len(pts), len(mesh.cells)
    pn, pc = rng.permutation(n), rng.permutation(m)              # new node i = old node pn[i]
    inv = np.empty(n, dtype=np.int64)
    inv[pn] = np.arange(n)
    cells2 = inv[mesh.cells[pc]]
    for row in cells2:                                           # shuffle local vertex order per cell
        rng.shuffle(row)
    pts2, lab2 = pts[pn], lab[pc]
    from glimslib_amd.mesh import Mesh
    mesh2 = Mesh(pts2, cells2.astype(np.int32))
    dt = 0.6
    c0 = np.exp(-6 * ((pts - pts.mean(0)) ** 2).sum(1) / np.ptp(pts[:, 0]) ** 2)
    bn = np.unique(f['vertices'][f['exterior']])
    dofs2 = (inv[bn][:, None] * dim + np.arange(dim)).ravel()
    o = _oracle(mesh2, lab2, dt, dirichlet_u=(dofs2, np.zeros(len(dofs2))))
    uo, co = o.run(c0[pn], 4 * dt)
    h = _handle(backend, mesh2, lab2, dt)
    h.set_dirichlet_u(dofs2, np.zeros(len(dofs2)))
    h.set_state(c0[pn])
    assert h.step(4) == 0 and h.solve_mechanics() == 0
    c, u = h.get_state()
    assert rel_l2(c, co) < 1e-9 and rel_l2(u, uo) < 1e-8
    st = h.stats()
    assert st['nnz_padded'] >= st['nnz']
    h.close()
    # same physics in the original numbering
    mesh1 = Mesh(pts, mesh.cells)
    h = _handle(backend, mesh1, lab, dt)
    dofs = (bn[:, None] * dim + np.arange(dim)).ravel()
    h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    h.set_state(c0)
    assert h.step(4) == 0 and h.solve_mechanics() == 0
    c1, u1 = h.get_state()
    h.close()
    assert rel_l2(c, c1[pn]) < 1e-10 and rel_l2(u.reshape(-1, dim), u1.reshape(-1, dim)[pn]) < 1e-8


def test_orphaned_vertex_is_rejected_with_a_message(backend):
    mesh, lab = _case(2)
    pts = np.vstack([mesh.points, [[100.0, 100.0]]])             # a vertex no cell uses (cf. data_io.py:429-467)
    with pytest.raises(backend.BackendError, match="orphan"):
        backend.Handle(pts, mesh.cells, lab)


def test_degenerate_cell_is_rejected_with_a_message(backend):
    """A zero-volume cell would put inf / NaN into every operator; glims_create names it instead."""
    mesh, lab = _case(3)
    pts = mesh.points.copy()
    c = mesh.cells[17]
    pts[c[3]] = (pts[c[0]] + pts[c[1]] + pts[c[2]]) / 3.0          # fourth vertex into the plane of the other three
    flat = np.flatnonzero(np.abs(np.linalg.det(pts[mesh.cells][:, 1:] - pts[mesh.cells][:, :1])) < 1e-14)
    with pytest.raises(backend.BackendError, match="degenerate cell") as ei:
        backend.Handle(pts, mesh.cells, lab)
    assert ("cell %d" % flat[0]) in str(ei.value)


def test_edge_cases_empty_and_degenerate_inputs(backend):
    """Zero steps, an all-zero state (||R_0|| = 0), a single-cell mesh (one slice of 61-62 padding rows), the largest
    admissible tissue id, and a mesh where every cell has rho = D = 0 (the sweep skips every incidence)."""
    # single triangle / single tetrahedron
    for pts, cells in ((np.array([[0., 0.], [1., 0.], [0., 1.]]), np.array([[0, 1, 2]], dtype=np.int32)),
                       (np.array([[0., 0, 0], [1., 0, 0], [0, 1., 0], [0, 0, 1.]]), np.array([[0, 1, 2, 3]], dtype=np.int32))):
        from glimslib_amd.mesh import Mesh
        mesh = Mesh(pts, cells)
        lab = np.array([255], dtype=np.int32)
        tabs = {k: np.zeros(256) for k in ('D', 'rho', 'gamma')}
        tabs['E'], tabs['nu'] = np.ones(256), np.full(256, 0.3)
        tabs['D'][255], tabs['rho'][255], tabs['gamma'][255] = 0.1, 0.2, 0.1
        h = _handle(backend, mesh, lab, 0.5, tabs)
        o = _oracle(mesh, lab, 0.5, tabs)
        c0 = np.linspace(0.1, 0.4, len(pts))
        h.set_state(c0)
        assert h.step(0) == 0                                        # zero steps: state untouched
        assert np.array_equal(h.get_state(want_u=False)[0], c0)
        assert h.step(3) == 0
        _, co = o.run(c0, 1.5, mechanics=False)
        assert rel_l2(h.get_state(want_u=False)[0], co) < 1e-10
        h.set_state(np.zeros(len(pts)))                               # zero state: R_0 = 0, nothing to solve
        assert h.step(2) == 0 and h.stats()['newton_its'] == h.stats()['newton_its']
        assert np.array_equal(h.get_state(want_u=False)[0], np.zeros(len(pts)))
        h.close()
    # inert tissue everywhere: c stays exactly what the mass matrix preserves (M c = M c_prev  =>  c = c_prev)
    mesh, lab = _case(3)
    tabs = dict(D=[0.0] * 4, rho=[0.0] * 4, gamma=[0.0] * 4, E=[1.0] * 4, nu=[0.3] * 4)
    h = _handle(backend, mesh, lab, 1.0, tabs, mechanics=False)
    c0 = np.random.default_rng(3).random(mesh.num_vertices())
    h.set_state(c0)
    assert h.step(2) == 0
    assert np.abs(h.get_state(want_u=False)[0] - c0).max() < 1e-12
    h.close()


def test_options_newton_tolerance_of_the_reference(backend):
    """SNES defaults of the reference (rtol 1e-9): still far inside the 1e-6 parity bar."""
    mesh, lab = _case(3)
    c0 = np.exp(-4 * ((mesh.points - mesh.points.mean(0)) ** 2).sum(1))
    o = _oracle(mesh, lab, 1.0)
    _, co = o.run(c0, 8.0, mechanics=False)
    h = _handle(backend, mesh, lab, 1.0, mechanics=False, newton_rtol=1e-9, newton_atol=1e-10)
    h.set_state(c0)
    assert h.step(8) == 0
    assert rel_l2(h.get_state(want_u=False)[0], co) < 1e-8
    h.close()


def test_delaunay_mesh_matches_both_oracles(backend):
    """Unstructured Delaunay mesh (row lengths ~6..45, volumes over 3 decades): HIP vs the C oracle (RD, 3 steps) and
    vs the numpy oracle (operators), with mechanics clamped on the hull."""
    from oracle.c_port import COracle
    w = workloads.config_unstructured(3000, mechanics=True)
    n = w.mesh.num_vertices()
    h = _handle(backend, w.mesh, w.cell_label, w.dt, w.tables)
    st = h.stats()
    assert st["nnz_padded"] < 1.35 * st["nnz"]                   # sigma-sorted SELL-64 keeps the padding small
    co = COracle(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.dt)
    ref = co.step(w.c0, 3, rtol=1e-11, cg_rtol=1e-4)
    dofs = (w.dirichlet_nodes[:, None] * 3 + np.arange(3)).ravel()
    h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    h.set_state(w.c0)
    assert h.step(3) == 0
    assert rel_l2(h.get_state(want_u=False)[0], ref) < 1e-9
    o = OracleTumorGrowth(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.per_cell('gamma'),
                          w.per_cell('E'), w.per_cell('nu'), w.dt, dirichlet_u=(dofs, np.zeros(len(dofs))))
    x = np.random.default_rng(1).standard_normal(n)
    xu = np.random.default_rng(2).standard_normal(3 * n)
    Kel, G = o._mech_setup()
    assert rel_l2(h.apply(1, x)[0], o.S @ x) < 1e-13 and rel_l2(h.apply(3, xu)[0], Kel @ xu) < 1e-13
    assert rel_l2(h.apply(4, x)[0], G @ x) < 1e-13
    assert h.solve_mechanics() == 0
    u = h.get_state()[1]
    assert rel_l2(u, o.mech_solve(ref)) < 1e-6                   # slivers: K_el is ill-conditioned, PCG at rtol 1e-10
    h.close()
    co.close()


@pytest.mark.parametrize("name", ["c2", "c3"])
def test_baseline_configs_complete_runs_match_the_c_oracle(backend, name):
    """BASELINE configs C2 (20 steps) and C3 (50 steps) run to completion at full size; final concentration against
    the C/OpenMP oracle run with a tighter Newton tolerance.  Bar: 1e-6 (north_star); observed ~1e-10."""
    from oracle.c_port import COracle
    w = workloads.by_name(name)
    co = COracle(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.dt)
    ref = co.step(w.c0, w.n_steps, rtol=1e-11, cg_rtol=1e-4)
    # Default options, then GLIMS_FLAG_FIXED_FORCING (every linear solve to cg_rtol: steps of three to four Newton iterations
    # with cheap residual evaluations in the middle of a step).  Between them the two runs exercise every path the Newton
    # iteration can take -- residuals from the quadratic structure, adaptive forcing, the midpoint correction (default run,
    # late steps) -- and both must land on the oracle's field.
    h = _handle(backend, w.mesh, w.cell_label, w.dt, w.tables, mechanics=False)
    counts = {}
    for fixed in (False, True):
        h.set_options(dt=w.dt, flags=backend.FLAG_WARM_START | (backend.FLAG_FIXED_FORCING if fixed else 0))
        h.set_state(w.c0)
        h.reset_stats()
        assert h.step(w.n_steps) == 0
        c = h.get_state(want_u=False)[0]
        err = rel_l2(c, ref)
        st = h.stats()
        counts[fixed] = st
        print("%s%s: %d steps, rel-L2 vs C oracle %.2e; Newton %d, PCG %d, sweeps %d, cheap residuals %d, midpoint-corrected "
              "steps %d, rebase events %d" % (name, " (fixed forcing)" if fixed else "", w.n_steps, err, st['newton_its'],
                                              st['cg_its'], st['rd_assemblies'], st['rd_quad_updates'], st['midpoint_steps'],
                                              st['rebase_events']))
        assert err < 1e-8
        assert st['steps'] == w.n_steps
    if name == "c3":
        assert counts[False]['rd_quad_updates'] > 0
        assert counts[True]['rd_quad_updates'] > 0 and counts[True]['midpoint_steps'] == 0
        # the forcing that follows the quadratic remainder saves Newton iterations, not accuracy
        assert counts[False]['newton_its'] < 0.8 * counts[True]['newton_its']
    h.close()
    co.close()


def test_column_index_streams_are_bitwise_equivalent(backend, monkeypatch):
    """
    The SpMV and the assembly sweep read the columns either as int32 or as 16-bit (window, offset) codes; slices that
    need more windows than the table holds use int32.  All three situations (all codes / none [GLIMS_FLAG_INT32_COLUMNS]
    / mixed, forced through the test hook GLIMS_WIN_LIMIT) must produce the same bits, on a mesh with several thousand
    slices and on a tiny one.
    """
    rng = np.random.default_rng(5)
    for mesh in (BoxMesh((0, 0, 0), (1.0, 1.3, 0.8), 30, 28, 26), _case(3, ragged=True)[0], _case(2)[0]):
        lab = (1 + (mesh.cell_midpoints()[:, 0] > mesh.points[:, 0].mean())).astype(np.int32)
        n = mesh.num_vertices()
        x = rng.standard_normal(n)
        c0 = np.exp(-4 * ((mesh.points - mesh.points.mean(0)) ** 2).sum(1))
        outs = []
        for int32, env in ((False, {}), (True, {}), (False, dict(GLIMS_WIN_LIMIT="2")), (False, dict(GLIMS_WIN_LIMIT="0"))):
            monkeypatch.delenv("GLIMS_WIN_LIMIT", raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            h = _handle(backend, mesh, lab, 1.0, mechanics=False,
                        flags=backend.FLAG_WARM_START | (backend.FLAG_INT32_COLUMNS if int32 else 0))
            st = h.stats()
            y = h.apply(1, x)[0]
            r = h.rd_residual(c0 + 0.1 * x, c0)
            h.set_state(c0)
            assert h.step(3) == 0
            outs.append((st['nnz_idx16'], st['nnz_padded'], y, r, h.get_state(want_u=False)[0]))
            h.close()
        assert outs[0][0] == outs[0][1]                       # default: every slice of these meshes is coded
        assert outs[1][0] == 0 and outs[3][0] == 0
        if n > 10000:
            assert 0 < outs[2][0] < outs[2][1]                # genuinely mixed
        for o in outs[1:]:
            for a, b in zip(o[2:], outs[0][2:]):
                assert np.array_equal(a, b)


def test_headline_config_c4_matches_the_c_oracle(backend):
    """BASELINE config C4 at its full size (10 077 696 DoF, 59.6 M tetrahedra): two implicit steps on the device against
    the independent C/OpenMP oracle run on the host cores (about half a minute of CPU work)."""
    from oracle.c_port import COracle
    w = workloads.config_c4()
    co = COracle(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.dt)
    ref = co.step(w.c0, 2, rtol=1e-11, cg_rtol=1e-4)
    del co
    h = _handle(backend, w.mesh, w.cell_label, w.dt, w.tables, mechanics=False)
    h.set_state(w.c0)
    assert h.step(2) == 0
    c = h.get_state(want_u=False)[0]
    st = h.stats()
    # oracle-free checks at the same size: symmetry and linearity of A(c), the dot-fused SpMV variant gives the same
    # product, and a uniform field follows the scalar backward-Euler logistic recurrence exactly (K1; rho = 0.05 in
    # both tissues)
    n = w.mesh.num_vertices()
    rng = np.random.default_rng(4)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    Ax, Ay = h.apply(0, x)[0], h.apply(0, y)[0]
    assert abs(x @ Ay - y @ Ax) < 1e-11 * abs(x @ Ay)
    assert rel_l2(h.apply(0, 2 * x - 3 * y)[0], 2 * Ax - 3 * Ay) < 1e-14
    assert np.array_equal(h.apply(5, x)[0], Ax)
    h.set_state(np.full(n, 0.25))
    assert h.step(2) == 0
    cn = 0.25
    for _ in range(2):
        cn = (-(1 - .05) + np.sqrt((1 - .05) ** 2 + 4 * .05 * cn)) / (2 * .05)
    assert np.abs(h.get_state(want_u=False)[0] - cn).max() < 1e-9      # max norm over 10 M values at Newton rtol 1e-10
    h.close()
    err = rel_l2(c, ref)
    print("C4: 2 steps, rel-L2 vs C oracle %.2e (%d rows, %d of %d entries with 16-bit column codes)" %
          (err, st['n_rows'], st['nnz_idx16'], st['nnz_padded']))
    assert st['n_rows'] == 10077696 and st['nnz_idx16'] == st['nnz_padded']
    assert err < 1e-9


def test_elasticity_history_guess_changes_only_the_iteration_count(backend, monkeypatch):
    """Consecutive displacement solves start from the least-squares combination of the previous solutions (K_el is
    linear and time independent).  Same answers as without the history and as the oracle, far fewer iterations."""
    mesh = BoxMesh((0, 0, 0), (10.0, 9.0, 8.0), 16, 14, 12)
    lab = (1 + (mesh.cell_midpoints()[:, 0] > 5.0)).astype(np.int32)
    f = mesh.facets()
    bn = np.unique(f['vertices'][f['exterior']])
    dofs = (bn[:, None] * 3 + np.arange(3)).ravel()
    vals = 1e-3 * np.cos(np.arange(len(dofs)))                     # inhomogeneous Dirichlet data
    c0 = np.exp(-0.2 * ((mesh.points - np.array([5.0, 4.5, 4.0])) ** 2).sum(1))
    o = _oracle(mesh, lab, 1.0, dirichlet_u=(dofs, vals))
    out = {}
    for depth in ("0", "6"):
        # mech_mixed = 2: fp32 inner operator + fp64 refinement, also on this small mesh
        h = _handle(backend, mesh, lab, 1.0, mech_mixed=2, mech_history=int(depth))
        h.set_dirichlet_u(dofs, vals)
        h.set_state(c0)
        us = []
        for _ in range(8):
            assert h.step(1) == 0 and h.solve_mechanics() == 0
            us.append(h.get_state()[1].copy())
        out[depth] = (us, h.stats()['mech_cg_its'])
        c_last = h.get_state()[0]
        h.close()
    for a, b in zip(out["0"][0], out["6"][0]):
        assert rel_l2(a, b) < 1e-8
    assert rel_l2(out["6"][0][-1], o.mech_solve(c_last)) < 1e-8
    assert out["6"][1] < 0.9 * out["0"][1]
    print("elasticity PCG iterations over 8 solves: %d without history, %d with" % (out["0"][1], out["6"][1]))


@pytest.mark.parametrize("seed", range(8))
def test_randomised_small_problems_match_the_oracle(backend, seed):
    """Random mesh size / dimension / tissue layout / materials / Dirichlet sets / loads / dt, one seed per case:
    operators, three coupled steps and the displacement against the scipy oracle."""
    rng = np.random.default_rng(1000 + seed)
    dim = 2 + seed % 2
    h0 = float(rng.uniform(0.2, 2.0))                               # mesh width; per-axis factors keep aspect ratios <= 4
    if dim == 2:
        nx, ny = rng.integers(1, 24, size=2)
        nz = 1
        mesh = RectangleMesh((0, 0), (nx * h0 * float(rng.uniform(0.5, 2)), ny * h0 * float(rng.uniform(0.5, 2))),
                             int(nx), int(ny))
    else:
        nx, ny, nz = rng.integers(1, 9, size=3)
        mesh = BoxMesh((0, 0, 0), tuple(float(k * h0 * f) for k, f in zip((nx, ny, nz), rng.uniform(0.5, 2, size=3))),
                       int(nx), int(ny), int(nz))
    pts = mesh.points + rng.uniform(-0.2, 0.2, size=mesh.points.shape) * (np.ptp(mesh.points, axis=0) / np.array(
        [nx, ny, nz][:dim])) * (0.0 if seed < 2 else 1.0)          # jittered vertices from the third case on
    from glimslib_amd.mesh import Mesh
    mesh = Mesh(pts, mesh.cells)
    n, m = mesh.num_vertices(), mesh.num_cells()
    n_lab = int(rng.integers(1, 5))
    lab = rng.integers(1, n_lab + 1, size=m).astype(np.int32)
    tabs = dict(D=[0.0] + list(rng.uniform(0.0, 0.5, n_lab)), rho=[0.0] + list(rng.uniform(0.0, 0.3, n_lab)),
                gamma=[0.0] + list(rng.uniform(0.0, 0.5, n_lab)), E=[1.0] + list(10.0 ** rng.uniform(-3, 1, n_lab)),
                nu=[0.3] + list(rng.uniform(0.05, 0.47, n_lab)))
    dt = float(rng.uniform(0.2, 2.0))
    bf, _ = boundary_facets(mesh.cells)
    bn = np.unique(bf)
    bn = bn[rng.random(len(bn)) < 0.7] if len(bn) > dim + 2 else bn
    bn = np.union1d(bn, np.unique(bf)[:dim + 1])                   # enough clamped nodes to remove rigid motion
    dofs = (bn[:, None] * dim + np.arange(dim)).ravel()
    vals = 1e-2 * rng.standard_normal(len(dofs))
    kw = dict(dirichlet_u=(dofs, vals))
    load = mload = None
    if seed % 3 == 0:
        load = dt * 0.01 * rng.random(n)
        mload = 1e-4 * rng.standard_normal(n * dim)
        kw.update(rd_load=load, mech_load=mload)
    o = _oracle(mesh, lab, dt, tabs, **kw)
    h = _handle(backend, mesh, lab, dt, tabs)
    h.set_dirichlet_u(dofs, vals)
    if load is not None:
        h.set_rd_load(load)
        h.set_mech_load(mload)
    x = rng.standard_normal(n)
    assert rel_l2(h.apply(1, x)[0], o.S @ x) < 1e-13 and rel_l2(h.apply(2, x)[0], o.M @ x) < 1e-13
    c0 = rng.random(n) * 0.8
    uo, co = o.run(c0, 3 * dt)
    h.set_state(c0)
    assert h.step(3) == 0 and h.solve_mechanics() == 0
    c, u = h.get_state()
    h.close()
    # solution errors = condition number x the residual tolerances (1e-10): material contrasts of 1e4 are allowed here
    assert rel_l2(c, co) < 2e-9, (seed, dim, n)
    assert rel_l2(u, uo) < 1e-6, (seed, dim, n)


def test_optional_fp32_jacobian_converges_to_the_same_fixed_point(backend):
    """GLIMS_FLAG_FP32_JACOBIAN (off by default): the Jacobian is stored in single precision inside the Krylov solves,
    the Newton residual stays fp64 -- the time steps converge to the same fp64 tolerance, the answers agree with the
    oracle like the default path's, and A(c) x through the operator hook is the fp32-rounded operator."""
    mesh = BoxMesh((0, 0, 0), (10.0, 9.0, 8.0), 20, 18, 16)
    lab = (1 + (mesh.cell_midpoints()[:, 0] > 5.0)).astype(np.int32)
    c0 = np.exp(-0.2 * ((mesh.points - np.array([5.0, 4.5, 4.0])) ** 2).sum(1))
    o = _oracle(mesh, lab, 1.0)
    _, co = o.run(c0, 5.0, mechanics=False)
    outs = {}
    for flags in (backend.FLAG_WARM_START, backend.FLAG_WARM_START | backend.FLAG_FP32_JACOBIAN):
        h = _handle(backend, mesh, lab, 1.0, mechanics=False, flags=flags)
        h.set_state(c0)
        assert h.step(5) == 0
        x = np.random.default_rng(1).standard_normal(mesh.num_vertices())
        outs[flags] = (h.get_state(want_u=False)[0], h.apply(0, x)[0], h.stats())
        h.close()
    c64, A64, st64 = outs[backend.FLAG_WARM_START]
    c32, A32, st32 = outs[backend.FLAG_WARM_START | backend.FLAG_FP32_JACOBIAN]
    assert rel_l2(c64, co) < 1e-9 and rel_l2(c32, co) < 1e-9
    assert st32['last_newton_res'] < 1e-9 * max(1.0, st64['last_newton_res'] / 1e-10)
    assert 1e-9 < rel_l2(A32, A64) < 1e-6                          # the hook shows the single-precision operator
    assert st32['newton_its'] <= st64['newton_its'] + 2


def test_long_run_breakdown_is_the_schemes_and_happens_at_the_same_step_as_in_the_c_oracle(backend):
    """
    BASELINE config C4 names 500 steps; with the reference's parameters the consistent-mass P1 scheme undershoots at the
    travelling front (front width sqrt(D/rho) below the mesh width), the logistic term amplifies negative values and
    Newton stops converging when A(c) turns indefinite -- at step 483 on C4, 210-250 on C3 (profiles/r01_long_c*_run.txt).
    Here the same mechanism on a mesh where it takes ~20 steps (n = 16, rho = 0.5, D = 0.05 / 0.25): device and C
    oracle side by side, the same min c trajectory to 4 digits and the same failing step; the device reports
    GLIMS_NOT_CONVERGED, counts the step as failed and keeps stepping possible (reference: simulation_base.py:301-305).
    """
    from oracle.c_port import COracle
    w = workloads.config_c3(16)
    hx = 240.0 / 16
    tabs = dict(w.tables)
    tabs['D'] = [0.0, 0.0, 0.05, 0.25, 0.0]
    tabs['rho'] = [0.0, 0.0, 0.5, 0.5, 0.0]
    c0 = np.exp(-((w.mesh.points - np.array([118.0, -109.0, 72.0])) ** 2).sum(1) / (2 * (1.5 * hx) ** 2))
    Dc, rc = np.asarray(tabs['D'])[w.cell_label], np.asarray(tabs['rho'])[w.cell_label]
    co = COracle(w.mesh.points, w.mesh.cells, Dc, rc, 1.0)
    h = _handle(backend, w.mesh, w.cell_label, 1.0, tabs, mechanics=False)
    h.set_state(c0)
    c_or = c0.copy()
    fail_or = fail_dev = None
    traj = []
    for k in range(1, 60):
        if fail_or is None:
            try:
                c_or = co.step(c_or, 1)
            except RuntimeError:
                fail_or = k
        if fail_dev is None:
            st = h.step(1)
            if st != 0:
                fail_dev = k
                assert st == backend.GLIMS_NOT_CONVERGED
        if fail_or is not None or fail_dev is not None:
            break
        c_dev = h.get_state(want_u=False)[0]
        traj.append((c_or.min(), c_dev.min()))
        assert abs(c_dev.min() - c_or.min()) <= 1e-4 * max(abs(c_or.min()), 1e-3), (k, traj[-1])
        # the nearer the indefinite Jacobian, the more the round-off of two different solvers is amplified
        assert rel_l2(c_dev, c_or) < (1e-8 if k <= 10 else 1e-4)
    print("min c (oracle, device) of the last steps before the breakdown:", ["%.4e / %.4e" % t for t in traj[-4:]])
    print("Newton gives up at step: oracle %s, device %s" % (fail_or, fail_dev))
    assert fail_or is not None and fail_or == fail_dev and 10 < fail_dev < 40
    assert traj[-1][0] < -0.2                                        # the undershoot had grown to O(1)
    st = h.stats()
    assert st['steps'] == fail_dev - 1 and st['failed_steps'] == 1
    h.close()
    co.close()


@pytest.mark.parametrize("dim", [2, 3])
def test_matrix_free_product_equals_the_assembled_one(backend, dim):
    """glims_apply(which = 7) rebuilds (S + 2 dt N(c)) x from the incidence lists for the current state; it must equal
    the assembled Jacobian's product for the same state and the oracle's (the operator the A/B of
    tools/ab_matfree.py times)."""
    mesh, lab = _case(dim)
    n = mesh.num_vertices()
    rng = np.random.default_rng(40 + dim)
    dt = 0.7
    h = _handle(backend, mesh, lab, dt, mechanics=False)
    o = _oracle(mesh, lab, dt)
    c = rng.random(n)
    x = rng.standard_normal(n)
    h.set_state(c)
    h.rd_residual(c, c)
    ya, ym = h.apply(0, x)[0], h.apply(7, x)[0]
    ref = o.rd_jacobian(c) @ x
    assert rel_l2(ym, ref) < 1e-14 and rel_l2(ya, ref) < 1e-14
    h.close()


def test_brain_like_mesh_reduced_matches_the_numpy_oracle(backend):
    """The brain-like unstructured workload (jittered-lattice Delaunay mesh, curved two-tissue interface, config C3's
    parameters; stand-in for the CGAL atlas meshes of test_case_comparison_3D_atlas.py:84-121) at reduced size: operators
    and 5 coupled steps against the numpy oracle's split loop (Newton + sparse LU), displacement clamped on the hull."""
    w = workloads.config_brain_like(4000, mechanics=True, workers=2, isolate=True)
    n = w.mesh.num_vertices()
    dofs = (w.dirichlet_nodes[:, None] * 3 + np.arange(3)).ravel()
    # a wider seed than the workload's (node spacing is ~13 mm at this size): the front must cross the tissue interface
    c0 = np.exp(-0.002 * ((w.mesh.points - np.array([118.0, -109.0, 72.0])) ** 2).sum(axis=1))
    o = OracleTumorGrowth(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.per_cell('gamma'),
                          w.per_cell('E'), w.per_cell('nu'), w.dt, dirichlet_u=(dofs, np.zeros(len(dofs))))
    h = _handle(backend, w.mesh, w.cell_label, w.dt, w.tables)
    h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    x = np.random.default_rng(3).standard_normal(n)
    assert rel_l2(h.apply(2, x)[0], o.M @ x) < 1e-13 and rel_l2(h.apply(1, x)[0], o.S @ x) < 1e-13
    h.set_state(c0)
    u_ref, c_ref = o.run(c0, 5 * w.dt)
    assert h.step(5) == 0 and h.solve_mechanics() == 0
    c, u = h.get_state()
    assert rel_l2(c, c_ref) < 1e-9
    assert rel_l2(u, u_ref) < 1e-7
    h.close()


def test_brain_like_mesh_full_size_matches_the_c_oracle(backend):
    """The same workload at the size bench.py times it (alt.unstructured: ~1.04 M nodes, 6.8 M tetrahedra): 5 steps and the
    static operators against the C/OpenMP oracle, plus size-independent properties (symmetry of S, row sums of M = nodal
    volumes adding up to the box)."""
    from oracle.c_port import COracle
    w = workloads.config_brain_like(1000000, isolate=True)
    n = w.mesh.num_vertices()
    co = COracle(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.dt)
    ref = co.step(w.c0, 5, rtol=1e-11, cg_rtol=1e-4)
    h = _handle(backend, w.mesh, w.cell_label, w.dt, w.tables, mechanics=False)
    h.set_state(w.c0)
    assert h.step(5) == 0
    c = h.get_state(want_u=False)[0]
    err = rel_l2(c, ref)
    print("brain-like, %d nodes: 5 steps, rel-L2 vs C oracle %.2e" % (n, err))
    assert err < 1e-9
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    Sx, Sy = h.apply(1, x)[0], h.apply(1, y)[0]
    assert rel_l2(h.apply(2, x)[0], co.apply(2, x)) < 1e-13 and rel_l2(Sx, co.apply(1, x)) < 1e-13
    assert abs(y @ Sx - x @ Sy) < 1e-11 * abs(y @ Sx)
    assert abs(h.apply(2, np.ones(n))[0].sum() - 240.0 * 240.0 * 155.0) < 1e-7 * 240.0 * 240.0 * 155.0
    st = h.stats()
    assert st["nnz_padded"] < 1.25 * st["nnz"]
    h.close()
    co.close()
