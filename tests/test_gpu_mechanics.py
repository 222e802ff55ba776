"""
GPU tests of the elasticity block (F_m, simulation_tumor_growth.py:110-113) and of BASELINE config C5 (coupled model,
4 DoF per node): the multigrid-preconditioned PCG against the CPU oracle's sparse LU, against the block-Jacobi path,
and -- at C5's full size -- through oracle-free properties.  All through the C-ABI.
"""
import numpy as np
import pytest

from glimslib_amd import workloads
from glimslib_amd.mesh import BoxMesh, RectangleMesh
from oracle.glims_oracle import OracleTumorGrowth, rel_l2

pytestmark = pytest.mark.gpu


def _c5_reduced(n):
    """C5 on a coarser box: the seed of the full-size config (a Gaussian of width 1 mm) falls between the nodes of a
    coarse mesh, so the reduced cases start from one that spans a few cells."""
    w = workloads.config_c5(n)
    hx = 240.0 / n
    w.c0 = np.exp(-((w.mesh.points - np.array([118.0, -109.0, 72.0])) ** 2).sum(axis=1) / (2.0 * (2.5 * hx) ** 2))
    return w


def _c5_handle(backend, w, **opts):
    h = backend.Handle(w.mesh.points, w.mesh.cells, w.cell_label)
    t = w.tables
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    h.set_options(dt=w.dt, **opts)
    d = w.mesh.points.shape[1]
    dofs = (np.asarray(w.dirichlet_nodes)[:, None] * d + np.arange(d)).ravel()
    h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    h.setup(True)
    h.set_state(w.c0)
    return h, dofs


def _c5_oracle(w, dofs):
    return OracleTumorGrowth(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.per_cell('gamma'),
                             w.per_cell('E'), w.per_cell('nu'), w.dt, dirichlet_u=(dofs, np.zeros(len(dofs))))


def _free_residual(h, c, u, dofs):
    """|| (K u - G c) on the free dofs || / || G c ||, through the operator hooks (no oracle)."""
    Ku = h.apply(3, u)[0]
    Gc = h.apply(4, c)[0]
    r = Ku - Gc
    r[dofs] = 0.0
    g = Gc.copy()
    g[dofs] = 0.0
    return np.linalg.norm(r) / np.linalg.norm(g)


@pytest.mark.parametrize("n,monolithic,steps", [(24, False, 3), (14, True, 3)])
def test_config_c5_reduced_matches_the_oracle(backend, n, monolithic, steps):
    """C5 (brain-extent box, WM ellipsoid in GM, u = 0 on the hull) at a size the sparse-LU oracle finishes in seconds:
    split loop (n = 24, 62 500 unknowns) and the reference's monolithic Newton (n = 14)."""
    w = _c5_reduced(n)
    h, dofs = _c5_handle(backend, w)
    o = _c5_oracle(w, dofs)
    uo, co = o.run(w.c0, steps * w.dt, monolithic=monolithic)
    assert h.step(steps) == 0 and h.solve_mechanics() == 0
    c, u = h.get_state()
    st = h.stats()
    print("C5 n=%d (%s): c %.2e, u %.2e, %d PCG its, %d multigrid levels" %
          (n, "monolithic" if monolithic else "split", rel_l2(c, co), rel_l2(u, uo), st['mech_cg_its'], st['mg_levels']))
    assert rel_l2(c, co) < 1e-9 and rel_l2(u, uo) < 1e-8
    assert st['mg_levels'] >= 3 and st['mg_cycles'] > 0
    h.close()


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("smooth", [1, 2, 3])
def test_multigrid_and_block_jacobi_give_the_same_displacement(backend, dim, smooth):
    """Inhomogeneous Dirichlet data on part of the boundary, two tissues with different stiffness and Poisson ratio,
    body load: both preconditioners against the oracle's LU; the multigrid one needs far fewer iterations."""
    if dim == 3:
        mesh = BoxMesh((0, 0, 0), (10.0, 9.0, 8.0), 20, 18, 16)
    else:
        mesh = RectangleMesh((-5, -5), (5, 5), 60, 52)
    mid = mesh.cell_midpoints()
    lab = (1 + (mid[:, 0] > mesh.points[:, 0].mean())).astype(np.int32)
    tabs = dict(D=[0, .1, .02], rho=[0, .1, .05], gamma=[0, .2, .1], E=[1.0, 1e-3, 3e-3], nu=[.3, .40, .45])
    f = mesh.facets()
    bn = np.unique(f['vertices'][f['exterior']])
    bn = bn[mesh.points[bn, 0] < mesh.points[:, 0].mean()]          # clamp only the left half of the hull
    dofs = (bn[:, None] * dim + np.arange(dim)).ravel()
    vals = 1e-3 * np.cos(np.arange(len(dofs)))
    rng = np.random.default_rng(dim)
    mload = 1e-6 * rng.standard_normal(mesh.num_vertices() * dim)
    c0 = np.exp(-0.2 * ((mesh.points - mesh.points.mean(0)) ** 2).sum(1))
    per = {k: np.asarray(v)[lab] for k, v in tabs.items()}
    o = OracleTumorGrowth(mesh.points, mesh.cells, per['D'], per['rho'], per['gamma'], per['E'], per['nu'], 1.0,
                          dirichlet_u=(dofs, vals), mech_load=mload)
    uo = o.mech_solve(c0)
    out = {}
    for pre in (backend.PRECOND_MULTIGRID, backend.PRECOND_BLOCK_JACOBI):
        h = backend.Handle(mesh.points, mesh.cells, lab)
        h.set_materials(tabs['D'], tabs['rho'], tabs['gamma'], tabs['E'], tabs['nu'])
        h.set_options(dt=1.0, mech_precond=pre, mg_smooth=smooth, mech_mixed=0)
        h.set_dirichlet_u(dofs, vals)
        h.set_mech_load(mload)
        h.setup(True)
        h.set_state(c0)
        assert h.solve_mechanics() == 0
        out[pre] = (h.get_state()[1], h.stats()['mech_cg_its'])
        h.close()
    (um, im), (ub, ib) = out[backend.PRECOND_MULTIGRID], out[backend.PRECOND_BLOCK_JACOBI]
    print("dim %d, Chebyshev degree %d: multigrid %d its, block-Jacobi %d its" % (dim, smooth, im, ib))
    assert rel_l2(um, uo) < 1e-8 and rel_l2(ub, uo) < 1e-8
    assert np.array_equal(um.reshape(-1)[dofs], vals)
    assert im < 0.5 * ib


def test_multigrid_iteration_count_does_not_grow_with_the_mesh(backend):
    """C5 at n = 16 / 24 / 32 / 48 from a zero guess to rtol 1e-10: mesh-independent iteration count (block-Jacobi:
    168 / 251 / 329 / ...), answers equal to the block-Jacobi path."""
    its = []
    for n in (16, 24, 32, 48):
        w = _c5_reduced(n)
        h, dofs = _c5_handle(backend, w, mech_history=0)
        assert h.solve_mechanics() == 0
        c, u = h.get_state()
        its.append(h.stats()['mech_cg_its'])
        assert _free_residual(h, c, u, dofs) < 1e-9
        h.close()
    print("multigrid PCG iterations at n = 16, 24, 32, 48:", its)
    assert max(its) <= 60 and its[-1] <= its[0] + 8


def test_multigrid_index_streams_give_the_same_bits(backend):
    """The level-0 smoother reads its columns as 16-bit window codes (default) or as int32 (GLIMS_FLAG_INT32_COLUMNS):
    the same columns in the same order, hence bitwise the same displacement and the same iteration count."""
    w = _c5_reduced(24)
    out = []
    for flags in (backend.FLAG_WARM_START, backend.FLAG_WARM_START | backend.FLAG_INT32_COLUMNS):
        h, dofs = _c5_handle(backend, w, mech_history=0, flags=flags)
        assert h.solve_mechanics() == 0
        out.append((h.get_state()[1], h.stats()['mech_cg_its']))
        h.close()
    assert out[0][1] == out[1][1] and np.array_equal(out[0][0], out[1][0])


def test_multigrid_with_a_stiff_clamped_exterior(backend):
    """The situation of TumorGrowthBrain's 'outside' subdomain (simulation_tumor_growth_brain.py:37-38: E = 10e3 next to
    tissue at 3e-3): the shell that touches the clamped hull 3e6 times stiffer than the ellipsoid inside.  1 M nodes: 31
    iterations (no jump: 19; block-Jacobi at a jump of 1e4: 891).  A FLOATING stiff inclusion is the hard case for
    d-linear interpolation (jump 1e2 / 1e4: 27 / 124 iterations at 1 M nodes)."""
    w = _c5_reduced(24)
    E = list(w.tables['E'])
    E[2] *= 3e6
    w.tables = dict(w.tables, E=E)
    h, dofs = _c5_handle(backend, w, mech_history=0, mech_rtol=1e-11)
    assert h.solve_mechanics() == 0
    u = h.get_state()[1]
    its = h.stats()['mech_cg_its']
    h.close()
    uo = _c5_oracle(w, dofs).mech_solve(w.c0)
    print("stiff clamped exterior (3e6), n = 24: %d PCG iterations, u vs LU %.2e" % (its, rel_l2(u, uo)))
    assert rel_l2(u, uo) < 1e-6 and its <= 60


def test_multigrid_near_the_incompressible_limit(backend):
    """nu = 0.49 in every tissue -- the upper end of the range the reference documents (simulation_tumor_growth.py:60,
    'poisson ratio nu: 0.4 ... 0.49'): lambda / mu = 49.  The point-block Chebyshev smoother loses some of its grip
    (39 its at 1 M nodes instead of 19 at nu = 0.45; 0.495 -> 53, 0.499 -> 104) but stays far from the block-Jacobi
    count (1806 at 1 M), and the answer is the oracle's sparse LU."""
    w = _c5_reduced(24)
    w.tables = dict(w.tables, nu=[0.49 if v > 0.4 else v for v in w.tables['nu']])
    its = {}
    for name, pre in (("mg", backend.PRECOND_MULTIGRID), ("bj", backend.PRECOND_BLOCK_JACOBI)):
        h, dofs = _c5_handle(backend, w, mech_history=0, mech_precond=pre, mech_rtol=1e-11)
        assert h.solve_mechanics() == 0
        u = h.get_state()[1]
        its[name] = h.stats()['mech_cg_its']
        h.close()
        if name == "mg":
            uo = _c5_oracle(w, dofs).mech_solve(w.c0)
        assert rel_l2(u, uo) < 1e-7
    print("nu = 0.49, n = 24: PCG iterations", its)
    assert its["mg"] <= 60 and its["mg"] < 0.2 * its["bj"]


def test_multigrid_on_an_unstructured_mesh_and_a_misaligned_lattice(backend):
    """General meshes take the 125-point coarse stencils: a Delaunay mesh (volumes over three decades) and a box mesh
    whose nodes were jittered off the lattice; both against the oracle."""
    w = workloads.config_unstructured(4000, mechanics=True)
    h, dofs = _c5_handle(backend, w, mech_history=0)
    o = _c5_oracle(w, dofs)
    assert h.solve_mechanics() == 0
    st = h.stats()
    print("Delaunay mesh: %d its, %d levels, operator complexity %.2f" % (st['mech_cg_its'], st['mg_levels'], st['mg_complexity']))
    assert rel_l2(h.get_state()[1], o.mech_solve(w.c0)) < 1e-6       # slivers: ill-conditioned K_el, PCG at rtol 1e-10
    assert st['mech_cg_its'] < 200
    h.close()
    mesh = BoxMesh((0, 0, 0), (1.0, 1.2, 0.9), 18, 16, 14)
    rng = np.random.default_rng(3)
    hmin = np.array([1.0 / 18, 1.2 / 16, 0.9 / 14])
    f = mesh.facets()
    bn = np.unique(f['vertices'][f['exterior']])
    interior = np.ones(mesh.num_vertices(), bool)
    interior[bn] = False
    mesh.points[interior] += 0.2 * hmin * (rng.random((interior.sum(), 3)) - 0.5)
    w2 = workloads.Workload("jittered box", mesh, np.ones(mesh.num_cells(), np.int32),
                            dict(D=[0, .1], rho=[0, .1], gamma=[0, .2], E=[1.0, 2e-3], nu=[.3, .42]),
                            np.exp(-6 * ((mesh.points - 0.5) ** 2).sum(1)), 1.0, 1, True, bn)
    h, dofs = _c5_handle(backend, w2, mech_history=0)
    o = _c5_oracle(w2, dofs)
    assert h.solve_mechanics() == 0
    st = h.stats()
    print("jittered box: %d its, complexity %.2f" % (st['mech_cg_its'], st['mg_complexity']))
    assert rel_l2(h.get_state()[1], o.mech_solve(w2.c0)) < 1e-8
    assert st['mech_cg_its'] <= 70
    h.close()


def test_unclamped_body_and_single_pinned_node(backend):
    """No Dirichlet data at all (reference quirk q3: every level then has the rigid-body modes in its kernel) and
    the minimal set of pins: the multigrid path must still converge; strains equal the oracle's."""
    mesh = BoxMesh((0, 0, 0), (4.0, 3.0, 2.0), 12, 10, 8)
    lab = np.ones(mesh.num_cells(), np.int32)
    tabs = dict(D=[0, .1], rho=[0, .1], gamma=[0, .15], E=[1.0, 2e-3], nu=[.3, .4])
    n = mesh.num_vertices()
    # pins that remove exactly the rigid-body motions: node 0 fully, its x-neighbour in y and z, its y-neighbour in z
    i0 = 0
    ix = int(np.flatnonzero((mesh.points[:, 1] == 0) & (mesh.points[:, 2] == 0) & (mesh.points[:, 0] > 0))[0])
    iy = int(np.flatnonzero((mesh.points[:, 0] == 0) & (mesh.points[:, 2] == 0) & (mesh.points[:, 1] > 0))[0])
    dofs = np.array([3 * i0, 3 * i0 + 1, 3 * i0 + 2, 3 * ix + 1, 3 * ix + 2, 3 * iy + 2])
    c_uni = np.full(n, 0.6)
    h = backend.Handle(mesh.points, mesh.cells, lab)
    h.set_materials(tabs['D'], tabs['rho'], tabs['gamma'], tabs['E'], tabs['nu'])
    h.set_options(dt=1.0)
    h.set_dirichlet_u(dofs, np.zeros(6))
    h.setup(True)
    h.set_state(c_uni)
    assert h.solve_mechanics() == 0
    u = h.get_state()[1].reshape(-1, 3)
    # K4: stress-free growth u = gamma c (x - x0)
    assert np.abs(u - 0.15 * 0.6 * (mesh.points - mesh.points[i0])).max() < 1e-8
    h.close()


def test_config_c5_full_size_properties(backend):
    """BASELINE config C5 at FULL size (n = 99: 1 000 000 nodes, 4 000 000 unknowns).  Concentration against the C
    oracle; displacement through oracle-free checks: the residual of K_el u = G c on the free dofs through the operator
    hooks, symmetry of the K_el hook, equality of the mixed-precision and the all-fp64 solve and of the multigrid and
    the block-Jacobi solve, K4 (u = gamma c (x - x0) for a uniform field with rigid-motion pins), iteration count."""
    from oracle.c_port import COracle
    w = workloads.config_c5()
    n = w.mesh.num_vertices()
    co = COracle(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.dt)
    ref = co.step(w.c0, 2, rtol=1e-11, cg_rtol=1e-4)
    co.close()
    h, dofs = _c5_handle(backend, w)
    assert h.step(2) == 0 and h.solve_mechanics() == 0
    c, u = h.get_state()
    st = h.stats()
    its_mg = st['mech_cg_its']
    print("C5 full size: c vs C oracle %.2e; multigrid PCG %d its, %d levels, complexity %.2f, set-up %.0f ms, solve %.0f ms"
          % (rel_l2(c, ref), its_mg, st['mg_levels'], st['mg_complexity'], st['ms_mg_setup'], st['ms_mech']))
    assert rel_l2(c, ref) < 1e-9
    assert its_mg <= 60
    assert _free_residual(h, c, u, dofs) < 5e-10
    assert np.all(u[dofs] == 0.0) and np.isfinite(u).all() and np.abs(u).max() > 0
    rng = np.random.default_rng(7)
    x, y = rng.standard_normal(3 * n), rng.standard_normal(3 * n)
    Kx, Ky = h.apply(3, x)[0], h.apply(3, y)[0]
    assert abs(x @ Ky - y @ Kx) < 1e-11 * abs(x @ Ky)
    h.close()
    # mixed-precision refinement around the multigrid-preconditioned solve (off by default with this preconditioner) and
    # the block-Jacobi preconditioner (mixed by default at this size): same displacement
    h2, _ = _c5_handle(backend, w, mech_mixed=2)
    h2.set_state(c)
    assert h2.solve_mechanics() == 0
    assert rel_l2(h2.get_state()[1], u) < 1e-8
    h2.close()
    h3, _ = _c5_handle(backend, w, mech_precond=backend.PRECOND_BLOCK_JACOBI)
    h3.set_state(c)
    assert h3.solve_mechanics() == 0
    its_bj = h3.stats()['mech_cg_its']
    ub = h3.get_state()[1]
    # both solves stop at the same RESIDUAL (rtol 1e-10); the error behind it scales with the condition number of the
    # preconditioned operator, which is what differs between the two (block-Jacobi: ~1e5)
    print("C5 full size: block-Jacobi PCG %d its, displacement differs by %.2e, its residual %.2e" %
          (its_bj, rel_l2(ub, u), _free_residual(h3, c, ub, dofs)))
    assert rel_l2(ub, u) < 1e-6
    h3.close()
    assert its_mg < 0.2 * its_bj
    # K4 at full size: uniform materials (C5's two tissues share E, nu, gamma), uniform c, rigid-motion pins only
    pts = w.mesh.points
    i0 = 0
    ix = int(np.flatnonzero((pts[:, 1] == pts[i0, 1]) & (pts[:, 2] == pts[i0, 2]) & (pts[:, 0] > pts[i0, 0]))[0])
    iy = int(np.flatnonzero((pts[:, 0] == pts[i0, 0]) & (pts[:, 2] == pts[i0, 2]) & (pts[:, 1] > pts[i0, 1]))[0])
    pins = np.array([3 * i0, 3 * i0 + 1, 3 * i0 + 2, 3 * ix + 1, 3 * ix + 2, 3 * iy + 2])
    h4 = backend.Handle(pts, w.mesh.cells, w.cell_label)
    t = w.tables
    h4.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    h4.set_options(dt=w.dt, mech_rtol=1e-12)   # six pins only: the softest modes are nearly rigid, so that a residual
    h4.set_dirichlet_u(pins, np.zeros(6))      # of 1e-10 still leaves an error of 2e-6 (measured)
    h4.setup(True)
    h4.set_state(np.full(n, 0.5))
    assert h4.solve_mechanics() == 0
    uk = h4.get_state()[1].reshape(-1, 3)
    exact = 0.1 * 0.5 * (pts - pts[i0])
    print("K4 at full size: %d its, max error %.2e of max |u| %.2e" %
          (h4.stats()['mech_cg_its'], np.abs(uk - exact).max(), np.abs(exact).max()))
    assert np.abs(uk - exact).max() < 1e-6 * np.abs(exact).max()
    h4.close()


def test_multigrid_on_a_larger_delaunay_mesh_with_slivers(backend):
    """100 000 random points, Delaunay: cell volumes over three decades and hull slivers, i.e. stiffness entries over
    many decades.  This is the case that exposed the first half-precision smoother copy (one global scale factor: most
    rows fell below the fp16 range, 7 984 iterations at 1 M points); with the symmetrically scaled copy the half- and
    the single-precision smoother need the same number of iterations, a small fraction of block-Jacobi's."""
    w = workloads.config_unstructured(100000, mechanics=True)
    its = {}
    us = {}
    for name, opts in (("half", {}), ("single", dict(flags=backend.FLAG_WARM_START | backend.FLAG_MG_FP32_SMOOTHER)),
                       ("double-vectors", dict(flags=backend.FLAG_WARM_START | backend.FLAG_MG_FP64_VECTORS)),
                       ("block-jacobi", dict(mech_precond=backend.PRECOND_BLOCK_JACOBI))):
        h, dofs = _c5_handle(backend, w, mech_history=0, **opts)
        assert h.solve_mechanics() == 0
        st = h.stats()
        its[name] = st['mech_cg_its']
        us[name] = h.get_state()[1]
        if name == "half":
            assert _free_residual(h, w.c0, us[name], dofs) < 1e-8
        h.close()
    print("Delaunay 100 k points: PCG iterations", its)
    assert abs(its["half"] - its["single"]) <= 0.15 * its["single"] + 2
    # single-precision cycle vectors (the default) against double ones: the same preconditioner up to rounding
    assert abs(its["half"] - its["double-vectors"]) <= 2 and rel_l2(us["half"], us["double-vectors"]) < 1e-6
    assert its["half"] < 0.25 * its["block-jacobi"] and its["half"] < 200
    assert rel_l2(us["half"], us["single"]) < 1e-6 and rel_l2(us["half"], us["block-jacobi"]) < 1e-5


def test_sliver_mesh_far_couplings_are_lumped(backend):
    """Random-point Delaunay meshes have edges whose end points' parents lie more than the stencil radius apart on the first
    grid.  Default: far POSITIVE couplings are lumped onto the rows' own diagonals in the mesh -> grid Galerkin product
    (mg.hip, k_mg_kp) -- never more iterations than with those couplings losing their cross terms only
    (GLIMS_FLAG_MG_NO_LUMPING), same displacement.  On a lattice mesh nothing is dropped: the flag changes nothing."""
    w = workloads.config_unstructured(60000, mechanics=True)
    res = {}
    for name, flags in (("default", backend.FLAG_WARM_START), ("no lumping", backend.FLAG_WARM_START | backend.FLAG_MG_NO_LUMPING)):
        h, dofs = _c5_handle(backend, w, mech_history=0, flags=flags)
        assert h.solve_mechanics() == 0
        res[name] = (h.stats()['mech_cg_its'], h.get_state()[1], _free_residual(h, w.c0, h.get_state()[1], dofs))
        h.close()
    print("Delaunay 60 k points, PCG iterations:", {k: v[0] for k, v in res.items()})
    for name in res:
        assert res[name][2] < 1e-8
        assert rel_l2(res[name][1], res["default"][1]) < 1e-6
    assert res["default"][0] <= res["no lumping"][0] + 1
    wl = _c5_reduced(40)
    its = []
    for flags in (backend.FLAG_WARM_START, backend.FLAG_WARM_START | backend.FLAG_MG_NO_LUMPING):
        h, _ = _c5_handle(backend, wl, mech_history=0, flags=flags)
        assert h.solve_mechanics() == 0
        its.append(h.stats()['mech_cg_its'])
        h.close()
    assert its[0] == its[1]


def test_default_options_over_fifty_coupled_steps(backend):
    """The DEFAULT configuration (multigrid preconditioner, solve-history initial guess of depth 8, no mixed precision) over
    50 consecutive step + solve_mechanics calls: the history keeps K x_k of every stored solve and later solves build
    their initial residual from those products without an operator pass -- so the products must be the operator's own
    (verification pass), not "right-hand side minus recurrence residual".  At the end the TRUE residual, through the
    operator hooks, is at the tolerance and the displacement equals the one of a run without any history."""
    w = _c5_reduced(24)
    h, dofs = _c5_handle(backend, w)                         # defaults
    h0, _ = _c5_handle(backend, w, mech_history=0)
    for _ in range(50):
        assert h.step(1) == 0 and h.solve_mechanics() == 0
        assert h0.step(1) == 0
    assert h0.solve_mechanics() == 0
    c, u = h.get_state()
    c0_, u0_ = h0.get_state()
    st, st0 = h.stats(), h0.stats()
    res = _free_residual(h, c, u, dofs)
    print("50 coupled steps, defaults: true residual %.2e (stats: %.2e), %.1f PCG its per solve with the history, "
          "%d from a zero guess; u vs no-history run %.2e" %
          (res, st['last_mech_res'], st['mech_cg_its'] / 50.0, st0['mech_cg_its'], rel_l2(u, u0_)))
    assert np.array_equal(c, c0_)
    assert res < 10 * 1e-10                                  # mech_rtol = 1e-10
    assert rel_l2(u, u0_) < 1e-8
    assert st['mech_cg_its'] / 50.0 < 0.7 * st0['mech_cg_its']   # the history still pays
    h.close()
    h0.close()


def test_changing_the_history_depth_restarts_the_ring(backend):
    """mech_history 2 -> 8 after three solves: the ring of stored solves restarts empty (a deeper ring over a full shallow
    one would count slots that were never written)."""
    w = _c5_reduced(16)
    h, dofs = _c5_handle(backend, w, mech_history=2)
    for _ in range(3):
        assert h.step(1) == 0 and h.solve_mechanics() == 0
    h.set_options(mech_history=8)
    for _ in range(4):
        assert h.step(1) == 0 and h.solve_mechanics() == 0
    h.set_options(mech_history=3)                            # and shrinking
    for _ in range(2):
        assert h.step(1) == 0 and h.solve_mechanics() == 0
    c, u = h.get_state()
    assert _free_residual(h, c, u, dofs) < 1e-9
    h.close()


def test_cheb_ratio_is_read_by_every_cycle_and_degenerate_frames_are_refused(backend):
    """mg_cheb_ratio on a handle that has already solved takes effect (the A/B sweeps of tools/run_c5.py rely on it) and
    does not rebuild the hierarchy; a global frame without extent on some axis is a usage error."""
    w = _c5_reduced(20)
    h, dofs = _c5_handle(backend, w, mech_history=0)
    assert h.solve_mechanics() == 0
    s1 = h.stats()
    h.set_options(mg_cheb_ratio=4.0)
    h.set_state(w.c0)
    assert h.solve_mechanics() == 0
    s2 = h.stats()
    its1, its2 = s1['mech_cg_its'], s2['mech_cg_its'] - s1['mech_cg_its']
    print("PCG iterations with the default interval %d, with lambda / 4: %d" % (its1, its2))
    assert s2['ms_mg_setup'] == s1['ms_mg_setup'] and its2 > its1
    with pytest.raises(backend.BackendError):
        h.set_mg_frame([0.0, 0.0, 0.0], [1.0, 0.0, 1.0])
    h.close()
