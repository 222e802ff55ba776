"""Newton residuals from the quadratic structure of the RD residual (default) against a sweep after every solve
(GLIMS_FLAG_FULL_NEWTON).  R(c + delta) = R(c) + A(c) delta + dt N(delta) delta holds exactly for the logistic term of
simulation_tumor_growth.py:115-120, so after a solve with the Jacobian A_0 = A(c_0) the next residual is the Krylov
solver's final residual plus dt N(2 (c_k - c_0) + delta) delta -- one pass over the incidence lists (k_rd_quad)."""
import numpy as np
import pytest

from glimslib_amd import workloads
from glimslib_amd.mesh import RectangleMesh
from oracle.glims_oracle import OracleTumorGrowth, rel_l2

pytestmark = pytest.mark.gpu


def _run(backend, w, tables, steps, flags, **opts):
    h = backend.Handle(w.mesh.points, w.mesh.cells, w.cell_label)
    h.set_materials(tables['D'], tables['rho'], tables['gamma'], tables['E'], tables['nu'])
    h.set_options(dt=w.dt, flags=flags, **opts)
    h.setup(False)
    h.set_state(w.c0)
    status = h.step(steps)
    c = h.get_state(want_u=False)[0]
    st = h.stats()
    h.close()
    return status, c, st


def _c3_reduced(n):
    w = workloads.config_c3(n)
    hx = 240.0 / n
    w.c0 = np.exp(-((w.mesh.points - np.array([118.0, -109.0, 72.0])) ** 2).sum(axis=1) / (2.0 * (2.5 * hx) ** 2))
    return w


def test_no_more_newton_iterations_and_fewer_sweeps_same_fields(backend):
    """BASELINE config C3's parameters (dt rho = 0.05) on a reduced mesh, 12 steps: no more Newton and Krylov iterations
    than the full-Newton path, mid-step sweeps replaced by cheap passes, fields equal to solver tolerance, and equal to the
    oracle's Newton + LU."""
    w = _c3_reduced(24)
    # (fixed forcing term: every solve to cg_rtol, i.e. steps of three to four Newton iterations with evaluations in the
    #  middle of a step -- the regime the cheap residuals were built for; the default forcing is compared at the end)
    ff = backend.FLAG_FIXED_FORCING
    s1, c1, st1 = _run(backend, w, w.tables, 12, backend.FLAG_WARM_START | ff)
    s2, c2, st2 = _run(backend, w, w.tables, 12, backend.FLAG_WARM_START | backend.FLAG_FULL_NEWTON | ff)
    assert s1 == 0 and s2 == 0
    print("quadratic updates: Newton %d, PCG %d, sweeps %d, cheap passes %d | full Newton: %d, %d, %d, %d" %
          (st1['newton_its'], st1['cg_its'], st1['rd_assemblies'], st1['rd_quad_updates'],
           st2['newton_its'], st2['cg_its'], st2['rd_assemblies'], st2['rd_quad_updates']))
    assert st2['rd_quad_updates'] == 0 and st1['rd_quad_updates'] >= 10
    assert st1['rd_assemblies'] + st1['rd_quad_updates'] <= st2['rd_assemblies'] + 2
    # (never more than the full-Newton path)
    # (Krylov passes: a dot-free solve that is followed by a cheap evaluation runs one more operator pass -- the one that leaves
    #  the final residual vector the evaluation builds on; a solve followed by a sweep does not need it)
    assert st1['newton_its'] <= st2['newton_its'] + 2 and st1['cg_its'] <= st2['cg_its'] + 12 + st1['rd_quad_updates']
    assert rel_l2(c1, c2) < 1e-9
    o = OracleTumorGrowth(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.per_cell('gamma'),
                          w.per_cell('E'), w.per_cell('nu'), w.dt)
    co = w.c0
    for _ in range(12):
        co, _ = o.rd_step(co)
    assert rel_l2(c1, co) < 1e-8
    # default options: from a step's second solve on the linear tolerance follows the quadratic remainder -- fewer Newton
    # iterations (a step is first solve, sweep, second solve, confirming sweep), the same fields
    s3, c3, st3 = _run(backend, w, w.tables, 12, backend.FLAG_WARM_START)
    print("default forcing: Newton %d, PCG %d, sweeps %d, cheap passes %d" %
          (st3['newton_its'], st3['cg_its'], st3['rd_assemblies'], st3['rd_quad_updates']))
    assert s3 == 0 and rel_l2(c3, co) < 1e-8 and rel_l2(c3, c1) < 1e-9
    # (on this small, coarse problem the tighter second solves cost Krylov iterations -- 242 against 208 -- for one Newton
    #  iteration less; at the sizes bench.py times, both counts drop: brain-like mesh 4.0 -> 2.25 / 39 -> 30 per step)
    assert st3['newton_its'] <= st1['newton_its'] and st3['cg_its'] <= 1.25 * st1['cg_its']


@pytest.mark.parametrize("dim", [2, 3])
def test_strong_nonlinearity_falls_back_to_sweeps(backend, dim):
    """dt rho = 0.6 with c near the carrying capacity: the Jacobian of the step's first iterate is a poor one for the later
    solves; iterations that contract by less than 10 x are followed by a sweep (a fresh Jacobian), so the run converges to
    the same fields with at most a few more Newton iterations than the full-Newton path."""
    if dim == 3:
        w = _c3_reduced(16)
        tables = dict(w.tables, rho=[12.0 * r for r in w.tables['rho']])
    else:
        mesh = RectangleMesh((0.0, 0.0), (10.0, 8.0), 60, 48)
        lab = np.ones(mesh.num_cells(), dtype=np.int32)
        tables = dict(D=[0.0, 0.05], rho=[0.0, 0.6], gamma=[0.0, 0.1], E=[1.0, 3e-3], nu=[0.3, 0.45])
        c0 = 0.9 * np.exp(-0.2 * ((mesh.points - np.array([5.0, 4.0])) ** 2).sum(axis=1))
        w = workloads.Workload("square", mesh, lab, tables, c0, 1.0, 8, False)
    s1, c1, st1 = _run(backend, w, tables, 8, backend.FLAG_WARM_START)
    s2, c2, st2 = _run(backend, w, tables, 8, backend.FLAG_WARM_START | backend.FLAG_FULL_NEWTON)
    print("dim %d, dt rho = 0.6: Newton %d (sweeps %d, cheap passes %d) against %d with a sweep after every solve; "
          "max c %.3f" % (dim, st1['newton_its'], st1['rd_assemblies'], st1['rd_quad_updates'], st2['newton_its'], c1.max()))
    assert s1 == 0 and s2 == 0
    assert st1['newton_its'] <= st2['newton_its'] + 3       # (after such an iteration the next eight steps use sweeps only)
    assert rel_l2(c1, c2) < 1e-7
    assert st1['last_newton_res'] <= 1e-9 * max(1.0, st2['last_newton_res'] / 1e-13)


def test_not_combined_with_the_extrapolated_guess_or_the_fp32_jacobian(backend):
    """Those two options have no verifying sweep / a rounded operator: they keep a sweep after every solve."""
    w = _c3_reduced(16)
    for flags in (backend.FLAG_EXTRAPOLATE_GUESS, backend.FLAG_WARM_START | backend.FLAG_FP32_JACOBIAN):
        s, c, st = _run(backend, w, w.tables, 4, flags)
        assert s == 0 and st['rd_quad_updates'] == 0


def test_a_second_run_on_the_same_handle_is_bitwise_the_run_of_a_fresh_handle(backend):
    """FenicsSimulation.run() may be called again on the same object (simulation_base.py:166-168; run_for_adjoint does so):
    glims_set_state forgets what the Newton iteration learnt from the previous run (iteration-count hints, contraction
    estimate, midpoint state machine, sweep-only penalty, elasticity solve history), so the second run takes the iteration
    path of a fresh handle and produces the same bits -- concentration, displacement, and the iteration counters."""
    from glimslib_amd import workloads
    w = workloads.config_c3(n=40, mechanics=True)
    n = w.mesh.num_vertices()
    dofs = (np.asarray(w.dirichlet_nodes)[:, None] * 3 + np.arange(3)).ravel()
    c0 = np.exp(-0.01 * ((w.mesh.points - np.array([118.0, -109.0, 72.0])) ** 2).sum(axis=1))

    def fresh():
        h = backend.Handle(w.mesh.points, w.mesh.cells, w.cell_label)
        t = w.tables
        # rho x 4 (and D x 200, so that the front stays resolved on this coarse mesh): a strong reaction term makes steps of
        # four Newton iterations, which switch the midpoint correction on
        h.set_materials([200.0 * d for d in t['D']], [4.0 * r for r in t['rho']], t['gamma'], t['E'], t['nu'])
        h.set_options(dt=w.dt)
        h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
        h.setup(True)
        return h

    def run(h):
        h.set_state(c0)
        h.reset_stats()
        for _ in range(3):
            assert h.step(8) == 0 and h.solve_mechanics() == 0
        c, u = h.get_state()
        st = h.stats()
        return c, u, {k: st[k] for k in ("newton_its", "cg_its", "rd_assemblies", "rd_quad_updates", "midpoint_steps",
                                         "rebase_events", "mech_cg_its")}

    h1 = fresh()
    c1, u1, s1 = run(h1)
    c2, u2, s2 = run(h1)            # same handle, second run
    h2 = fresh()
    c3, u3, s3 = run(h2)            # fresh handle
    h1.close()
    h2.close()
    print("iteration counters of the three runs:", s1, s2, s3)
    assert s1 == s2 == s3
    assert np.array_equal(c1, c3) and np.array_equal(u1, u3)      # run-to-run reproducibility of a fresh handle
    assert np.array_equal(c2, c3) and np.array_equal(u2, u3)      # the re-run
    assert s1["rd_quad_updates"] > 0


def test_first_solve_controller_two_newton_iterations_per_step_through_a_long_run(backend):
    """Default forcing: the first solve of a step runs at 0.3 cg_rtol, and once steps start taking a third iteration (the
    quadratic term of a whole step no longer small: later in a run) its right-hand side gets the midpoint correction from
    the extrapolated increment -- two Newton iterations per step where GLIMS_FLAG_FIXED_FORCING (round 3's rules) takes
    three, the same fields.  BASELINE config C3 at full size, 170 steps (the config itself runs 50)."""
    w = workloads.config_c3()
    steps = 170
    s1, c1, st1 = _run(backend, w, w.tables, steps, backend.FLAG_WARM_START)
    s2, c2, st2 = _run(backend, w, w.tables, steps, backend.FLAG_WARM_START | backend.FLAG_FIXED_FORCING)
    print("default: Newton %.2f, PCG %.2f per step, %d sweeps, %d cheap passes, %d steps with the midpoint correction | "
          "fixed forcing: Newton %.2f, PCG %.2f, %d sweeps, %d cheap passes, %d" %
          (st1['newton_its'] / steps, st1['cg_its'] / steps, st1['rd_assemblies'], st1['rd_quad_updates'], st1['midpoint_steps'],
           st2['newton_its'] / steps, st2['cg_its'] / steps, st2['rd_assemblies'], st2['rd_quad_updates'], st2['midpoint_steps']))
    assert s1 == 0 and s2 == 0
    assert rel_l2(c1, c2) < 1e-9
    assert st1['newton_its'] <= 2.15 * steps            # (the first steps of a run take three)
    assert st2['newton_its'] >= 2.6 * steps
    assert st1['cg_its'] <= 1.1 * st2['cg_its']         # (2.05 against 3.2 Newton iterations per step, 16.6 against 15.7 Krylov passes)
    assert st1['rd_assemblies'] <= 2.1 * steps
    # With PCG in every solve (rd_linear = PCG) steps start taking a third iteration late in the run and the controller answers
    # with the midpoint correction; the dot-free first solves (whose pass count comes from a bound: they end below their
    # tolerance) keep two iterations per step without it.
    s3, c3, st3 = _run(backend, w, w.tables, steps, backend.FLAG_WARM_START, rd_linear=backend.RD_LINEAR_PCG)
    print("PCG in every solve: Newton %.2f, PCG %.2f per step, %d steps with the midpoint correction" %
          (st3['newton_its'] / steps, st3['cg_its'] / steps, st3['midpoint_steps']))
    assert s3 == 0 and rel_l2(c1, c3) < 1e-9
    assert st3['newton_its'] <= 2.15 * steps and st3['midpoint_steps'] >= 20
    assert st1['midpoint_steps'] >= 20 or st1['newton_its'] <= 2.1 * steps
