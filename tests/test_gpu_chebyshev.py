"""
The dot-free RD linear solves (glims_options.rd_linear, ABI 6): Chebyshev semi-iteration fused into the operator pass, interval
from the Lanczos coefficients of recorded PCG solves.  Through the C-ABI, against the oracle and against the PCG path.

Reference counterpart: the KSP behind `self.solver.solve()` (simulation_base.py:302; solver parameters
simulation_tumor_growth.py:126-130) -- the reference's sparse LU is exact, so every Krylov variant here must land on the same
Newton fixed point.
"""
import numpy as np
import pytest

from glimslib_amd import workloads
from oracle.glims_oracle import OracleTumorGrowth, rel_l2

pytestmark = pytest.mark.gpu


def _c3_reduced(n):
    w = workloads.config_c3(n)
    hx = 240.0 / n
    w.c0 = np.exp(-((w.mesh.points - np.array([118.0, -109.0, 72.0])) ** 2).sum(axis=1) / (2.0 * (2.5 * hx) ** 2))
    return w


def _run(backend, w, steps, **opts):
    h = backend.Handle(w.mesh.points, w.mesh.cells, w.cell_label)
    t = w.tables
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    h.set_options(dt=w.dt, **opts)
    h.setup(False)
    h.set_state(w.c0)
    st = h.step(steps)
    c = h.get_state(want_u=False)[0]
    s = h.stats()
    h.close()
    return st, c, s


def test_chebyshev_solves_land_on_the_oracle_and_on_the_pcg_path(backend):
    """Reduced C3 (dt rho = 0.05), 12 steps: the default (auto = Chebyshev after the learning step) against rd_linear = PCG and
    against the oracle's Newton + LU; the dot-free path really ran, needed no fallback, and no more than ~PCG's passes."""
    w = _c3_reduced(28)
    s1, c1, st1 = _run(backend, w, 12)
    s2, c2, st2 = _run(backend, w, 12, rd_linear=backend.RD_LINEAR_PCG)
    assert s1 == 0 and s2 == 0
    print("default: Newton %d, Krylov passes %d (Chebyshev solves %d / passes %d, learning solves %d, fallbacks %d, interval "
          "[%.3f, %.3f]) | PCG: Newton %d, iterations %d" %
          (st1['newton_its'], st1['cg_its'], st1['cheb_solves'], st1['cheb_its'], st1['cheb_learn_solves'],
           st1['cheb_fallbacks'], st1['cheb_lmin'], st1['cheb_lmax'], st2['newton_its'], st2['cg_its']))
    assert st2['cheb_solves'] == 0 and st2['cheb_learn_solves'] == 0
    assert st1['cheb_solves'] >= 15 and st1['cheb_learn_solves'] >= 2 and st1['cheb_fallbacks'] == 0
    assert 0.3 < st1['cheb_lmin'] < 1.0 < st1['cheb_lmax'] < 3.5
    assert st1['newton_its'] <= st2['newton_its'] + 3
    assert st1['cg_its'] <= 1.25 * st2['cg_its'] + 6
    assert rel_l2(c1, c2) < 1e-9
    o = OracleTumorGrowth(w.mesh.points, w.mesh.cells, w.per_cell('D'), w.per_cell('rho'), w.per_cell('gamma'),
                          w.per_cell('E'), w.per_cell('nu'), w.dt)
    co = w.c0
    for _ in range(12):
        co, _ = o.rd_step(co)
    assert rel_l2(c1, co) < 1e-8


def test_a_wrong_interval_is_taken_back_and_the_step_repeated_with_pcg(backend, monkeypatch):
    """TEST HOOK GLIMS_CHEB_TEST_SCALE_HI = 0.45: the upper end of the interval is set far below the spectrum, the polynomial
    grows on most of the right-hand side, the Newton residual does not contract -> the correction is taken back, the iteration
    repeated with PCG, the interval measured again; the run still lands on the PCG path's field."""
    w = _c3_reduced(24)
    s2, c2, st2 = _run(backend, w, 8, rd_linear=backend.RD_LINEAR_PCG)
    monkeypatch.setenv("GLIMS_CHEB_TEST_SCALE_HI", "0.45")
    s1, c1, st1 = _run(backend, w, 8)
    monkeypatch.delenv("GLIMS_CHEB_TEST_SCALE_HI")
    print("wrong interval: %d Chebyshev solves, %d taken back, %d learning solves; Newton %d vs %d" %
          (st1['cheb_solves'], st1['cheb_fallbacks'], st1['cheb_learn_solves'], st1['newton_its'], st2['newton_its']))
    assert s1 == 0 and s2 == 0
    assert st1['cheb_fallbacks'] >= 1 and st1['cheb_learn_solves'] >= 4
    assert rel_l2(c1, c2) < 1e-9


def test_chebyshev_with_dirichlet_concentration_and_source(backend):
    """Constrained rows (Dirichlet c) and a load vector through the dot-free path: equal to the PCG path."""
    w = _c3_reduced(20)
    f = w.mesh.facets()
    bn = np.unique(f['vertices'][f['exterior']])
    res = {}
    for name, lin in (("cheb", backend.RD_LINEAR_CHEBYSHEV), ("pcg", backend.RD_LINEAR_PCG)):
        h = backend.Handle(w.mesh.points, w.mesh.cells, w.cell_label)
        t = w.tables
        h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
        h.set_options(dt=w.dt, rd_linear=lin)
        h.set_dirichlet_c(bn, np.full(len(bn), 0.01))
        h.set_rd_load(1e-3 * np.exp(-((w.mesh.points - np.array([100.0, -100.0, 70.0])) ** 2).sum(axis=1) / 400.0))
        h.setup(False)
        h.set_state(w.c0)
        assert h.step(6) == 0
        res[name] = (h.get_state(want_u=False)[0], h.stats())
        h.close()
    assert res["cheb"][1]['cheb_solves'] > 0 and res["cheb"][1]['cheb_fallbacks'] == 0
    assert np.allclose(res["cheb"][0][bn], 0.01)
    assert rel_l2(res["cheb"][0], res["pcg"][0]) < 1e-9


def test_brain_like_mesh_chebyshev_against_pcg(backend):
    """The unstructured brain-like mesh (reduced): true spectrum of Dinv A far wider than what the right-hand sides excite
    (tools/proto_chebyshev.py: [0.17, 3.3] against Ritz [0.66, 2.0]) -- the dot-free path must still land on PCG's field, and
    if a solve is taken back the run recovers."""
    w = workloads.config_brain_like(60000, isolate=True)
    s1, c1, st1 = _run(backend, w, 10)
    s2, c2, st2 = _run(backend, w, 10, rd_linear=backend.RD_LINEAR_PCG)
    print("brain-like 60 k: Chebyshev solves %d, passes %d, fallbacks %d, interval [%.3f, %.3f]; Krylov passes %d vs PCG %d; "
          "Newton %d vs %d" % (st1['cheb_solves'], st1['cheb_its'], st1['cheb_fallbacks'], st1['cheb_lmin'], st1['cheb_lmax'],
                               st1['cg_its'], st2['cg_its'], st1['newton_its'], st2['newton_its']))
    assert s1 == 0 and s2 == 0
    assert st1['cheb_solves'] > 0
    assert rel_l2(c1, c2) < 1e-9
    assert st1['cg_its'] <= 1.3 * st2['cg_its'] + 10


def test_stream_policy_and_explicit_chebyshev_do_not_change_the_bits(backend):
    """The cache policy of the operator streams (glims_options.stream_policy) only changes how loads are issued: cached and
    non-temporal runs are bitwise equal, counters included; rd_linear = CHEBYSHEV differs from AUTO only where AUTO's cost model
    sends a tight solve to PCG -- on this problem nowhere."""
    w = _c3_reduced(24)
    s1, c1, st1 = _run(backend, w, 10, stream_policy=backend.STREAM_CACHED)
    s2, c2, st2 = _run(backend, w, 10, stream_policy=backend.STREAM_NONTEMPORAL)
    s3, c3, st3 = _run(backend, w, 10, rd_linear=backend.RD_LINEAR_CHEBYSHEV)
    assert s1 == 0 and s2 == 0 and s3 == 0
    assert st1['stream_nontemporal'] == 0 and st2['stream_nontemporal'] == 1
    assert np.array_equal(c1, c2)
    for k in ('newton_its', 'cg_its', 'cheb_its', 'cheb_solves', 'rd_assemblies', 'rd_quad_updates'):
        assert st1[k] == st2[k], k
    assert st1['krylov_working_set'] > 0
    assert np.array_equal(c1, c3) or rel_l2(c1, c3) < 1e-10


def test_guesses_of_both_solves_change_the_counts_not_the_fields(backend):
    """Both linear solves of a step start from a guess (GLIMS_FLAG_WARM_START, default): the first from the extrapolated increment,
    the second from the extrapolated second correction of the two steps before (solver.hip: k_d2_guess; on the PCG path applied only
    if it brings the residual down to a fifth).  Reduced C3, 40 steps, on the dot-free path and on the PCG path, each against the
    same run without guesses: the same fields (the Newton tolerance decides what a step returns, not where its solves start),
    fewer Krylov passes, no more Newton iterations than a few."""
    w = _c3_reduced(32)
    out = {}
    for name, opts in (("dot-free", {}), ("pcg", dict(rd_linear=backend.RD_LINEAR_PCG))):
        s_w, c_w, st_w = _run(backend, w, 40, **opts)
        h = backend.Handle(w.mesh.points, w.mesh.cells, w.cell_label)
        flags = h.options.flags & ~backend.FLAG_WARM_START   # (the library's defaults, minus the guesses)
        h.close()
        s_c, c_c, st_c = _run(backend, w, 40, flags=flags, **opts)
        assert s_w == 0 and s_c == 0
        print("%s: with guesses Newton %d, Krylov passes %d (take-backs %d) | without Newton %d, passes %d | fields %.1e apart" %
              (name, st_w['newton_its'], st_w['cg_its'], st_w['cheb_fallbacks'], st_c['newton_its'], st_c['cg_its'],
               rel_l2(c_w, c_c)))
        assert rel_l2(c_w, c_c) < 1e-9
        assert st_w['cg_its'] < 0.9 * st_c['cg_its']
        assert st_w['newton_its'] <= st_c['newton_its'] + 4
        out[name] = c_w
    assert rel_l2(out["dot-free"], out["pcg"]) < 1e-9
