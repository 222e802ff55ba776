"""
Method of manufactured solutions on the HIP path: the discrete forms against the CONTINUOUS equations that the
reference's UFL states (simulation_tumor_growth.py:110-120), not against this repository's own restatement.

  RD block   -div(D grad c) - rho c (1 - c) = s            (steady state of F_rd, reached by implicit steps)
  mechanics  -div sigma(u) + grad(gamma (2 mu + d lambda) c) = f,   sigma = 2 mu eps(u) + lambda tr eps(u) I
             (F_m: int sigma(u):eps(v) - int sigma(v):(c gamma I) - int f.v, integrated by parts)

A smooth solution is chosen, the source that makes it exact is derived symbolically (sympy), the loads enter through
glims_set_rd_load / glims_set_mech_load as M * (nodal source) (the mass operator hook), Dirichlet data are the exact
solution on the boundary.  P1 elements: the L2 error must fall with order 2 under uniform refinement (three levels).
A wrong factor, sign or missing term in any form shows up as an error that does not converge at all.
"""
import numpy as np
import pytest

from glimslib_amd.mesh import BoxMesh, RectangleMesh
from mms_common import D_, E_, GAMMA, NU, RHO, manufactured as _manufactured, manufactured_transient

pytestmark = pytest.mark.gpu


def _solve(backend, dim, n):
    mesh = RectangleMesh((0, 0), (1, 1), n, n) if dim == 2 else BoxMesh((0, 0, 0), (1, 1, 1), n, n, n)
    P = mesh.points
    cols = [P[:, a] for a in range(dim)]
    c_f, s_f, u_f, f_f = _manufactured(dim)
    bc = lambda v: np.broadcast_to(np.asarray(v, dtype=np.float64), (len(P),)).copy()
    c_ex, s_n = bc(c_f(*cols)), bc(s_f(*cols))
    u_ex = np.stack([bc(g(*cols)) for g in u_f], axis=1)
    f_n = np.stack([bc(g(*cols)) for g in f_f], axis=1)
    fac = mesh.facets()
    bn = np.unique(fac['vertices'][fac['exterior']])
    dofs = (bn[:, None] * dim + np.arange(dim)).ravel()
    dt = 20.0
    h = backend.Handle(P, mesh.cells, np.ones(mesh.num_cells(), np.int32))
    h.set_materials([0, D_], [0, RHO], [0, GAMMA], [1, E_], [0.3, NU])
    h.set_options(dt=dt, mech_rtol=1e-12)
    h.setup(True)
    h.set_rd_load(dt * h.apply(2, s_n)[0])                        # dt * int s phi_i  with s interpolated
    h.set_mech_load(np.stack([h.apply(2, f_n[:, a])[0] for a in range(dim)], axis=1))
    h.set_dirichlet_c(bn, c_ex[bn])
    h.set_dirichlet_u(dofs, u_ex.reshape(-1)[dofs])
    h.set_state(c_ex)
    assert h.step(12) == 0                                         # backward Euler into the discrete steady state
    c1 = h.get_state(want_u=False)[0]
    assert h.step(1) == 0
    c2 = h.get_state(want_u=False)[0]
    assert np.abs(c2 - c1).max() < 1e-11                           # steady
    # mechanics with the EXACT concentration, so that the two error sources stay separate
    h.set_state(c_ex)
    assert h.solve_mechanics() == 0
    u = h.get_state()[1].reshape(-1, dim)
    l2 = lambda e: np.sqrt(max(e @ h.apply(2, e)[0], 0.0))
    ec = l2(c2 - c_ex)
    eu = np.sqrt(sum(l2(u[:, a] - u_ex[:, a]) ** 2 for a in range(dim)))
    h.close()
    return ec, eu


@pytest.mark.parametrize("dim,levels", [(2, (16, 32, 64)), (3, (12, 24, 48))])   # n = 8 in 3-D is pre-asymptotic (1.90)
def test_manufactured_solutions_converge_with_order_two(backend, dim, levels):
    errs = [_solve(backend, dim, n) for n in levels]
    ec, eu = np.array([e[0] for e in errs]), np.array([e[1] for e in errs])
    oc, ou = np.log2(ec[:-1] / ec[1:]), np.log2(eu[:-1] / eu[1:])
    print("dim %d: L2 errors c %s (orders %s), u %s (orders %s)" %
          (dim, ["%.2e" % e for e in ec], ["%.2f" % o for o in oc], ["%.2e" % e for e in eu], ["%.2f" % o for o in ou]))
    assert oc.min() > 1.9 and ou.min() > 1.9
    assert ec[-1] < 2e-3 and eu[-1] < 2e-3


def test_backward_euler_is_first_order_in_time(backend):
    """Transient manufactured solution on a fine 2-D mesh, source and Dirichlet data changing every step
    (glims_set_rd_load / glims_set_dirichlet_c between the steps): the error at T = 1 halves with the time step.  This is
    the M (c - c_prev) term and its dt scaling, which the steady cases do not see, and the time-dependent boundary data
    of simulation_base.py's run loop against an analytical answer."""
    n, T = 128, 1.0
    mesh = RectangleMesh((0, 0), (1, 1), n, n)
    P = mesh.points
    X, Y = P[:, 0].copy(), P[:, 1].copy()
    c_f, s_f = manufactured_transient()
    fac = mesh.facets()
    bn = np.unique(fac['vertices'][fac['exterior']])
    errs = []
    for steps in (5, 10, 20):
        dt = T / steps
        h = backend.Handle(P, mesh.cells, np.ones(mesh.num_cells(), np.int32))
        h.set_materials([0, D_], [0, RHO], [0, 0.0], [1, 1.0], [0.3, 0.3])
        h.set_options(dt=dt)
        h.setup(False)
        h.set_state(np.asarray(c_f(X, Y, 0.0), dtype=np.float64))
        for k in range(1, steps + 1):
            t1 = k * dt
            h.set_rd_load(dt * h.apply(2, np.asarray(s_f(X, Y, t1), dtype=np.float64))[0])
            h.set_dirichlet_c(bn, np.asarray(c_f(X[bn], Y[bn], t1), dtype=np.float64))
            assert h.step(1) == 0
        c = h.get_state(want_u=False)[0]
        e = c - c_f(X, Y, T)
        errs.append(np.sqrt(max(e @ h.apply(2, e)[0], 0.0)))
        h.close()
    errs = np.array(errs)
    orders = np.log2(errs[:-1] / errs[1:])
    print("transient: L2 errors %s, temporal orders %s" % (["%.2e" % e for e in errs], ["%.2f" % o for o in orders]))
    assert orders.min() > 0.9 and orders.max() < 1.15
