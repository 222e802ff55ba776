"""
Method of manufactured solutions on the CPU ORACLE: the restatement in oracle/glims_oracle.py against the continuous
equations of the reference's UFL (simulation_tumor_growth.py:110-120), with the same manufactured solution the device
path is held to in tests/test_gpu_mms.py.  The oracle's parity with FEniCS itself cannot be pinned in this container
(no dolfin); what this test pins instead is that the oracle discretises the RIGHT equations: any wrong factor, sign or
missing term in its forms leaves an error that does not fall with order 2 under refinement.
"""
import numpy as np
import pytest

from mms_common import D_, E_, GAMMA, NU, RHO, manufactured, manufactured_transient
from oracle.glims_oracle import OracleTumorGrowth, boundary_facets, box_mesh, rectangle_mesh


def _solve(dim, n):
    pts, cells = rectangle_mesh((0, 0), (1, 1), n, n) if dim == 2 else box_mesh((0, 0, 0), (1, 1, 1), n, n, n)
    cols = [pts[:, a] for a in range(dim)]
    c_f, s_f, u_f, f_f = manufactured(dim)
    bc = lambda v: np.broadcast_to(np.asarray(v, dtype=np.float64), (len(pts),)).copy()
    c_ex, s_n = bc(c_f(*cols)), bc(s_f(*cols))
    u_ex = np.stack([bc(g(*cols)) for g in u_f], axis=1)
    f_n = np.stack([bc(g(*cols)) for g in f_f], axis=1)
    bn = np.unique(boundary_facets(cells)[0])
    dofs = (bn[:, None] * dim + np.arange(dim)).ravel()
    dt = 20.0
    o = OracleTumorGrowth(pts, cells, D_, RHO, GAMMA, E_, NU, dt)
    o.rd_load = dt * (o.M @ s_n)                                   # dt * int s phi_i  with s interpolated
    o.mech_load = np.stack([o.M @ f_n[:, a] for a in range(dim)], axis=1).reshape(-1)
    o.dirichlet_c = (bn, c_ex[bn])
    o.dirichlet_u = (dofs, u_ex.reshape(-1)[dofs])
    c = c_ex.copy()
    for _ in range(12):                                            # backward Euler into the discrete steady state
        c, _its = o.rd_step(c, rtol=1e-10, atol=1e-11)
    c2, _its = o.rd_step(c, rtol=1e-10, atol=1e-11)
    assert np.abs(c2 - c).max() < 1e-10
    u = o.mech_solve(c_ex).reshape(-1, dim)                        # exact concentration: the two error sources stay apart
    l2 = lambda e: np.sqrt(max(e @ (o.M @ e), 0.0))
    return l2(c2 - c_ex), np.sqrt(sum(l2(u[:, a] - u_ex[:, a]) ** 2 for a in range(dim)))


@pytest.mark.parametrize("dim,levels", [(2, (16, 32, 64)), (3, (8, 16, 24))])
def test_oracle_converges_to_the_manufactured_solution_with_order_two(dim, levels):
    errs = [_solve(dim, n) for n in levels]
    ec, eu = np.array([e[0] for e in errs]), np.array([e[1] for e in errs])
    ref = np.log(np.array(levels[1:], dtype=float) / np.array(levels[:-1], dtype=float))
    oc, ou = np.log(ec[:-1] / ec[1:]) / ref, np.log(eu[:-1] / eu[1:]) / ref
    print("oracle, dim %d: L2 errors c %s (orders %s), u %s (orders %s)" %
          (dim, ["%.2e" % e for e in ec], ["%.2f" % o for o in oc], ["%.2e" % e for e in eu], ["%.2f" % o for o in ou]))
    # the coarse 3-D pair (8 -> 16) is still pre-asymptotic (1.90 / 1.88; with n = 32, which takes minutes of sparse LU here,
    # the next pair gives 1.97 / 1.97, and the device test, which can afford 12 / 24 / 48, sees 1.94 .. 2.00); what a wrong
    # form would give is an order near 0
    assert oc.min() > (1.9 if dim == 2 else 1.85) and ou.min() > (1.9 if dim == 2 else 1.85)
    assert oc[-1] > 1.9 and ou[-1] > 1.9


def test_oracle_backward_euler_is_first_order_in_time():
    """Transient manufactured solution with time-dependent source and Dirichlet data on a fine 2-D mesh: the error at
    T = 1 halves with the time step (the M (c - c_prev) term and its dt scaling, which the steady cases do not see)."""
    n, T = 128, 1.0
    pts, cells = rectangle_mesh((0, 0), (1, 1), n, n)
    c_f, s_f = manufactured_transient()
    X, Y = pts[:, 0], pts[:, 1]
    bn = np.unique(boundary_facets(cells)[0])
    errs = []
    for steps in (5, 10, 20):
        dt = T / steps
        o = OracleTumorGrowth(pts, cells, D_, RHO, 0.0, 1.0, 0.3, dt)
        c = np.asarray(c_f(X, Y, 0.0), dtype=np.float64)
        for k in range(1, steps + 1):
            t1 = k * dt
            o.rd_load = dt * (o.M @ s_f(X, Y, t1))
            o.dirichlet_c = (bn, c_f(X[bn], Y[bn], t1))
            c, _its = o.rd_step(c, rtol=1e-10, atol=1e-12)
        e = c - c_f(X, Y, T)
        errs.append(np.sqrt(e @ (o.M @ e)))
    errs = np.array(errs)
    orders = np.log2(errs[:-1] / errs[1:])
    print("oracle, transient: L2 errors %s, temporal orders %s" % (["%.2e" % e for e in errs], ["%.2f" % o for o in orders]))
    assert orders.min() > 0.9 and orders.max() < 1.15
