"""
Manufactured solutions shared by tests/test_gpu_mms.py (device path) and tests/test_oracle_mms.py (CPU oracle): a smooth
(c, u) and the sources that make it an exact solution of the CONTINUOUS equations the reference's UFL states
(simulation_tumor_growth.py:110-120), derived symbolically:

  RD block   -div(D grad c) - rho c (1 - c) = s            (steady state of F_rd)
  mechanics  -div sigma(u) + grad(gamma (2 mu + d lambda) c) = f,   sigma = 2 mu eps(u) + lambda tr eps(u) I
"""
import sympy as sp

D_, RHO, GAMMA, E_, NU = 0.7, 1.3, 0.2, 2.5, 0.3
MU = E_ / (2 * (1 + NU))
LAM = E_ * NU / ((1 + NU) * (1 - 2 * NU))


def manufactured(dim):
    """-> (c, s, [u_a], [f_a]) as numpy callables of the coordinates."""
    X = sp.symbols('x y z')[:dim]
    pi = sp.pi
    c = sp.Rational(3, 10) + sp.Rational(1, 5) * sp.prod([sp.sin(pi * x) for x in X]) + sp.Rational(1, 10) * X[0] * X[-1]
    u = [sp.sin(pi * X[0]) * sp.cos(pi * X[1] / 2) * (1 + (X[-1] if dim == 3 else 0)) / 10,
         X[0] * (1 - X[1]) * sp.exp(X[0] / 2) / 8]
    if dim == 3:
        u.append(sp.sin(pi * X[2] / 2) * (X[0] + X[1] ** 2) / 12)
    lap = lambda f: sum(sp.diff(f, x, 2) for x in X)
    s = -D_ * lap(c) - RHO * c * (1 - c)
    div_u = sum(sp.diff(u[a], X[a]) for a in range(dim))
    kappa = GAMMA * (2 * MU + dim * LAM)
    f = [-(MU * lap(u[a]) + (LAM + MU) * sp.diff(div_u, X[a])) + kappa * sp.diff(c, X[a]) for a in range(dim)]
    fn = lambda e: sp.lambdify(X, e, 'numpy')
    return fn(c), fn(s), [fn(e) for e in u], [fn(e) for e in f]


def manufactured_transient():
    """2-D, time dependent: c(x, y, t) and the source s = c_t - D lap c - rho c (1 - c), as numpy callables of (x, y, t).
    For the temporal order of the backward-Euler step (1), with time-dependent source AND Dirichlet data."""
    x, y, t = sp.symbols('x y t')
    c = sp.Rational(3, 10) + sp.Rational(1, 5) * sp.sin(sp.pi * x) * sp.sin(sp.pi * y) * sp.exp(-t) \
        + sp.Rational(1, 10) * x * y * sp.cos(2 * t)
    s = sp.diff(c, t) - D_ * (sp.diff(c, x, 2) + sp.diff(c, y, 2)) - RHO * c * (1 - c)
    return sp.lambdify((x, y, t), c, 'numpy'), sp.lambdify((x, y, t), s, 'numpy')
