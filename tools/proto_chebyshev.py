#!/usr/bin/env python3
"""
CPU prototype (scipy): Jacobi-PCG against a fixed-length Chebyshev iteration (no dot products, no reductions) on the linear
systems of the RD Newton iteration, same mesh width and parameters as BASELINE configs C3 / C4 (a sub-box around the seed so
that it runs in seconds).  Question (round-4 review, item 1c): how many Chebyshev iterations with spectral bounds taken from a
PCG solve buy the residual reduction that PCG reaches -- the price of a dot-free iteration.

    python3 tools/proto_chebyshev.py [h_mm] [n]        e.g. 2.42 40 (C3's width) or 1.116 40 (C4's)
"""
import sys
import os
import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.glims_oracle import OracleTumorGrowth, box_mesh   # noqa: E402


def lanczos_bounds(alphas, betas):
    m = len(alphas)
    T = np.zeros((m, m))
    for k in range(m):
        T[k, k] = 1.0 / alphas[k] + (betas[k] / alphas[k - 1] if k > 0 else 0.0)
        if k + 1 < m:
            T[k, k + 1] = T[k + 1, k] = np.sqrt(betas[k + 1]) / alphas[k]
    ev = np.linalg.eigvalsh(T)
    return ev[0], ev[-1]


def pcg(A, dinv, b, rtol):
    x = np.zeros_like(b)
    r = b.copy()
    u = dinv * r
    p = u.copy()
    g = r @ u
    nb = np.linalg.norm(b)
    al, be, hist = [], [0.0], []
    for it in range(500):
        w = A @ p
        a = g / (p @ w)
        x += a * p
        r -= a * w
        al.append(a)
        hist.append(np.linalg.norm(r) / nb)
        if hist[-1] <= rtol:
            break
        u = dinv * r
        g2 = r @ u
        be.append(g2 / g)
        p = u + (g2 / g) * p
        g = g2
    return x, hist, al, be


def cheb(A, dinv, b, lmin, lmax, m):
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta
    rho = 1.0 / sigma
    r = b.copy()
    d = dinv * r / theta
    x = d.copy()
    nb = np.linalg.norm(b)
    hist = []
    for k in range(1, m + 1):
        r = r - A @ d
        hist.append(np.linalg.norm(r) / nb)
        rho_n = 1.0 / (2.0 * sigma - rho)
        d = rho_n * rho * d + (2.0 * rho_n / delta) * (dinv * r)
        x += d
        rho = rho_n
    return x, hist


def main():
    h = float(sys.argv[1]) if len(sys.argv) > 1 else 2.42
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    ctr = np.array([118.0, -109.0, 72.0])
    L = h * n
    if len(sys.argv) > 3 and sys.argv[3] == 'bl':      # jittered-lattice Delaunay mesh (workloads.brain_like_mesh), scaled to width h
        from glimslib_amd import workloads
        pts, cells = workloads.brain_like_mesh(n ** 3, workers=4)
        h0 = (workloads._BL_EXT.prod() / float(n ** 3)) ** (1.0 / 3.0)
        pts = ctr + (pts - (workloads._BL_ORG + 0.5 * workloads._BL_EXT)) * (h / h0)
    else:
        pts, cells = box_mesh(tuple(ctr - L / 2), tuple(ctr + L / 2), n, n, n)
    mid = pts[cells].mean(axis=1)
    q = ((mid[:, 0] - 120.0) / 80.0) ** 2 + ((mid[:, 1] + 120.0) / 80.0) ** 2 + ((mid[:, 2] - 77.5) / 50.0) ** 2
    wm = q < 1.0
    D = np.where(wm, 0.05, 0.01)
    rho = np.full(len(cells), 0.05)
    z = np.zeros(len(cells))
    o = OracleTumorGrowth(pts, cells, D, rho, z + 0.1, z + 3e-3, z + 0.45, 1.0)
    c = np.exp(-0.5 * ((pts - ctr) ** 2).sum(axis=1))
    print("h = %.3f mm, %d nodes" % (h, len(pts)))
    for step in range(1, 12):
        c_prev = c.copy()
        if step in (2, 10):
            A = sp.csr_matrix(o.rd_jacobian(c))
            b = -o.rd_residual(c, c_prev)
            dinv = 1.0 / A.diagonal()
            for rtol in (3e-4, 1e-5, 1e-7):
                x, hist, al, be = pcg(A, dinv, b, rtol)
                lmin, lmax = lanczos_bounds(al, be)
                ev = None
                line = "step %2d rtol %.0e: PCG %2d its; Ritz [%.3f, %.3f]" % (step, rtol, len(hist), lmin, lmax)
                for (lo, hi, tag) in ((0.9 * lmin, 1.05 * lmax, "Ritz 0.9/1.05"), (0.5 * lmin, 1.1 * lmax, "Ritz 0.5/1.1")):
                    _, hc = cheb(A, dinv, b, lo, hi, 60)
                    need = next((k + 1 for k, v in enumerate(hc) if v <= rtol), None)
                    line += "; Chebyshev(%s) %s its" % (tag, need)
                print(line)
            if step == 10:
                import scipy.sparse.linalg as sla
                Dh = sp.diags(np.sqrt(dinv))
                B = Dh @ A @ Dh
                lo = sla.eigsh(B, k=1, which='SA', return_eigenvectors=False, tol=1e-4)[0]
                hi = sla.eigsh(B, k=1, which='LA', return_eigenvectors=False, tol=1e-4)[0]
                print("   true spectrum of Dinv A: [%.4f, %.4f], kappa %.1f" % (lo, hi, hi / lo))
                for rtol in (3e-4, 1e-5, 1e-7):
                    _, hc = cheb(A, dinv, b, lo, hi, 80)
                    print("   Chebyshev on the true interval, rtol %.0e: %s its" %
                          (rtol, next((k + 1 for k, v in enumerate(hc) if v <= rtol), None)))
        c, _ = o.rd_step(c_prev, rtol=1e-10, linear='cg')


if __name__ == "__main__":
    main()
