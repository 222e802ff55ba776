"""
CPU prototype (scipy) of the multilevel preconditioner for K_el that csrc/mg.hip implements: geometric multigrid on
auxiliary Cartesian grids -- trilinear interpolation from a Cartesian grid of width H ~ 2h onto the (unstructured) mesh
nodes, Galerkin coarse operators, 2:1 coarsening between the Cartesian levels, damped block-Jacobi smoothing.
Prints PCG iteration counts (rtol 1e-10) against plain block-Jacobi.   python tools/proto_gmg.py [n ...]
"""
import sys, os, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import glims_oracle as go
from glimslib_amd import workloads


def interp1d(x, lo, H, nc):
    """rows: points x, cols: grid nodes 0..nc (nc cells); linear hat weights"""
    t = (x - lo) / H
    i0 = np.clip(np.floor(t).astype(np.int64), 0, nc - 1)
    w1 = t - i0
    w1 = np.where(np.abs(w1) < 1e-7, 0.0, np.where(np.abs(w1 - 1.0) < 1e-7, 1.0, w1))   # snap lattice-aligned nodes
    return i0, 1.0 - w1, w1


def trilinear_P(points, lo, H, nc):
    """scalar prolongation [n_points, prod(nc+1)]"""
    d = points.shape[1]
    n = len(points)
    idx = [interp1d(points[:, a], lo[a], H[a], nc[a]) for a in range(d)]
    rows, cols, vals = [], [], []
    dims = [m + 1 for m in nc]
    for corner in range(2 ** d):
        w = np.ones(n)
        lin = np.zeros(n, dtype=np.int64)
        for a in range(d):
            bit = (corner >> a) & 1
            i0, w0, w1 = idx[a]
            w = w * (w1 if bit else w0)
            stride = int(np.prod(dims[:a]))
            lin = lin + (i0 + bit) * stride
        rows.append(np.arange(n)); cols.append(lin); vals.append(w)
    P = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(n, int(np.prod(dims)))).tocsr()
    P.eliminate_zeros()
    return P


def grid_points(lo, H, nc):
    axes = [lo[a] + H[a] * np.arange(nc[a] + 1) for a in range(len(nc))]
    g = np.meshgrid(*axes, indexing='ij')
    # x fastest
    return np.stack([gi.transpose(*reversed(range(len(nc)))).ravel() for gi in g], axis=1)


def block_diag_inv(A, d):
    n = A.shape[0] // d
    Ab = sp.bsr_matrix(A, blocksize=(d, d))
    Ab.sort_indices()
    D = np.zeros((n, d, d))
    for i in range(n):
        for q in range(Ab.indptr[i], Ab.indptr[i + 1]):
            if Ab.indices[q] == i:
                D[i] = Ab.data[q]
    for i in range(n):
        if np.abs(D[i]).sum() == 0:
            D[i] = np.eye(d)
    Dinv = np.linalg.inv(D)
    return sp.bsr_matrix((Dinv, np.arange(n), np.arange(n + 1)), shape=A.shape).tocsr()


def build(points, K, free, d, H0_factor=2.0, aligned=True, nmin=3, verbose=True):
    """returns levels: list of dict(A, Dinv, P) with P mapping level l+1 -> l"""
    n = len(points)
    lo = points.min(axis=0); hi = points.max(axis=0)
    # mesh width estimate per axis
    if aligned:
        h = np.array([np.min(np.diff(np.unique(np.round(points[:, a], 9)))) for a in range(d)])
    else:
        h = np.full(d, ((hi - lo).prod() / n) ** (1.0 / d))
    H = H0_factor * h
    nc = [max(1, int(np.ceil((hi[a] - lo[a]) / H[a] - 1e-9))) for a in range(d)]
    if not aligned:
        lo = lo - 0.37 * h     # deliberately misaligned
        nc = [m + 1 for m in nc]
    Ps = trilinear_P(points, lo, H, nc)
    I = sp.identity(d, format='csr')
    Fm = sp.diags(free.astype(float))
    P = Fm @ sp.kron(Ps, I, format='csr')
    A = (Fm @ K @ Fm + sp.diags((~free).astype(float))).tocsr()
    levels = [dict(A=A)]
    while True:
        Ac = (P.T @ levels[-1]['A'] @ P).tocsr()
        # inactive coarse dofs -> identity
        dg = Ac.diagonal()
        dead = dg <= 1e-300
        Ac = Ac + sp.diags(dead.astype(float))
        levels[-1]['P'] = P
        levels.append(dict(A=Ac.tocsr()))
        if verbose:
            print("  level %d: grid %s, %d dofs (%d active), nnz %d" % (len(levels) - 1, [m + 1 for m in nc], Ac.shape[0], (~dead).sum(), Ac.nnz))
        if max(nc) <= nmin:
            break
        # 2:1 coarsening
        nc2 = [max(1, (m + 1) // 2) for m in nc]
        H2 = [2 * Hh for Hh in H]
        gp = grid_points(lo, H, nc)
        Ps = trilinear_P(gp, lo, np.array(H2), nc2)
        P = sp.kron(Ps, I, format='csr')
        nc, H = nc2, np.array(H2)
    for L in levels:
        L['Dinv'] = block_diag_inv(L['A'], d)
    levels[-1]['lu'] = spla.splu(levels[-1]['A'].tocsc())
    return levels


def lam_max(A, Dinv, its=20):
    rng = np.random.default_rng(1)
    x = rng.standard_normal(A.shape[0])
    lam = 1.0
    for _ in range(its):
        y = Dinv @ (A @ x)
        lam = np.linalg.norm(y) / np.linalg.norm(x)
        x = y / np.linalg.norm(y)
    return lam


def cheb_smooth(L, r, x, deg):
    # Chebyshev on D^-1 A over [lam/ratio, lam]; ratio: RATIO env (csrc/mg.hip: 30 on lattice meshes, 10 on general ones;
    # the first version used 4: 31 / 25 iterations with degree 2 / 3 at n = 24 against 24 / 17 with 30)
    A, Dinv = L['A'], L['Dinv']
    lmax = 1.1 * L['lam']; lmin = lmax / L.get('ratio', float(os.environ.get('RATIO', '30')))
    theta = 0.5 * (lmax + lmin); delta = 0.5 * (lmax - lmin)
    sigma = theta / delta; rho = 1.0 / sigma
    res = r - A @ x if x is not None else r
    dvec = (Dinv @ res) / theta
    x = dvec if x is None else x + dvec
    for _ in range(deg - 1):
        res = r - A @ x
        rho_new = 1.0 / (2.0 * sigma - rho)
        dvec = rho_new * rho * dvec + (2.0 * rho_new / delta) * (Dinv @ res)
        x = x + dvec
        rho = rho_new
    return x


def vcycle(levels, l, r, nu=1, cheb=0, nuc=None, gamma=1):
    L = levels[l]
    if 'lu' in L:
        return L['lu'].solve(r)
    A, Dinv, P, om = L['A'], L['Dinv'], L['P'], L['omega']
    k = nu if (l == 0 or nuc is None) else nuc
    if cheb:
        x = cheb_smooth(L, r, None, k)
    else:
        x = om * (Dinv @ r)
        for _ in range(k - 1):
            x = x + om * (Dinv @ (r - A @ x))
    for g in range(gamma if l > 0 else 1):
        rc = P.T @ (r - A @ x)
        x = x + P @ vcycle(levels, l + 1, rc, nu, cheb, nuc, gamma)
    if cheb:
        x = cheb_smooth(L, r, x, k)
    else:
        for _ in range(k):
            x = x + om * (Dinv @ (r - A @ x))
    return x


def pcg(A, b, M, rtol=1e-10, maxit=3000):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z
    nb = np.linalg.norm(b)
    for it in range(1, maxit + 1):
        Ap = A @ p
        al = rz / (p @ Ap)
        x += al * p; r -= al * Ap
        if np.linalg.norm(r) <= rtol * nb:
            return x, it
        z = M(r); rz2 = r @ z
        p = z + (rz2 / rz) * p; rz = rz2
    return x, maxit


def run(n, nu_p=0.45, aligned=True, H0=2.0, nsm=1, unstructured=False):
    if unstructured:
        w = workloads.config_unstructured(n, mechanics=True)
    else:
        w = workloads.config_c5(n)
    pts, cells = w.mesh.points, w.mesh.cells
    d = 3
    t = dict(w.tables)
    nu = np.asarray(t['nu'], float).copy()
    if nu_p is not None:
        nu[nu == 0.45] = nu_p
    E = np.asarray(t['E'])[w.cell_label]; nuc = nu[w.cell_label]
    K = go.assemble_elasticity(pts, cells, go.compute_mu(E, nuc), go.compute_lambda(E, nuc))
    G = go.assemble_coupling(pts, cells, go.compute_mu(E, nuc), go.compute_lambda(E, nuc), np.asarray(t['gamma'])[w.cell_label])
    free = np.ones(len(pts) * d, bool)
    dn = np.asarray(w.dirichlet_nodes)
    free[(dn[:, None] * d + np.arange(d)).ravel()] = False
    b = G @ w.c0
    b[~free] = 0
    t0 = time.time()
    levels = build(pts, K, free, d, H0_factor=H0, aligned=aligned and not unstructured, verbose=True)
    for L in levels[:-1]:
        L['lam'] = lam_max(L['A'], L['Dinv'])
        L['omega'] = 4.0 / (3.0 * L['lam'])
    A = levels[0]['A']
    cx = sum(L['A'].nnz for L in levels) / A.nnz
    _, it_bj = pcg(A, b, lambda r: levels[0]['Dinv'] @ r, maxit=1 if os.environ.get('NOBJ') else 3000)
    for (nm, kw) in [("V(1,1) jac", dict(nu=1)), ("V(2,2) jac", dict(nu=2)), ("V(1,1) fine,(3,3) coarse", dict(nu=1, nuc=3)),
                     ("V(1,1) fine, W coarse(2,2)", dict(nu=1, nuc=2, gamma=2)),
                     ("cheb2", dict(nu=2, cheb=1)), ("cheb3", dict(nu=3, cheb=1)), ("cheb1 fine cheb3 coarse", dict(nu=1, nuc=3, cheb=1))]:
        _, it = pcg(A, b, lambda r: vcycle(levels, 0, r, **kw))
        print("     %-32s %d its" % (nm, it))
    _, it_mg = pcg(A, b, lambda r: vcycle(levels, 0, r, nsm))
    print("n=%d nodes=%d nu=%.3f aligned=%s H0=%.1f V(%d,%d): block-Jacobi %d its, MG %d its, op complexity %.2f, omegas %s (%.1fs)" %
          (n, len(pts), nu_p, aligned, H0, nsm, nsm, it_bj, it_mg, cx, ["%.2f" % L['omega'] for L in levels[:-1]], time.time() - t0))


if __name__ == "__main__":
    ns = [int(a) for a in sys.argv[1:]] or [16, 24]
    nu_p = float(os.environ.get('NU', '0.45'))
    for n in ns:
        run(n, nu_p=nu_p, aligned=not os.environ.get('MISALIGN'), H0=float(os.environ.get('H0', '2.0')),
            unstructured=bool(os.environ.get('UNSTRUCT')))
