"""
CPU prototype (scipy): polynomial acceleration of the Cartesian levels (an AMLI-type cycle).  On sliver meshes the
auxiliary-grid hierarchy loses its iterations below the first grid (tools/proto_coarse_space.py: two-grid 16, V-cycle 58-64),
and inner CG iterations on the first-grid system recover most of them -- but make the preconditioner nonlinear.  A FIXED
Chebyshev polynomial of degree m in (B A_1), B = the Cartesian V-cycle, is a linear symmetric operator: plain PCG stays valid,
no inner dot products.  Needs the smallest eigenvalue of B A_1 (power iteration on I - B A_1 at set-up).
  python tools/proto_amli.py 30000 u|b
"""
import sys, os, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import proto_gmg as pg
from proto_coarse_space import build_scalar
from glimslib_amd import workloads
from oracle.glims_oracle import OracleTumorGrowth

def spectrum_BA(lv, l, deg, its=12, seed=0):
    """largest eigenvalue of I - B A_l (power iteration) -> lambda_min(B A_l); and lambda_max by power iteration on B A_l"""
    A=lv[l]['A']; rng=np.random.default_rng(seed)
    B=lambda q: pg.vcycle(lv,l,q,nu=deg,cheb=1)
    x=rng.standard_normal(A.shape[0]); x/=np.linalg.norm(x)
    mu=0
    for _ in range(its):
        y=x-B(A@x); mu=x@y; x=y/np.linalg.norm(y)
    x2=rng.standard_normal(A.shape[0]); x2/=np.linalg.norm(x2); la=0
    for _ in range(its):
        y=B(A@x2); la=x2@y; x2=y/np.linalg.norm(y)
    return 1.0-mu, la

def cheb_BA(lv, l, r, m, a, b, deg):
    """m steps of the Chebyshev iteration for A_l x = r, preconditioned by the V-cycle B from level l, spectrum of B A in [a,b]"""
    A=lv[l]['A']; B=lambda q: pg.vcycle(lv,l,q,nu=deg,cheb=1)
    th=0.5*(a+b); de=0.5*(b-a); sg=th/de; rho=1.0/sg
    res=r.copy(); d=B(res)/th; x=d.copy()
    for k in range(1,m):
        res=res-A@d
        z=B(res)
        rho2=1.0/(2*sg-rho)
        d=rho2*rho*d+(2*rho2/de)*z
        rho=rho2
        x=x+d
    return x

def top_cycle(lv, r, m, a, b, deg=3, at=1):
    """V-cycle down to level `at`, whose system is solved by the degree-m polynomial"""
    def rec(l, q):
        if l==at: return cheb_BA(lv,l,q,m,a,b,deg) if m>1 else pg.vcycle(lv,l,q,nu=deg,cheb=1)
        L=lv[l]; x=pg.cheb_smooth(L,q,None,deg)
        x=x+L['P']@rec(l+1, L['P'].T@(q-L['A']@x))
        return pg.cheb_smooth(L,q,x,deg)
    return rec(0,r)

def main():
    n=int(sys.argv[1]) if len(sys.argv)>1 else 30000
    kind=sys.argv[2] if len(sys.argv)>2 else 'u'
    w=workloads.config_unstructured(n) if kind=='u' else workloads.config_brain_like(n, workers=4)
    t=dict(w.tables); t['D']=[float(os.environ.get('DSCALE','2000'))*x for x in t['D']]
    per={k:np.asarray(v)[w.cell_label] for k,v in t.items()}
    o=OracleTumorGrowth(w.mesh.points,w.mesh.cells,per['D'],per['rho'],per['gamma'],per['E'],per['nu'],1.0)
    A=o.S.tocsr(); rhs=o.M@w.c0
    print(kind,n,'nodes',A.shape[0])
    if os.environ.get('EMIN_CART'):
        for k in (1,2,4):
            lv=build_scalar(w.mesh.points,A,emin_cart=k)
            for L in lv[:-1]: L['ratio']=30.0
            _,it=pg.pcg(A,rhs,lambda r: pg.vcycle(lv,0,r,nu=3,cheb=1)); print("energy-minimised P (trilinear pattern, %d sweeps) on the Cartesian levels: V-cycle cheb3: %d its"%(k,it))
        return
    lv=build_scalar(w.mesh.points,A)
    for ratio in (10,30):
        for L in lv[:-1]: L['ratio']=float(ratio)
        _,it=pg.pcg(A,rhs,lambda r: pg.vcycle(lv,0,r,nu=3,cheb=1)); print("ratio %d: plain V-cycle cheb3: %d its"%(ratio,it))
        lmin,lmax=spectrum_BA(lv,1,3)
        print("   spectrum of B A_1 (12 power iterations each): lambda_min <= %.4f, lambda_max >= %.4f"%(lmin,lmax))
        for safety in (1.0,0.5):
            a=max(lmin*safety,1e-3); b=1.05*max(lmax,1.0) if lmax>1 else 1.0
            for m in (2,3,4,6):
                _,it=pg.pcg(A,rhs,lambda r: top_cycle(lv,r,m,a,b)); print("   interval [%.3f, %.2f], degree %d: %d its"%(a,b,m,it))
        # the polynomial one level further down (1/8 of the first grid's nodes)
        lmin2,lmax2=spectrum_BA(lv,2,3)
        print("   spectrum of B A_2: lambda_min <= %.4f, lambda_max >= %.4f"%(lmin2,lmax2))
        for m in (2,3,4,6,10):
            _,it=pg.pcg(A,rhs,lambda r: top_cycle(lv,r,m,max(lmin2,1e-3),1.0,at=2)); print("   polynomial at level 2, degree %d: %d its"%(m,it))

if __name__=='__main__' and not os.environ.get('NESTED'):
    main()


# ---- nested polynomials: polys = {level: (degree, a, b)}; B at a level = smoothing + cycle(level + 1) + smoothing
def cycle(lv, l, r, polys, deg=3):
    L = lv[l]
    if 'lu' in L: return L['lu'].solve(r)
    def B(q):
        x = pg.cheb_smooth(L, q, None, deg)
        x = x + L['P'] @ cycle(lv, l + 1, L['P'].T @ (q - L['A'] @ x), polys, deg)
        return pg.cheb_smooth(L, q, x, deg)
    if l not in polys: return B(r)
    m, a, b = polys[l]
    th = 0.5 * (a + b); de = 0.5 * (b - a); sg = th / de; rho = 1.0 / sg
    res = r.copy(); d = B(res) / th; x = d.copy()
    for k in range(1, m):
        res = res - L['A'] @ d
        z = B(res)
        rho2 = 1.0 / (2 * sg - rho)
        d = rho2 * rho * d + (2 * rho2 / de) * z
        rho = rho2
        x = x + d
    return x

def lam_min_BA(lv, l, polys, its=12, seed=0):
    A = lv[l]['A']; rng = np.random.default_rng(seed)
    x = rng.standard_normal(A.shape[0]); x /= np.linalg.norm(x); mu = 0
    sub = {k: v for k, v in polys.items() if k > l}
    def B(q):
        L = lv[l]
        xx = pg.cheb_smooth(L, q, None, 3)
        xx = xx + L['P'] @ cycle(lv, l + 1, L['P'].T @ (q - L['A'] @ xx), sub)
        return pg.cheb_smooth(L, q, xx, 3)
    for _ in range(its):
        y = x - B(A @ x); mu = x @ y; x = y / np.linalg.norm(y)
    return 1.0 - mu

def main2():
    n = int(sys.argv[1]); kind = sys.argv[2]
    w = workloads.config_unstructured(n) if kind == 'u' else workloads.config_brain_like(n, workers=4)
    t = dict(w.tables); t['D'] = [float(os.environ.get('DSCALE', '2000')) * x for x in t['D']]
    per = {k: np.asarray(v)[w.cell_label] for k, v in t.items()}
    o = OracleTumorGrowth(w.mesh.points, w.mesh.cells, per['D'], per['rho'], per['gamma'], per['E'], per['nu'], 1.0)
    A = o.S.tocsr(); rhs = o.M @ w.c0
    print(kind, n, 'nodes', A.shape[0], flush=True)
    lv = build_scalar(w.mesh.points, A)
    for L in lv[:-1]: L['ratio'] = 30.0
    nl = len(lv)
    _, it = pg.pcg(A, rhs, lambda r: cycle(lv, 0, r, {})); print("plain V-cycle: %d its" % it, flush=True)
    for l in range(1, nl - 1):
        print("   lambda_min(B A_%d), plain V below: %.4f" % (l, lam_min_BA(lv, l, {})), flush=True)
    for name, levels, m in (("polynomial at level 2, degree 3", [2], 3), ("polynomial at level 2, degree 4", [2], 4),
                            ("levels 2 and 3, degree 2", [2, 3], 2), ("levels 2 and 3, degree 3", [2, 3], 3),
                            ("levels 2, 3, 4, degree 2", [2, 3, 4], 2), ("levels 2, 3, 4, degree 3", [2, 3, 4], 3),
                            ("levels 1, 2, 3, degree 2", [1, 2, 3], 2)):
        polys = {}
        for l in sorted([q for q in levels if q < nl - 1], reverse=True):   # bottom-up: each bound measured with the levels below in place
            a = lam_min_BA(lv, l, polys)
            polys[l] = (m, max(0.7 * a, 1e-3), 1.0)
        _, it = pg.pcg(A, rhs, lambda r: cycle(lv, 0, r, polys))
        print("   %-34s: %d its   (lambda_min used: %s)" % (name, it, ", ".join("L%d %.3f" % (l, polys[l][1]) for l in sorted(polys))), flush=True)

if __name__ == '__main__' and os.environ.get('NESTED'):
    main2()
