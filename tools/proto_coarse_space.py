"""
CPU prototype (scipy, on top of tools/proto_gmg.py): WHERE the auxiliary-grid multigrid loses its iterations on general
meshes.  Scalar stiff RD operator S (diffusivities x DSCALE) on a Delaunay mesh of random points ('u') or on the brain-like
quality mesh ('b'):  python tools/proto_coarse_space.py 30000 u|b
Compares, as preconditioners of CG: the hierarchy with trilinear prolongation; an energy-minimised prolongation restricted to
the 8-parent pattern; a fully smoothed (smoothed-aggregation style) prolongation; symmetric Gauss-Seidel smoothing; the TWO-GRID
method with an exact solve of the first grid; W-cycles / heavier smoothing on the Cartesian levels; and the first-grid system
solved by m inner CG iterations preconditioned with the Cartesian V-cycle.  Result (profiles/r04_proto_coarse_space_*.txt):
on random points the two-grid method needs 16-19 iterations where the V-cycle needs 58-64 -- the loss is in the Cartesian
levels (2:1 geometric coarsening of a rough Galerkin operator), not in the mesh -> grid transfer and not in the level-0
smoother; on the quality mesh V-cycle = two-grid = 13-14.
"""
import sys, os, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
HERE = os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import proto_gmg as pg
from glimslib_amd import workloads
from oracle.glims_oracle import OracleTumorGrowth

def build_scalar(points, A, H0=2.0, emin=0, omega=0.6, full_smooth=False, emin_cart=0):
    n=len(points); d=3
    lo=points.min(0); hi=points.max(0)
    h=np.full(d, ((hi-lo).prod()/n)**(1/3))
    H=H0*h
    lo=lo-0.37*h
    nc=[max(1,int(np.ceil((hi[a]-lo[a])/H[a]-1e-9)))+1 for a in range(d)]
    P=pg.trilinear_P(points, lo, H, nc)
    levels=[dict(A=A.tocsr())]
    first=True
    while True:
        Af=levels[-1]['A']
        if (emin and first) or (emin_cart and not first):
            pat=(P!=0).astype(float).tocsr()
            Dinv=sp.diags(1.0/np.where(Af.diagonal()>1e-300,Af.diagonal(),1.0))
            for _ in range(emin if first else emin_cart):
                U=(Dinv@(Af@P)).tocsr()
                if not full_smooth:
                    U=U.multiply(pat).tocsr()
                    # keep row sums: subtract the row mean over the pattern
                    rs=np.asarray(U.sum(1)).ravel(); cnt=np.asarray(pat.sum(1)).ravel()
                    U=U-sp.diags(rs/np.maximum(cnt,1))@pat
                P=(P-omega*U).tocsr()
            print("   emin: P nnz/row %.1f, row-sum err %.2e"%(P.nnz/P.shape[0], np.abs(np.asarray(P.sum(1)).ravel()-1).max()))
        first=False
        Ac=(P.T@Af@P).tocsr()
        dead=Ac.diagonal()<=1e-300
        Ac=Ac+sp.diags(dead.astype(float))
        levels[-1]['P']=P
        levels.append(dict(A=Ac.tocsr()))
        print("  level %d: grid %s, %d dofs, nnz %d"%(len(levels)-1,[m+1 for m in nc],Ac.shape[0],Ac.nnz))
        if max(nc)<=3: break
        nc2=[max(1,(m+1)//2) for m in nc]; H2=[2*x for x in H]
        gp=pg.grid_points(lo,H,nc)
        P=pg.trilinear_P(gp,lo,np.array(H2),nc2)
        nc,H=nc2,np.array(H2)
    for L in levels:
        L['Dinv']=sp.diags(1.0/L['A'].diagonal()).tocsr()
    levels[-1]['lu']=spla.splu(levels[-1]['A'].tocsc())
    for L in levels[:-1]:
        L['lam']=pg.lam_max(L['A'],L['Dinv']); L['omega']=4/(3*L['lam']); L['ratio']=10.0
    return levels


def main():
    n=int(sys.argv[1]) if len(sys.argv)>1 else 30000
    kind=sys.argv[2] if len(sys.argv)>2 else 'u'
    w=workloads.config_unstructured(n) if kind=='u' else workloads.config_brain_like(n, workers=4)
    t=dict(w.tables); t['D']=[float(os.environ.get('DSCALE','2000'))*x for x in t['D']]
    per={k:np.asarray(v)[w.cell_label] for k,v in t.items()}
    o=OracleTumorGrowth(w.mesh.points,w.mesh.cells,per['D'],per['rho'],per['gamma'],per['E'],per['nu'],1.0)
    A=o.S.tocsr(); b=o.M@w.c0
    print(kind, n, 'nodes', A.shape[0])
    _,itj=pg.pcg(A,b,lambda r: r/A.diagonal()); print('Jacobi its',itj)
    for name,kw in (('trilinear',dict()),('emin1 pattern',dict(emin=1)),('emin2 pattern',dict(emin=2)),('emin4 pattern',dict(emin=4)),('smoothed (full) 1',dict(emin=1,full_smooth=True,omega=0.66))):
        t0=time.time(); lv=build_scalar(w.mesh.points,A,**kw)
        for deg in (1,3):
            _,it=pg.pcg(A,b,lambda r: pg.vcycle(lv,0,r,nu=deg,cheb=1))
            print("   %-20s cheb%d: %d its   (%.1fs, complexity %.2f)"%(name,deg,it,time.time()-t0,sum(L['A'].nnz for L in lv)/A.nnz))

    # --- which component limits: smoother or coarse space?  symmetric Gauss-Seidel smoothing on every level
    from scipy.sparse.linalg import spsolve_triangular
    def vcycle_gs(levels,l,r,nu=1):
        L=levels[l]
        if 'lu' in L: return L['lu'].solve(r)
        A=L['A']
        if 'Lo' not in L:
            L['Lo']=sp.tril(A,format='csr'); L['Up']=sp.triu(A,format='csr')
        x=np.zeros_like(r)
        for _ in range(nu):
            x=x+spsolve_triangular(L['Lo'],r-A@x,lower=True)
        rc=L['P'].T@(r-A@x)
        x=x+L['P']@vcycle_gs(levels,l+1,rc,nu)
        for _ in range(nu):
            x=x+spsolve_triangular(L['Up'],r-A@x,lower=False)
        return x
    lv=build_scalar(w.mesh.points,A)
    for nu in (1,2):
        _,it=pg.pcg(A,b,lambda r: vcycle_gs(lv,0,r,nu)); print("   trilinear, symmetric GS(%d): %d its"%(nu,it))
    # two-grid with exact coarse solve and cheb3
    lv2=lv[:2]; lv2=[dict(lv[0]), dict(lv[1])]; lv2[1]['lu']=spla.splu(lv[1]['A'].tocsc())
    _,it=pg.pcg(A,b,lambda r: pg.vcycle(lv2,0,r,nu=3,cheb=1)); print("   two-grid (exact coarse), cheb3: %d its"%it)
    _,it=pg.pcg(A,b,lambda r: vcycle_gs(lv2,0,r,1)); print("   two-grid (exact coarse), SGS(1): %d its"%it)
    for ratio,deg in ((30,3),(30,6),(100,6),(10,6)):
        for L in lv2[:-1]: L['ratio']=float(ratio)
        _,it=pg.pcg(A,b,lambda r: pg.vcycle(lv2,0,r,nu=deg,cheb=1)); print("   two-grid, cheb%d ratio %d: %d its"%(deg,ratio,it))
    print("--- full hierarchy, trilinear P: coarse-level treatment")
    lv=build_scalar(w.mesh.points,A)
    for name,kw in (("V cheb3/3",dict(nu=3,cheb=1)),("V cheb3 fine, cheb6 coarse",dict(nu=3,nuc=6,cheb=1)),("W cheb3/3",dict(nu=3,cheb=1,gamma=2)),("W cheb3 fine cheb2 coarse",dict(nu=3,nuc=2,cheb=1,gamma=2)),("W(3) cheb3/3 gamma=3",dict(nu=3,cheb=1,gamma=3)), ("V cheb3 fine, cheb12 coarse",dict(nu=3,nuc=12,cheb=1))):
        for ratio in (10,30):
            for L in lv[:-1]: L['ratio']=float(ratio)
            _,it=pg.pcg(A,b,lambda r: pg.vcycle(lv,0,r,**kw)); print("   %-30s ratio %d: %d its"%(name,ratio,it))
    print("--- first-grid system solved by m inner CG iterations preconditioned with the Cartesian V-cycle (cheb3)")
    def inner_solve(lv, r, m, deg=3):
        A1=lv[1]['A']
        M=lambda q: pg.vcycle(lv,1,q,nu=deg,cheb=1)
        x=np.zeros_like(r); res=r.copy(); z=M(res); p=z.copy(); rz=res@z
        for it in range(m):
            Ap=A1@p; al=rz/(p@Ap); x+=al*p; res-=al*Ap
            z=M(res); rz2=res@z; p=z+(rz2/rz)*p; rz=rz2
        return x
    def top_cycle(lv, r, m, deg=3):
        L=lv[0]; A0=L['A']
        x=pg.cheb_smooth(L,r,None,deg)
        rc=L['P'].T@(r-A0@x)
        x=x+L['P']@inner_solve(lv,rc,m,deg)
        return pg.cheb_smooth(L,r,x,deg)
    for ratio in (10,30):
        for L in lv[:-1]: L['ratio']=float(ratio)
        for m in (1,2,3,4,6):
            _,it=pg.pcg(A,b,lambda r: top_cycle(lv,r,m)); print("   ratio %d, inner CG its %d: outer %d its"%(ratio,m,it))


if __name__ == '__main__':
    main()
