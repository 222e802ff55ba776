import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from glimslib_amd import _backend
from glimslib_amd.mesh import RectangleMesh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
mesh = RectangleMesh((-5.0, -5.0), (5.0, 5.0), n, n)
lv = np.where(mesh.points[:, 0] >= 0.0, 1.0, 2.0)[mesh.cells]
label = lv.mean(axis=1).astype(np.int64).astype(np.int32)
contrast = os.environ.get("CONTRAST", "c1")
tabs = dict(D=[0.0, 0.1, 0.0], rho=[0.0, 0.1, 0.0], gamma=[0.0, 0.2, 0.0],
            E=[10e6, 0.001, 0.001] if contrast == "c1" else [1.0, 0.001, 0.003], nu=[0.49, 0.40, 0.10])
if contrast == "jump":      # stiff / soft halves
    tabs['E'] = [1.0, 1.0, 1e-4]; tabs['gamma'] = [0.0, 0.2, 0.1]
r = np.hypot(mesh.points[:, 0] - 2.5, mesh.points[:, 1] - 2.5)
c0 = np.exp(-r ** 2 / 0.5)
f = mesh.facets(); bn = np.unique(f['vertices'][f['exterior']])
dofs = (bn[:, None] * 2 + np.arange(2)).ravel()
for pre in ("mg", "bj"):
    h = _backend.Handle(mesh.points, mesh.cells, label)
    h.set_materials(tabs['D'], tabs['rho'], tabs['gamma'], tabs['E'], tabs['nu'])
    h.set_options(dt=1.0, mech_history=0, mech_precond=_backend.PRECOND_MULTIGRID if pre == "mg" else _backend.PRECOND_BLOCK_JACOBI,
                  mech_maxit=20000)
    h.set_dirichlet_u(dofs, np.zeros(len(dofs)))
    h.setup(True); h.set_state(c0)
    t0 = time.perf_counter(); st = h.solve_mechanics(); t1 = time.perf_counter()
    s = h.stats()
    print("2-D %dx%d (%d nodes), contrast %s, %s: status %d, %d PCG its, %.1f ms (set-up %.0f ms, %d levels), res %.2e" %
          (n, n, mesh.num_vertices(), contrast, pre, st, s['mech_cg_its'], 1e3 * (t1 - t0), s['ms_mg_setup'], s['mg_levels'], s['last_mech_res']), flush=True)
    h.close()
