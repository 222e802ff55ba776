"""
The reference-side binding of INTEGRATION.md section B, complete: what a glimslib maintainer would add next to
glimslib/simulation/simulation_tumor_growth.py to put libglimship.so behind ``self.solver``.

``HipSolver`` needs nothing but ctypes + numpy and an *adapter* with four array getters; everything that touches
FEniCS objects lives in the adapter:

    mesh_arrays()            -> (xyz [N, d] float64, cells [M, d+1] int32, labels [M] int32)      vertex numbering
    material_tables(n)       -> D, rho, gamma, E, nu   (one value per tissue id < n)
    dirichlet_u()            -> (dofs node*d + a  int64, values float64)
    get_nodal(f) / set_nodal(f, c, u)   mixed Function  <->  vertex-numbered arrays c [N], u [N, d]

``DolfinAdapter`` is the one for glimslib itself (DOLFIN 2017.2 calls; it cannot be executed in this repository --
FEniCS is not installed -- and is kept to the documented DOLFIN API).  ``ShimAdapter`` implements the same four getters
on ``glimslib_amd.fenics_local`` objects; tests/test_gpu_binding_example.py runs ``HipSolver`` through it on the GPU and
compares with the oracle, so the ctypes part -- the part a maintainer would paste -- is exercised as written.

Usage inside ``TumorGrowth._setup_problem`` (simulation_tumor_growth.py:126-130), instead of
``fenics.NonlinearVariationalSolver(problem)``:

    self.solution = fenics.Function(self.functionspace.function_space)
    self.solver = HipSolver(DolfinAdapter(self), u_previous, self.solution, dt=self.params.sim_time_step)
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.environ.get("GLIMSHIP_LIB", os.path.join(HERE, "..", "glimslib_amd", "libglimship.so"))

dp, i32p, i64p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


class Options(C.Structure):      # glims_options, include/glims_hip.h (ABI 6)
    _fields_ = [("dt", C.c_double), ("newton_rtol", C.c_double), ("newton_atol", C.c_double),
                ("newton_maxit", C.c_int), ("cg_rtol", C.c_double), ("cg_atol", C.c_double), ("cg_maxit", C.c_int),
                ("mech_rtol", C.c_double), ("mech_atol", C.c_double), ("mech_maxit", C.c_int),
                ("check_every", C.c_int), ("flags", C.c_int), ("mech_precond", C.c_int), ("mech_mixed", C.c_int),
                ("mech_history", C.c_int), ("mg_smooth", C.c_int), ("mg_coarse_nodes", C.c_int),
                ("mg_h_factor", C.c_double), ("mg_cheb_ratio", C.c_double), ("time_kernels", C.c_int),
                ("rd_precond", C.c_int), ("rd_mg_smooth", C.c_int),
                ("rd_linear", C.c_int), ("stream_policy", C.c_int)]


def _load():
    lib = C.CDLL(LIB)
    h = C.c_void_p
    lib.glims_abi_version.restype = C.c_int
    assert lib.glims_abi_version() == 6, "rebuild libglimship.so: this binding is written for ABI 6"
    lib.glims_create.argtypes = [C.POINTER(h), C.c_int, C.c_int64, C.c_int64, C.c_int64, dp, i32p, i32p, C.c_int]
    lib.glims_destroy.argtypes = [h]
    lib.glims_last_error.restype = C.c_char_p
    lib.glims_last_error.argtypes = [h]
    lib.glims_set_materials.argtypes = [h, C.c_int, dp, dp, dp, dp, dp]
    lib.glims_options_default.argtypes = [C.POINTER(Options)]
    lib.glims_set_options.argtypes = [h, C.POINTER(Options)]
    lib.glims_set_dirichlet_u.argtypes = [h, C.c_int64, i64p, dp]
    lib.glims_setup.argtypes = [h, C.c_int]
    lib.glims_set_state.argtypes = [h, dp, dp]
    lib.glims_get_state.argtypes = [h, dp, dp]
    lib.glims_step.argtypes = [h, C.c_int]
    lib.glims_solve_mechanics.argtypes = [h]
    return lib


def P(a, t):
    return a.ctypes.data_as(t)


class HipSolver:
    """Drop-in for the object stored in ``self.solver`` (simulation_tumor_growth.py:127-130): ``solve()`` advances
    ``solution`` by one implicit step and raises when the nonlinear solve does not converge, like DOLFIN's solver."""

    def __init__(self, adapter, u_previous, solution, dt, device=0):
        self.lib = _load()
        self.adapter, self.solution = adapter, solution
        xyz, cells, labels = adapter.mesh_arrays()
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        self.n, self.d = xyz.shape
        self.h = C.c_void_p()
        st = self.lib.glims_create(C.byref(self.h), self.d, self.n, self.n, len(cells), P(xyz, dp), P(cells, i32p),
                                   P(labels, i32p), int(device))
        if st != 0:
            raise RuntimeError((self.lib.glims_last_error(None) or b"").decode())
        n_lab = int(labels.max()) + 1
        tabs = [np.ascontiguousarray(t, dtype=np.float64) for t in adapter.material_tables(n_lab)]
        self._ok(self.lib.glims_set_materials(self.h, n_lab, *[P(t, dp) for t in tabs]))
        opt = Options()
        self.lib.glims_options_default(C.byref(opt))
        opt.dt = float(dt)                                  # params.sim_time_step (simulation_tumor_growth.py:108)
        self._ok(self.lib.glims_set_options(self.h, C.byref(opt)))
        dofs, vals = adapter.dirichlet_u()
        dofs = np.ascontiguousarray(dofs, dtype=np.int64)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        self._ok(self.lib.glims_set_dirichlet_u(self.h, len(dofs), P(dofs, i64p), P(vals, dp)))
        self._ok(self.lib.glims_setup(self.h, 1))
        c, u = adapter.get_nodal(u_previous)
        c = np.ascontiguousarray(c, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1)
        self._ok(self.lib.glims_set_state(self.h, P(c, dp), P(u, dp)))

    def solve(self):
        st = self.lib.glims_step(self.h, 1)
        if st != 0:      # reference: any exception from solve() -> "Solver did not converge" (simulation_base.py:303-305)
            raise RuntimeError((self.lib.glims_last_error(self.h) or b"").decode() or "libglimship status %d" % st)
        self._ok(self.lib.glims_solve_mechanics(self.h))
        c, u = np.empty(self.n), np.empty(self.n * self.d)
        self._ok(self.lib.glims_get_state(self.h, P(c, dp), P(u, dp)))
        self.adapter.set_nodal(self.solution, c, u.reshape(self.n, self.d))

    def close(self):
        if self.h:
            self.lib.glims_destroy(self.h)
            self.h = C.c_void_p()

    def _ok(self, st):
        if st != 0:
            raise RuntimeError((self.lib.glims_last_error(self.h) or b"").decode() or "libglimship status %d" % st)


def _table(param, n_lab, fenics_constant):
    """Per-tissue table of one model parameter: DiscontinuousScalar.coeffs is indexed by tissue id
    (helper_classes.py:47-58, 564-575); scalars / Constants are uniform."""
    coeffs = getattr(param, 'coeffs', None)
    if coeffs is None:
        return np.full(n_lab, float(param))
    items = coeffs.items() if isinstance(coeffs, dict) else enumerate(coeffs)
    t = np.zeros(n_lab)
    for k, v in items:
        if 0 <= k < n_lab:
            t[k] = float(v.values()[0]) if fenics_constant and hasattr(v, 'values') else float(getattr(v, 'value', v))
    return t


class DolfinAdapter:
    """The four getters on DOLFIN 2017.2 objects, for ``sim`` = a glimslib ``TumorGrowth`` (NOT executed in this
    repository).  W = MixedElement([VectorElement P1, FiniteElement P1]) (simulation_tumor_growth.py:67-72):
    ``vertex_to_dof_map(W)[v * (d + 1) + k]`` is the dof of component k (0..d-1 displacement, d concentration) at vertex v."""

    def __init__(self, sim):
        import dolfin
        self.dolfin, self.sim = dolfin, sim
        self.W = sim.functionspace.function_space
        self.d = sim.mesh.geometry().dim()
        self.v2d = dolfin.vertex_to_dof_map(self.W).reshape(-1, self.d + 1)
        self.d2v = dolfin.dof_to_vertex_map(self.W)

    def mesh_arrays(self):
        m = self.sim.mesh
        return m.coordinates(), m.cells(), self.sim.subdomains.subdomains.array()

    def material_tables(self, n_lab):
        p = self.sim.params
        return [_table(getattr(p, k), n_lab, True) for k in ('diffusion', 'proliferation', 'coupling', 'E', 'poisson')]

    def dirichlet_u(self):
        dofs, vals = [], []
        for bc in self.sim.bcs.dirichlet_bcs:               # list of fenics.DirichletBC (helper_classes.py:705-717)
            for dof, val in bc.get_boundary_values().items():
                v, k = divmod(int(self.d2v[dof]), self.d + 1)
                if k < self.d:                              # displacement component
                    dofs.append(v * self.d + k)
                    vals.append(val)
        return np.asarray(dofs, dtype=np.int64), np.asarray(vals, dtype=np.float64)

    def get_nodal(self, f):
        a = f.vector().get_local()[self.v2d]                # [N, d + 1]
        return a[:, self.d].copy(), a[:, :self.d].copy()

    def set_nodal(self, f, c, u):
        vec = f.vector().get_local()
        vec[self.v2d[:, :self.d]] = u
        vec[self.v2d[:, self.d]] = c
        f.vector().set_local(vec)
        f.vector().apply('insert')


class ShimAdapter:
    """The same four getters on glimslib_amd.fenics_local objects (``sim`` = glimslib_amd.simulation.TumorGrowth after
    setup_global_parameters / setup_model_parameters)."""

    def __init__(self, sim):
        self.sim = sim
        self.d = sim.mesh.geometry().dim()

    def mesh_arrays(self):
        m = self.sim.mesh
        return m.coordinates(), m.cells, self.sim.subdomains.subdomains.array()

    def material_tables(self, n_lab):
        p = self.sim.params
        return [_table(getattr(p, k), n_lab, False) for k in ('diffusion', 'proliferation', 'coupling', 'E', 'poisson')]

    def dirichlet_u(self):
        return self.sim.bcs.dirichlet_dofs(0)

    def get_nodal(self, f):
        return f.components[1].copy(), f.components[0].copy()

    def set_nodal(self, f, c, u):
        f.components[1] = c
        f.components[0] = u
