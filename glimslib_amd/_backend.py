"""
ctypes binding of libglimship.so (the only compute backend; there is no CPU fallback).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C glimslib_amd/csrc``.
Signatures mirror ``include/glims_hip.h`` one to one.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libglimship.so")

GLIMS_OK, GLIMS_NOT_CONVERGED, GLIMS_NAN = 0, 1, 2
GLIMS_E_USAGE, GLIMS_E_HIP, GLIMS_E_RCCL, GLIMS_E_NO_DEVICE = -1, -2, -3, -4
GLIMS_UNIQUE_ID_BYTES = 256
FLAG_EXTRAPOLATE_GUESS = 1
FLAG_WARM_START = 2
FLAG_FP32_JACOBIAN = 4
FLAG_MG_FP32_SMOOTHER = 8
FLAG_INT32_COLUMNS = 16
FLAG_MG_FP64_VECTORS = 32
FLAG_MG_WHOLE_GRID = 64
FLAG_FULL_NEWTON = 128
FLAG_FIXED_FORCING = 256
FLAG_MG_NO_LUMPING = 512
PRECOND_BLOCK_JACOBI, PRECOND_MULTIGRID = 0, 1
RD_PRECOND_AUTO, RD_PRECOND_JACOBI, RD_PRECOND_MULTIGRID = 0, 1, 2
RD_LINEAR_AUTO, RD_LINEAR_PCG, RD_LINEAR_CHEBYSHEV = 0, 1, 2
STREAM_AUTO, STREAM_NONTEMPORAL, STREAM_CACHED = 0, 1, 2
ABI_VERSION = 6


class BackendError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libglimship: %s (status %d)" % (msg, code))
        self.code = code


class Options(C.Structure):
    _fields_ = [("dt", C.c_double), ("newton_rtol", C.c_double), ("newton_atol", C.c_double),
                ("newton_maxit", C.c_int), ("cg_rtol", C.c_double), ("cg_atol", C.c_double),
                ("cg_maxit", C.c_int), ("mech_rtol", C.c_double), ("mech_atol", C.c_double),
                ("mech_maxit", C.c_int), ("check_every", C.c_int), ("flags", C.c_int),
                ("mech_precond", C.c_int), ("mech_mixed", C.c_int), ("mech_history", C.c_int),
                ("mg_smooth", C.c_int), ("mg_coarse_nodes", C.c_int), ("mg_h_factor", C.c_double),
                ("mg_cheb_ratio", C.c_double), ("time_kernels", C.c_int),
                ("rd_precond", C.c_int), ("rd_mg_smooth", C.c_int),
                ("rd_linear", C.c_int), ("stream_policy", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("steps", C.c_int64), ("newton_its", C.c_int64), ("rd_assemblies", C.c_int64),
                ("cg_its", C.c_int64), ("mech_solves", C.c_int64), ("mech_cg_its", C.c_int64),
                ("last_newton_res", C.c_double), ("last_cg_res", C.c_double), ("last_mech_res", C.c_double),
                ("ms_steps", C.c_double), ("ms_spmv", C.c_double), ("n_rows", C.c_int64), ("nnz", C.c_int64),
                ("nnz_padded", C.c_int64), ("n_corners", C.c_int64), ("nnz_idx16", C.c_int64),
                ("ms_spmv_steps", C.c_double), ("n_spmv_steps", C.c_int64),
                ("failed_steps", C.c_int64), ("mg_levels", C.c_int64), ("mg_cycles", C.c_int64),
                ("mg_complexity", C.c_double), ("ms_mg_setup", C.c_double), ("ms_mech", C.c_double),
                ("ms_sweep_steps", C.c_double), ("n_sweep_steps", C.c_int64), ("ms_update_steps", C.c_double),
                ("n_update_steps", C.c_int64), ("us_spmv_median", C.c_double), ("us_sweep_median", C.c_double),
                ("us_update_median", C.c_double),
                ("rd_precond_used", C.c_int64), ("rd_stiffness_ratio", C.c_double), ("rd_mg_levels", C.c_int64),
                ("rd_mg_cycles", C.c_int64), ("rd_mg_complexity", C.c_double), ("ms_rd_mg_setup", C.c_double),
                ("ms_mgfine_mech", C.c_double), ("n_mgfine_mech", C.c_int64), ("us_mgfine_median", C.c_double),
                ("ms_spmvb_mech", C.c_double), ("n_spmvb_mech", C.c_int64), ("us_spmvb_median", C.c_double),
                ("rd_quad_updates", C.c_int64), ("ms_quad_steps", C.c_double), ("n_quad_steps", C.c_int64),
                ("us_quad_median", C.c_double),
                ("midpoint_steps", C.c_int64), ("rebase_events", C.c_int64),
                ("halo_exchanges", C.c_int64), ("halo_bytes", C.c_int64), ("ms_exchange", C.c_double),
                ("ms_exchange_exposed", C.c_double), ("allreduces", C.c_int64), ("reduce_transport", C.c_int64),
                ("mg_grid1_bytes", C.c_int64),
                ("halo_exchanges_timed", C.c_int64), ("cheb_solves", C.c_int64), ("cheb_its", C.c_int64),
                ("cheb_fallbacks", C.c_int64), ("cheb_learn_solves", C.c_int64), ("cheb_lmin", C.c_double),
                ("cheb_lmax", C.c_double), ("ms_cheb_steps", C.c_double), ("n_cheb_steps", C.c_int64),
                ("us_cheb_median", C.c_double), ("stream_nontemporal", C.c_int64), ("krylov_working_set", C.c_int64),
                ("mg_box_fraction", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_dp = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_h = C.c_void_p

# name -> (restype, argtypes); every symbol declared in include/glims_hip.h
SIGNATURES = {
    "glims_abi_version": (C.c_int, []),
    "glims_create": (C.c_int, [C.POINTER(_h), C.c_int, C.c_int64, C.c_int64, C.c_int64, _dp, _i32p, _i32p, C.c_int]),
    "glims_destroy": (C.c_int, [_h]),
    "glims_last_error": (C.c_char_p, [_h]),
    "glims_set_materials": (C.c_int, [_h, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "glims_options_default": (C.c_int, [C.POINTER(Options)]),
    "glims_set_options": (C.c_int, [_h, C.POINTER(Options)]),
    "glims_set_dirichlet_u": (C.c_int, [_h, C.c_int64, _i64p, _dp]),
    "glims_set_dirichlet_c": (C.c_int, [_h, C.c_int64, _i64p, _dp]),
    "glims_set_rd_load": (C.c_int, [_h, _dp]),
    "glims_set_mech_load": (C.c_int, [_h, _dp]),
    "glims_setup": (C.c_int, [_h, C.c_int]),
    "glims_set_state": (C.c_int, [_h, _dp, _dp]),
    "glims_get_state": (C.c_int, [_h, _dp, _dp]),
    "glims_step": (C.c_int, [_h, C.c_int]),
    "glims_solve_mechanics": (C.c_int, [_h]),
    "glims_get_stats": (C.c_int, [_h, C.POINTER(Stats)]),
    "glims_reset_stats": (C.c_int, [_h]),
    "glims_apply": (C.c_int, [_h, C.c_int, _dp, _dp, C.c_int, _dp]),
    "glims_rd_residual": (C.c_int, [_h, _dp, _dp, _dp]),
    "glims_get_numbering": (C.c_int, [_h, _i32p]),
    "glims_pattern_checksum": (C.c_int, [_h, C.POINTER(C.c_uint64)]),
    "glims_snapshot_save": (C.c_int, [_h, _i64p]),
    "glims_snapshot_load": (C.c_int, [_h, C.c_int64, _dp]),
    "glims_snapshot_mechanics": (C.c_int, [_h, C.c_int64, _dp]),
    "glims_snapshot_clear": (C.c_int, [_h]),
    "glims_project": (C.c_int, [_h, _dp, _dp, C.c_int, C.c_double]),
    "glims_comm_unique_id": (C.c_int, [C.c_char_p]),
    "glims_comm_init": (C.c_int, [_h, C.c_int, C.c_int, C.c_char_p]),
    "glims_comm_selftest": (C.c_int, [_h]),
    "glims_comm_mailbox": (C.c_int, [_h, C.c_char_p]),
    "glims_comm_mailbox_selftest": (C.c_int, [_h]),
    "glims_set_halo": (C.c_int, [_h, C.c_int, _i32p, _i64p, _i32p, _i64p]),
    "glims_set_mg_frame": (C.c_int, [_h, _dp, _dp]),
    "glims_set_transport": (C.c_int, [_h, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
}

HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, _i64p, C.c_void_p, _i64p, C.c_int, _i32p, C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)

_lib = None


def load_library():
    """Load libglimship.so or fail loudly -- there is no alternative compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "glimslib_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C glimslib_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.glims_abi_version.restype = C.c_int
    if lib.glims_abi_version() != ABI_VERSION:
        raise ImportError("glimslib_amd: %s has ABI version %d, this binding needs %d -- rebuild it (make -C glimslib_amd/csrc)"
                          % (LIB_PATH, lib.glims_abi_version(), ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == shape, (a.shape, shape)
    return a


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


class Handle:
    """Thin OO view of one ``glims_ctx`` (one mesh on one GPU)."""

    def __init__(self, points, cells, cell_label, n_own=None, device=0):
        lib = load_library()
        self.lib = lib
        pts = _f64(points)
        cl = np.ascontiguousarray(cells, dtype=np.int32)
        lab = np.ascontiguousarray(cell_label, dtype=np.int32)
        self.dim = pts.shape[1]
        self.n_nodes = pts.shape[0]
        self.n_own = self.n_nodes if n_own is None else int(n_own)
        self.n_cells = cl.shape[0]
        assert cl.shape[1] == self.dim + 1 and lab.shape == (self.n_cells,)
        h = _h()
        st = lib.glims_create(C.byref(h), self.dim, self.n_nodes, self.n_own, self.n_cells,
                              _ptr(pts, _dp), _ptr(cl, _i32p), _ptr(lab, _i32p), int(device))
        if st != GLIMS_OK:
            raise BackendError(st, (lib.glims_last_error(None) or b"").decode())
        self._h = h
        self.options = Options()
        lib.glims_options_default(C.byref(self.options))

    # -- plumbing --------------------------------------------------------------------------------
    def _check(self, st, allow=()):
        if st != GLIMS_OK and st not in allow:
            raise BackendError(st, (self.lib.glims_last_error(self._h) or b"").decode())
        return st

    def close(self):
        if getattr(self, "_h", None):
            self.lib.glims_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- model data ------------------------------------------------------------------------------
    def set_materials(self, D, rho, gamma, E, nu):
        arrs = [_f64(a) for a in (D, rho, gamma, E, nu)]
        n = len(arrs[0])
        assert all(a.shape == (n,) for a in arrs)
        self._check(self.lib.glims_set_materials(self._h, n, *[_ptr(a, _dp) for a in arrs]))

    def set_options(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.options, k):
                raise KeyError(k)
            setattr(self.options, k, v)
        self._check(self.lib.glims_set_options(self._h, C.byref(self.options)))

    def set_dirichlet_u(self, dofs, values):
        dofs = np.ascontiguousarray(dofs, dtype=np.int64)
        values = _f64(np.broadcast_to(values, dofs.shape))
        self._check(self.lib.glims_set_dirichlet_u(self._h, len(dofs), _ptr(dofs, _i64p), _ptr(values, _dp)))

    def set_dirichlet_c(self, nodes, values):
        nodes = np.ascontiguousarray(nodes, dtype=np.int64)
        values = _f64(np.broadcast_to(values, nodes.shape))
        self._check(self.lib.glims_set_dirichlet_c(self._h, len(nodes), _ptr(nodes, _i64p), _ptr(values, _dp)))

    def set_rd_load(self, f):
        f = None if f is None else _f64(f, (self.n_nodes,))
        self._check(self.lib.glims_set_rd_load(self._h, _ptr(f, _dp)))

    def set_mech_load(self, f):
        f = None if f is None else _f64(np.asarray(f).reshape(-1), (self.n_nodes * self.dim,))
        self._check(self.lib.glims_set_mech_load(self._h, _ptr(f, _dp)))

    def setup(self, with_mechanics=True):
        self._check(self.lib.glims_setup(self._h, 1 if with_mechanics else 0))

    # -- state -----------------------------------------------------------------------------------
    def set_state(self, c, u=None):
        c = _f64(c, (self.n_nodes,))
        u = None if u is None else _f64(np.asarray(u).reshape(-1), (self.n_nodes * self.dim,))
        self._check(self.lib.glims_set_state(self._h, _ptr(c, _dp), _ptr(u, _dp)))

    def get_state(self, want_u=True):
        c = np.empty(self.n_nodes)
        u = np.empty(self.n_nodes * self.dim) if want_u else None
        self._check(self.lib.glims_get_state(self._h, _ptr(c, _dp), _ptr(u, _dp)))
        return c, u

    def step(self, n_steps=1):
        """Returns the status (GLIMS_OK / GLIMS_NOT_CONVERGED / GLIMS_NAN); raises on usage/HIP/RCCL errors."""
        return self._check(self.lib.glims_step(self._h, int(n_steps)), allow=(GLIMS_NOT_CONVERGED, GLIMS_NAN))

    def solve_mechanics(self):
        return self._check(self.lib.glims_solve_mechanics(self._h), allow=(GLIMS_NOT_CONVERGED, GLIMS_NAN))

    def stats(self):
        s = Stats()
        self._check(self.lib.glims_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def reset_stats(self):
        self._check(self.lib.glims_reset_stats(self._h))

    # -- operator hooks --------------------------------------------------------------------------
    def apply(self, which, x, reps=1):
        """which: 0 A(c), 1 S, 2 M, 3 K_el, 4 G.  Returns (y, ms_total)."""
        d = self.dim
        nin = self.n_nodes * (d if which == 3 else 1)
        nout = self.n_nodes * (d if which in (3, 4) else 1)
        x = _f64(np.asarray(x).reshape(-1), (nin,))
        y = np.empty(nout)
        ms = C.c_double(0.0)
        self._check(self.lib.glims_apply(self._h, int(which), _ptr(x, _dp), _ptr(y, _dp), int(reps), C.byref(ms)))
        return y, ms.value

    def rd_residual(self, c, c_prev):
        c = _f64(c, (self.n_nodes,))
        cp = _f64(c_prev, (self.n_nodes,))
        R = np.empty(self.n_nodes)
        self._check(self.lib.glims_rd_residual(self._h, _ptr(c, _dp), _ptr(cp, _dp), _ptr(R, _dp)))
        return R

    def numbering(self):
        """old2new: internal index of every node of the caller's numbering."""
        a = np.empty(self.n_nodes, dtype=np.int32)
        self._check(self.lib.glims_get_numbering(self._h, _ptr(a, _i32p)))
        return a

    def pattern_checksum(self):
        """Thirteen 64-bit hashes of the device-resident discretisation structures (see glims_pattern_checksum)."""
        a = (C.c_uint64 * 13)()
        self._check(self.lib.glims_pattern_checksum(self._h, a))
        return [int(v) for v in a]

    def snapshot_save(self):
        sid = C.c_int64(-1)
        self._check(self.lib.glims_snapshot_save(self._h, C.byref(sid)))
        return int(sid.value)

    def snapshot_load(self, sid):
        c = np.empty(self.n_nodes)
        self._check(self.lib.glims_snapshot_load(self._h, int(sid), _ptr(c, _dp)))
        return c

    def snapshot_mechanics(self, sid):
        u = np.empty(self.n_nodes * self.dim)
        st = self._check(self.lib.glims_snapshot_mechanics(self._h, int(sid), _ptr(u, _dp)),
                         allow=(GLIMS_NOT_CONVERGED, GLIMS_NAN))
        return u, st

    def snapshot_clear(self):
        self._check(self.lib.glims_snapshot_clear(self._h))

    def project(self, rhs, rtol=1e-12):
        """Solve M x = rhs for rhs [n_nodes] or [n_nodes, k] (L2 projection onto P1 given integrated loads)."""
        rhs = np.asarray(rhs, dtype=np.float64)
        k = 1 if rhs.ndim == 1 else rhs.shape[1]
        r = _f64(rhs.reshape(self.n_nodes, k))
        x = np.empty_like(r)
        self._check(self.lib.glims_project(self._h, _ptr(r, _dp), _ptr(x, _dp), int(k), float(rtol)))
        return x.reshape(rhs.shape)

    # -- multi-GPU -------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        lib = load_library()
        buf = C.create_string_buffer(GLIMS_UNIQUE_ID_BYTES)
        st = lib.glims_comm_unique_id(buf)
        if st != GLIMS_OK:
            raise BackendError(st, "ncclGetUniqueId failed")
        return buf.raw

    def comm_init(self, rank, world, uid):
        assert len(uid) == GLIMS_UNIQUE_ID_BYTES
        self._check(self.lib.glims_comm_init(self._h, int(rank), int(world), uid))

    def comm_selftest(self):
        self._check(self.lib.glims_comm_selftest(self._h))

    def comm_mailbox(self, shm_name):
        """Node-local all-reduce through the POSIX shm object `shm_name` ('/...'); None switches it off."""
        self._check(self.lib.glims_comm_mailbox(self._h, None if shm_name is None else shm_name.encode()))

    def comm_mailbox_selftest(self):
        self._check(self.lib.glims_comm_mailbox_selftest(self._h))

    def set_transport(self, rank, world, halo_cb, allreduce_cb):
        """halo_cb / allreduce_cb: HALO_FN / ALLREDUCE_FN instances (kept alive by this handle)."""
        self._transport_refs = (halo_cb, allreduce_cb)
        self._check(self.lib.glims_set_transport(self._h, int(rank), int(world), C.cast(halo_cb, C.c_void_p),
                                                 C.cast(allreduce_cb, C.c_void_p), None))

    def set_mg_frame(self, lo, hi):
        """Bounding box of the WHOLE (global) mesh: the elasticity multigrid of a partitioned run gets replicated coarse
        levels on one global grid frame (see glims_set_mg_frame)."""
        lo, hi = _f64(lo, (self.dim,)), _f64(hi, (self.dim,))
        self._check(self.lib.glims_set_mg_frame(self._h, _ptr(lo, _dp), _ptr(hi, _dp)))

    def set_halo(self, peer_rank, send_ptr, send_idx, recv_count):
        pr = np.ascontiguousarray(peer_rank, dtype=np.int32)
        sp = np.ascontiguousarray(send_ptr, dtype=np.int64)
        si = np.ascontiguousarray(send_idx, dtype=np.int32)
        rc = np.ascontiguousarray(recv_count, dtype=np.int64)
        assert len(sp) == len(pr) + 1 and len(rc) == len(pr)
        self._check(self.lib.glims_set_halo(self._h, len(pr), _ptr(pr, _i32p), _ptr(sp, _i64p), _ptr(si, _i32p),
                                            _ptr(rc, _i64p)))
