"""ctypes view of oracle/libglims_oracle_c.so (TEST INFRASTRUCTURE ONLY -- see glims_oracle_c.c)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _usable_cpus(cap=16):
    """CPUs this process may really use: affinity mask and cgroup-v2 quota, at most `cap`."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libglims_oracle_c.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE])
        L = C.CDLL(path)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.oc_create.restype = C.c_void_p
        L.oc_create.argtypes = [C.c_int, C.c_int64, C.c_int64, dp, ip, dp, dp, C.c_double]
        L.oc_destroy.argtypes = [C.c_void_p]
        L.oc_step.argtypes = [C.c_void_p, dp, C.c_int, C.c_double, C.c_double, C.c_double, dp]
        L.oc_apply.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.oc_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.oc_nnz.restype = C.c_int64
        L.oc_nnz.argtypes = [C.c_void_p]
        L.oc_set_threads.argtypes = [C.c_int]
        # a GPU box gives one job a share of ~16 cores although it reports many more; oversubscribed OpenMP barriers
        # are catastrophic for the small configs, so cap the team (GLIMS_ORACLE_THREADS overrides)
        n = int(os.environ.get("GLIMS_ORACLE_THREADS", _usable_cpus()))
        L.oc_set_threads(n)
        _lib = L
    return _lib


class COracle:
    def __init__(self, points, cells, D, rho, dt):
        L = lib()
        self.pts = np.ascontiguousarray(points, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        m = len(self.cells)
        self.D = np.ascontiguousarray(np.broadcast_to(np.asarray(D, dtype=np.float64), (m,)))
        self.rho = np.ascontiguousarray(np.broadcast_to(np.asarray(rho, dtype=np.float64), (m,)))
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        self.n = len(self.pts)
        self.h = L.oc_create(self.pts.shape[1], self.n, m, self.pts.ctypes.data_as(dp), self.cells.ctypes.data_as(ip),
                             self.D.ctypes.data_as(dp), self.rho.ctypes.data_as(dp), float(dt))

    def step(self, c, n_steps=1, rtol=1e-10, atol=1e-13, cg_rtol=1e-3, load=None):
        c = np.array(c, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        ld = None if load is None else np.ascontiguousarray(load, dtype=np.float64)
        st = lib().oc_step(self.h, c.ctypes.data_as(dp), int(n_steps), rtol, atol, cg_rtol,
                           None if ld is None else ld.ctypes.data_as(dp))
        if st != 0:
            raise RuntimeError("C oracle: Newton did not converge")
        return c

    def apply(self, which, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.n)
        dp = C.POINTER(C.c_double)
        lib().oc_apply(self.h, which, x.ctypes.data_as(dp), y.ctypes.data_as(dp))
        return y

    def stats(self):
        a = (C.c_int64 * 3)()
        lib().oc_stats(self.h, a)
        return dict(newton_its=a[0], cg_its=a[1], sweeps=a[2])

    @staticmethod
    def threads():
        return lib().oc_threads()

    def close(self):
        if self.h:
            lib().oc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
