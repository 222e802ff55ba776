#!/usr/bin/env python3
"""
Interleaved A/B of the Krylov iteration and its stream policy at the per-rank size of an 8-GPU run (round-4 review, item 1c):

    python tools/ab_rank_sized.py [workload:size ...]      default: c4:107 c4:215      (c4:107 = 1/8 of config C4, 1.26 M rows)

One handle per workload; every variant = (rd_linear, stream_policy, fp32 Jacobian storage) runs the same W + K steps from the
same initial state, the variants interleaved over ROUNDS rounds, ms per step = wall clock of glims_step(K).  Fields of every
variant are compared with the first one's (the Newton tolerance is the same for all, so they agree to ~1e-10).
"""
import os
import sys
import time

# numpy's BLAS threads busy-wait after a call (one per visible core); on a box whose CPU share is smaller than that the whole
# process is throttled for tens of milliseconds afterwards -- seen as 2-3x slower time steps at 1 M rows.  One thread.
for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "OMP_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads, _backend as B      # noqa: E402

VARIANTS = [
    ("pcg  nt     f64", dict(rd_linear=B.RD_LINEAR_PCG, stream_policy=B.STREAM_NONTEMPORAL), 0),
    ("pcg  cached f64", dict(rd_linear=B.RD_LINEAR_PCG, stream_policy=B.STREAM_CACHED), 0),
    ("cheb nt     f64", dict(rd_linear=B.RD_LINEAR_CHEBYSHEV, stream_policy=B.STREAM_NONTEMPORAL), 0),
    ("cheb cached f64", dict(rd_linear=B.RD_LINEAR_CHEBYSHEV, stream_policy=B.STREAM_CACHED), 0),
    ("pcg  nt     f32", dict(rd_linear=B.RD_LINEAR_PCG, stream_policy=B.STREAM_NONTEMPORAL), B.FLAG_FP32_JACOBIAN),
    ("pcg  cached f32", dict(rd_linear=B.RD_LINEAR_PCG, stream_policy=B.STREAM_CACHED), B.FLAG_FP32_JACOBIAN),
    ("cheb nt     f32", dict(rd_linear=B.RD_LINEAR_CHEBYSHEV, stream_policy=B.STREAM_NONTEMPORAL), B.FLAG_FP32_JACOBIAN),
    ("cheb cached f32", dict(rd_linear=B.RD_LINEAR_CHEBYSHEV, stream_policy=B.STREAM_CACHED), B.FLAG_FP32_JACOBIAN),
]


def main():
    specs = sys.argv[1:] or ["c4:107", "c4:215"]
    W, K = int(os.environ.get("WARMUP", "5")), int(os.environ.get("STEPS", "20"))
    rounds = int(os.environ.get("ROUNDS", "2"))
    only = os.environ.get("ONLY")
    variants = [v for v in VARIANTS if not only or any(tok in v[0] for tok in only.split(","))]
    for spec in specs:
        name, _, size = spec.partition(":")
        w = workloads.by_name(name, int(size) if size else None)   # (bl:<points> = the brain-like unstructured mesh)
        h = B.Handle(w.mesh.points, w.mesh.cells, w.cell_label)
        t = w.tables
        h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
        n = w.mesh.num_vertices()
        print("== %s: %d rows" % (w.name, n), flush=True)
        ref = None
        best = {}
        for rnd in range(rounds):
            for label, opts, flags in variants:
                h.set_options(dt=w.dt, flags=B.FLAG_WARM_START | flags, time_kernels=int(os.environ.get('TIME_KERNELS', '1')), **opts)
                h.setup(False)
                h.set_state(w.c0)
                st = h.step(W)
                h.reset_stats()
                t0 = time.perf_counter()
                st |= h.step(K)
                ms = 1e3 * (time.perf_counter() - t0) / K
                s = h.stats()
                c = h.get_state(want_u=False)[0]
                if ref is None:
                    ref = c
                d = float(np.linalg.norm(c - ref) / np.linalg.norm(ref))
                it_us = (1e3 * s['ms_cheb_steps'] / s['n_cheb_steps']) if s['n_cheb_steps'] else \
                        (1e3 * s['ms_spmv_steps'] / s['n_spmv_steps']) if s['n_spmv_steps'] else 0.0
                best[label] = min(best.get(label, 1e9), ms)
                print("  round %d  %s: %7.3f ms/step  (device %7.3f)  Newton %.2f  Krylov %.2f per step  [cheb solves %d, "
                      "passes %d, fallbacks %d, learn %d, interval %.3f..%.3f]  operator pass %.1f us  nt %d  ws %.0f MB  "
                      "status %d  rel-L2 vs first %.1e" %
                      (rnd, label, ms, s['ms_steps'] / K, s['newton_its'] / K, s['cg_its'] / K, s['cheb_solves'], s['cheb_its'],
                       s['cheb_fallbacks'], s['cheb_learn_solves'], s['cheb_lmin'], s['cheb_lmax'], it_us,
                       s['stream_nontemporal'], s['krylov_working_set'] / 1e6, st, d), flush=True)
        base = best[variants[0][0]]
        for label, _, _ in variants:
            print("  best  %s: %7.3f ms/step = %.3f ns per row and step  (%.2fx the first variant)" %
                  (label, best[label], 1e6 * best[label] / n, base / best[label]), flush=True)
        h.close()


if __name__ == "__main__":
    main()
