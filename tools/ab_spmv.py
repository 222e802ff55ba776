#!/usr/bin/env python3
"""A/B of SpMV variants, interleaved rounds in ONE process (cdna_hip_programming.md section 5.4 rule 24)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads
from glimslib_amd._backend import Handle

w = workloads.by_name(sys.argv[1] if len(sys.argv) > 1 else 'c4')
h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
t = w.tables
h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
h.set_options(dt=w.dt)
h.setup(False)
st = h.stats()
b_alg = workloads.b_spmv_bytes(st['nnz'], st['n_rows'])
x = np.random.default_rng(0).standard_normal(h.n_nodes)
variants = [("u4 remap", dict(GLIMS_SPMV_UNROLL="4", GLIMS_XCD_REMAP="1", GLIMS_SPMV_NT="0")),
            ("u4", dict(GLIMS_SPMV_UNROLL="4", GLIMS_XCD_REMAP="0", GLIMS_SPMV_NT="0")),
            ("u8", dict(GLIMS_SPMV_UNROLL="8", GLIMS_XCD_REMAP="0", GLIMS_SPMV_NT="0")),
            ("u4 nt", dict(GLIMS_SPMV_UNROLL="4", GLIMS_XCD_REMAP="0", GLIMS_SPMV_NT="1")),
            ("u8 nt", dict(GLIMS_SPMV_UNROLL="8", GLIMS_XCD_REMAP="0", GLIMS_SPMV_NT="1"))]
res = {n: [] for n, _ in variants}
for rnd in range(6):
    for name, env in variants:
        os.environ.update(env)
        _, ms = h.apply(0, x, reps=30)
        if rnd > 0:
            res[name].append(ms / 30 * 1e3)
for name, v in res.items():
    v = np.array(v)
    print("%-18s median %7.1f us  min %7.1f us   -> %6.0f GB/s (median)" % (name, np.median(v), v.min(), b_alg / np.median(v) / 1e3))

# ---- assembly sweep A/B (non-temporal streams on / off) ----
c = w.c0
resr = {"rd nt=0": [], "rd nt=1": []}
import time
for rnd in range(5):
    for name, val in (("rd nt=0", "0"), ("rd nt=1", "1")):
        os.environ["GLIMS_RD_NT"] = val
        h.rd_residual(c, c)
        t0 = time.perf_counter()
        for _ in range(3):
            h.rd_residual(c, c)
        if rnd > 0:
            resr[name].append((time.perf_counter() - t0) / 3 * 1e3)
for name, v in resr.items():
    print("%-10s median %.2f ms per glims_rd_residual call (includes 2 x 80 MB H2D + 1 D2H + mass SpMV)" % (name, np.median(v)))
