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
study = sys.argv[2] if len(sys.argv) > 2 else 'remap'
variants = [("plain", dict(GLIMS_XCD_REMAP="0")), ("eighths", dict(GLIMS_XCD_REMAP="1")),
            ("chunks of 4", dict(GLIMS_XCD_REMAP="4")), ("chunks of 16", dict(GLIMS_XCD_REMAP="16")),
            ("chunks of 64", dict(GLIMS_XCD_REMAP="64")), ("chunks of 256", dict(GLIMS_XCD_REMAP="256"))]
if study == 'idx16':
    variants = [("int32 columns", dict(GLIMS_IDX16="0")), ("16-bit window codes", dict(GLIMS_IDX16="1"))]
    y0 = None
    for name, env in variants:           # both index streams must give bitwise the same product
        os.environ.update(env)
        y, _ = h.apply(0, x, reps=1)
        assert y0 is None or np.array_equal(y, y0), "index streams disagree"
        y0 = y
if study == 'unroll':
    variants = [("unroll 4", dict(GLIMS_SPMV_UNROLL="4")), ("unroll 8", dict(GLIMS_SPMV_UNROLL="8"))]
if study == 'nt':
    variants = [("nt values+columns", dict(GLIMS_SPMV_NT="1")), ("plain loads", dict(GLIMS_SPMV_NT="0")),
                ("nt values only", dict(GLIMS_SPMV_NT="2"))]
which = {}
if study == 'dots':
    variants = [("plain (k_spmv<0,..>)", {}), ("fused dot (k_spmv<1,..>)", {})]
    which = {"fused dot (k_spmv<1,..>)": 5}
if study == 'pairs':
    variants = [("8-B value loads (product kernel)", {}), ("slot pairs, 16-B value loads (study)", {})]
    which = {"slot pairs, 16-B value loads (study)": 6}
    ya, yb = h.apply(0, x)[0], h.apply(6, x)[0]
    print("max |difference| of the two products: %.2e (summation order differs)" % np.abs(ya - yb).max())
res = {n: [] for n, _ in variants}
for rnd in range(9):
    for name, env in variants:
        os.environ.update(env)
        _, ms = h.apply(which.get(name, 0), x, reps=30)
        if rnd > 0:
            res[name].append(ms / 30 * 1e3)
for name, v in res.items():
    v = np.array(v)
    print("%-18s median %7.1f us  min %7.1f us   -> %6.0f GB/s (median)" % (name, np.median(v), v.min(), b_alg / np.median(v) / 1e3))

