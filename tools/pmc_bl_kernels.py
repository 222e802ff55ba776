#!/usr/bin/env python3
"""
The Krylov SpMV (glims_apply which = 5: k_spmv<1, UNR, ..> with the fused dot product), the assembly sweep (8) and the
quadratic-term pass (9) of one workload, `REPS` back-to-back launches each -- the program that tools/collect_counters.sh
runs under `rocprofv3 --pmc ...` (program directly after `--`; the brain-like mesh comes from GLIMS_MESH_CACHE, filled by an
un-profiled step first).     python3 tools/pmc_bl_kernels.py [bl|c3|c4:107 ...]
"""
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads                      # noqa: E402
from glimslib_amd._backend import Handle                # noqa: E402

reps = int(os.environ.get("REPS", "10"))
for spec in (sys.argv[1:] or ["bl"]):
    name, _, size = spec.partition(":")
    if name == "bl":
        w = workloads.config_brain_like(int(size or 1000000))
    else:
        w = workloads.by_name(name, int(size) if size else None)
    h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
    t = w.tables
    h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
    h.set_options(dt=w.dt, stream_policy=int(os.environ.get("STREAM_POLICY", "0")))
    h.setup(False)
    h.set_state(w.c0)
    h.step(2)
    x = np.random.default_rng(0).random(h.n_nodes)
    for which in (5, 8, 9):
        h.apply(which, x, reps=reps)
    st = h.stats()
    print("%s: rows %d nnz %d padded %d incidences %d" % (w.name, st['n_rows'], st['nnz'], st['nnz_padded'], st['n_corners']))
    h.close()
