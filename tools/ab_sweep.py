#!/usr/bin/env python3
"""
A/B of the assembly sweep (glims_apply which = 8), the quadratic-term pass (which = 9) and the Krylov SpMV (which = 5) on
one handle per workload:  python tools/ab_sweep.py [bl|c3|c4|u ...]
GLIMS_DEV_VARIANTS="0,1,..." selects kernel variants (development knob, read by the library per launch); results of every
variant are compared with variant 0's.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads                      # noqa: E402
from glimslib_amd._backend import Handle                # noqa: E402


def main():
    names = sys.argv[1:] or ["bl"]
    variants = [int(v) for v in os.environ.get("GLIMS_DEV_VARIANTS", "0").split(",")]
    reps = int(os.environ.get("REPS", "20"))
    for name in names:
        t0 = time.perf_counter()
        w = workloads.by_name(name)
        h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
        t = w.tables
        h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
        h.set_options(dt=w.dt)
        h.setup(False)
        h.set_state(w.c0)
        st = h.stats()
        print("%s: %d rows, nnz %d (padded %d), incidences %d  [%.1f s]" %
              (w.name, st['n_rows'], st['nnz'], st['nnz_padded'], st['n_corners'], time.perf_counter() - t0), flush=True)
        x = np.random.default_rng(0).random(h.n_nodes)
        alg = {8: 12 * st['n_corners'] + 20 * st['nnz'] + 32 * st['n_rows'],
               9: 8 * st['n_corners'] + 4 * st['nnz'] + 24 * st['n_rows'],
               5: 12 * st['nnz'] + 20 * st['n_rows']}
        ref = {}
        for rnd in range(2):
            for v in variants:
                os.environ["GLIMS_DEV_VARIANT"] = str(v)
                for which in (8, 9, 5):
                    h.apply(which, x, reps=3)
                    y1 = h.apply(which, x, reps=1)[0]
                    _, ms = h.apply(which, x, reps=reps)
                    us = 1e3 * ms / reps
                    if v == variants[0] and rnd == 0:
                        ref[which] = y1
                    d = np.abs(y1 - ref[which]).max() / max(1e-300, np.abs(ref[which]).max())
                    print("  round %d variant %2d  which %d: %8.1f us  %6.0f GB/s (%.3f of 8 TB/s)  max diff vs variant %d: %.1e" %
                          (rnd, v, which, us, alg[which] / us / 1e3, alg[which] / us / 1e3 / 8000.0, variants[0], d), flush=True)
        h.close()


if __name__ == "__main__":
    main()
