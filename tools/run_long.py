#!/usr/bin/env python3
"""Long run of a BASELINE config in chunks, printing the solver statistics of every chunk (robustness check)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads
from glimslib_amd._backend import Handle

name = sys.argv[1] if len(sys.argv) > 1 else 'c4'
total = int(sys.argv[2]) if len(sys.argv) > 2 else 500
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 10
w = workloads.by_name(name)
h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
t = w.tables
h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
h.set_options(dt=w.dt)
h.setup(False)
h.set_state(w.c0)
done = 0
prev = h.stats()
while done < total:
    st = h.step(min(chunk, total - done))
    s = h.stats()
    n = s['steps'] - prev['steps']
    c = h.get_state(want_u=False)[0] if (st != 0 or (done // chunk) % 10 == 9) else None
    print("steps %4d..%4d status %d  newton/step %.2f (sweeps %.2f, cheap passes %.2f)  cg/step %.2f  |R| %.3e  cg res %.3e  ms/step %.2f%s" %
          (done, done + n, st, (s['newton_its'] - prev['newton_its']) / max(n, 1),
           (s['rd_assemblies'] - prev['rd_assemblies']) / max(n, 1),
           (s['rd_quad_updates'] - prev['rd_quad_updates']) / max(n, 1),
           (s['cg_its'] - prev['cg_its']) / max(n, 1), s['last_newton_res'], s['last_cg_res'],
           (s['ms_steps'] - prev['ms_steps']) / max(n, 1),
           "" if c is None else "  c in [%.3e, %.6f], mass %.6e" % (c.min(), c.max(), c.sum())), flush=True)
    prev = s
    done += max(n, 1) if st == 0 else chunk
    if st != 0:
        print("FAILED with status", st, flush=True)
        break
