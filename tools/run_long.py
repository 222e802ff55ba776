#!/usr/bin/env python3
"""Long run of a BASELINE config in chunks, printing the solver statistics of every chunk (robustness check).
    python tools/run_long.py [c4 | c4:107 | c4o:107 | c3 | bl ...] [500] [20] [summary.json]
The JSON summary (steps completed, status, mean / min / max ms per step over the chunks, hash of the library sources) is what
bench.py quotes as config.full_run_* -- commit it as profiles/long_c4_run.json together with the printed log."""
import json, os, sys
for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glimslib_amd import workloads
from glimslib_amd._backend import Handle

name = sys.argv[1] if len(sys.argv) > 1 else 'c4'
total = int(sys.argv[2]) if len(sys.argv) > 2 else 500
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 10
name, _, size = name.partition(':')   # (c4:107 = the same configuration at another size)
w = workloads.by_name(name, int(size)) if size else workloads.by_name(name)
h = Handle(w.mesh.points, w.mesh.cells, w.cell_label)
t = w.tables
h.set_materials(t['D'], t['rho'], t['gamma'], t['E'], t['nu'])
h.set_options(dt=w.dt)
h.setup(False)
h.set_state(w.c0)
done = 0
prev = h.stats()
windows = []
final_status = 0
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
    if n > 0:
        windows.append((n, (s['ms_steps'] - prev['ms_steps']) / n))
    final_status = st
    prev = s
    done += max(n, 1) if st == 0 else chunk
    if st != 0:
        print("FAILED with status", st, flush=True)
        break

if len(sys.argv) > 4:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernels_sha16
    s = h.stats()
    full = [m for k, m in windows if k == chunk]
    tot = sum(k for k, _ in windows)
    json.dump({"workload": w.name, "steps_requested": total, "steps_completed": int(tot), "final_status": int(final_status),
               "chunk": chunk, "ms_per_step_mean": sum(k * m for k, m in windows) / max(1, tot),
               "ms_per_step_min_window": min(full) if full else None, "ms_per_step_max_window": max(full) if full else None,
               "newton_its_per_step": s['newton_its'] / max(1, tot), "krylov_passes_per_step": s['cg_its'] / max(1, tot),
               "chebyshev_fallbacks": int(s['cheb_fallbacks']), "kernels_sha16": kernels_sha16()},
              open(sys.argv[4], "w"), indent=1)
