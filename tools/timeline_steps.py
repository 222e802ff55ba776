#!/usr/bin/env python3
"""
Kernel time vs dispatch gaps inside the timed steps of a profiled `bench.py` run (rocprofv3 --kernel-trace CSV): the
window from the 30 % mark of the k_cg_update<1> launches to the last one -- i.e. steady-state time steps only, not the
set-up or the post-step roofline launches.  Per kernel: calls, time, calls and microseconds per step; per kernel pair:
the idle gaps between consecutive kernels.

usage: tools/timeline_steps.py <dir with *_kernel_trace.csv>
"""
import collections, csv, glob, re, sys
import numpy as np

f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r'(k_[a-z_0-9]+(<[^>]*>)?)', r['Kernel_Name'])
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1) if m else r['Kernel_Name'][:30]))
rows.sort()
upd = [i for i, r in enumerate(rows) if r[2].startswith('k_cg_update<1>')]
R = rows[upd[int(len(upd) * 0.3)]:upd[-1] + 1]
wall = R[-1][1] - R[0][0]
busy, cnt, gap = collections.defaultdict(float), collections.Counter(), collections.defaultdict(list)
for a, b in zip(R[:-1], R[1:]):
    busy[a[2]] += a[1] - a[0]
    cnt[a[2]] += 1
    gap[(a[2], b[2])].append(b[0] - a[1])
tot = sum(busy.values())
nsteps = max(1, sum(1 for r in R if r[2].startswith('k_ws_delta')))
print("window %.2f ms = %d steps (%.3f ms/step under the profiler): kernels %.2f ms (%.1f %%), idle %.2f ms" %
      (wall / 1e6, nsteps, wall / 1e6 / nsteps, tot / 1e6, 100 * tot / wall, (wall - tot) / 1e6))
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print("  %-42s %5d calls %7.2f ms  avg %7.1f us   per step: %5.1f calls %7.1f us" %
          (k, cnt[k], v / 1e6, v / cnt[k] / 1e3, cnt[k] / nsteps, v / 1e3 / nsteps))
print("idle gaps by kernel pair (total ms, count, median us, us per step):")
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:12]:
    v = np.array(v)
    print("  %-34s -> %-34s %6.2f ms  n=%4d  med %5.1f us  %6.1f us/step" %
          (k[0][:34], k[1][:34], v.sum() / 1e6, len(v), np.median(v) / 1e3, v.sum() / 1e3 / nsteps))
