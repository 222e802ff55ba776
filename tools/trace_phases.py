#!/usr/bin/env python3
"""
Splits a rocprofv3 --kernel-trace CSV into phases separated by idle gaps of more than GAP_MS (default 20 ms: host-side set-up
between the variants of an A/B run) and reports per phase: wall time, kernel-busy time, idle time, launches, and the kernels
that account for most of the busy time with their median duration.

usage: tools/trace_phases.py <dir with *_kernel_trace.csv> [gap_ms] [min_launches]
"""
import collections, csv, glob, re, sys
import numpy as np

d = sys.argv[1]
gap_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
min_launches = int(sys.argv[3]) if len(sys.argv) > 3 else 300
f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r'(k_[a-z_0-9]+(<[^>]*>)?)', r['Kernel_Name'])
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1) if m else r['Kernel_Name'][:40]))
rows.sort()
phases, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - cur[-1][1] > gap_ms * 1e6:
        phases.append(cur)
        cur = []
    cur.append(r)
phases.append(cur)
for i, ph in enumerate(phases):
    if len(ph) < min_launches:
        continue
    wall = ph[-1][1] - ph[0][0]
    busy = collections.defaultdict(list)
    gaps = []
    pe = None
    for s, e, n in ph:
        busy[n].append(e - s)
        if pe is not None:
            gaps.append(max(0, s - pe))
        pe = max(pe or 0, e)
    tb = sum(sum(v) for v in busy.values())
    gaps = np.array(gaps, dtype=float)
    print("phase %d: %d launches, wall %.2f ms, busy %.2f ms (%.0f %%), idle %.2f ms; gap median %.1f us, mean %.1f us, p90 %.1f us" %
          (i, len(ph), wall / 1e6, tb / 1e6, 100.0 * tb / wall, (wall - tb) / 1e6, np.median(gaps) / 1e3, gaps.mean() / 1e3,
           np.percentile(gaps, 90) / 1e3))
    for k, v in sorted(busy.items(), key=lambda kv: -sum(kv[1]))[:6]:
        v = np.array(v, dtype=float)
        print("      %-40s %5d calls %8.2f ms  median %8.1f us" % (k, len(v), v.sum() / 1e6, np.median(v) / 1e3))
