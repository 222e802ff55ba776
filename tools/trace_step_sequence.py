#!/usr/bin/env python3
"""
Prints the launch sequence of ONE time step from a rocprofv3 --kernel-trace CSV: every launch with its duration and the idle gap
before it.  Steps are delimited by a marker kernel that runs once per step (default k_ws_delta: the warm start's increment).

usage: tools/trace_step_sequence.py <dir with *_kernel_trace.csv> [marker] [which step, counted from the end, default 5]
Also prints, over the last 15 steps: wall per step, kernel-busy per step, launches per step, and the busy time by kernel.
"""
import collections, csv, glob, re, sys
import numpy as np

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else 'k_ws_delta'
which = int(sys.argv[3]) if len(sys.argv) > 3 else 5
f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r'(k_[a-z_0-9]+(<[^>]*>)?)', r['Kernel_Name'])
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1) if m else r['Kernel_Name'][:40]))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].startswith(marker)]
if len(marks) < which + 2:
    sys.exit("fewer than %d marker launches (%s)" % (which + 2, marker))
a, b = marks[-which - 1], marks[-which]
print("step between launches %d and %d of %d: %d launches, wall %.1f us" % (a, b, len(rows), b - a, (rows[b][0] - rows[a][0]) / 1e3))
pe = rows[a - 1][1] if a > 0 else rows[a][0]
for s, e, n in rows[a:b]:
    print("   gap %7.1f us   %-44s %8.1f us" % ((s - pe) / 1e3, n, (e - s) / 1e3))
    pe = max(pe, e)
last = marks[-16:] if len(marks) >= 16 else marks
walls, busys, counts = [], [], []
by = collections.defaultdict(float)
for a, b in zip(last[:-1], last[1:]):
    walls.append((rows[b][0] - rows[a][0]) / 1e3)
    busys.append(sum(e - s for s, e, _ in rows[a:b]) / 1e3)
    counts.append(b - a)
    for s, e, n in rows[a:b]:
        by[n] += (e - s) / 1e3
print("last %d steps: wall %.1f us per step (median %.1f), kernel-busy %.1f us, launches %.1f" %
      (len(walls), np.mean(walls), np.median(walls), np.mean(busys), np.mean(counts)))
for n, t in sorted(by.items(), key=lambda kv: -kv[1])[:14]:
    print("      %-44s %8.1f us per step" % (n, t / len(walls)))
