#!/usr/bin/env python3
"""
Per-kernel summary of a rocprofv3 --kernel-trace CSV: calls, total, share, and the median duration of the REAL
launches (launches that the device-side `done` word turned into no-ops are much shorter than 20 % of the longest one
and are left out of the median).

usage: tools/kernel_medians.py <dir with *_kernel_trace.csv>
"""
import collections, csv, glob, re, sys
import numpy as np

f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r'(k_[a-z_0-9]+(<[^>]*>)?)', r['Kernel_Name'])
    dur[m.group(1) if m else r['Kernel_Name'][:32]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in dur.values())
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v, dtype=float)
    real = v[v > 0.2 * v.max()]
    print("%-34s %6d calls %9.1f ms %5.1f%%  real %5d  median(real) %9.1f us" %
          (k, len(v), v.sum() / 1e6, 100 * v.sum() / tot, len(real), np.median(real) / 1e3))
