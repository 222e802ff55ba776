#!/usr/bin/env python3
"""Per-kernel medians of arbitrary rocprofv3 --pmc counters:  tools/pmc_counters.py <rocprof output dir> [kernel-name filter ...]"""
import collections, csv, glob, re, sys
import numpy as np

d = sys.argv[1]
filt = sys.argv[2:]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/*/*_counter_collection.csv') + glob.glob(d + '/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        m = re.search(r'(k_[a-z_0-9]+(<[^>]*>)?)', name)
        key = m.group(1) if m else name[:40]
        if filt and not any(s in key for s in filt):
            continue
        rows[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(rows):
    print(k)
    for c in sorted(rows[k]):
        v = np.array(rows[k][c])
        print("   %-28s n=%4d  median %.4g  max %.4g" % (c, len(v), np.median(v), v.max()))
