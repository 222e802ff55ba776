#!/usr/bin/env python3
"""
Reads a rocprofv3 --kernel-trace CSV and reports, for the last `frac` of the trace (the timed region of bench.py),
how the wall time splits into kernel execution and idle gaps between consecutive kernels, per kernel name.

usage: tools/timeline_gaps.py <dir with *_kernel_trace.csv> [frac=0.5]
"""
import collections, csv, glob, re, sys
import numpy as np


def main():
    d = sys.argv[1]
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_[a-z_0-9]+(<[^>]*>)?)', r['Kernel_Name'])
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1) if m else r['Kernel_Name'][:32]))
    rows.sort()
    t_end = rows[-1][1]
    t_begin = rows[0][0]
    cut = t_end - frac * (t_end - t_begin)
    rows = [r for r in rows if r[0] >= cut]
    wall = rows[-1][1] - rows[0][0]
    busy = collections.defaultdict(float)
    gap_after = collections.defaultdict(list)
    calls = collections.Counter()
    prev_end = None
    prev_name = None
    for s, e, n in rows:
        busy[n] += e - s
        calls[n] += 1
        if prev_end is not None:
            gap_after[prev_name + ' -> ' + n].append(max(0, s - prev_end))
        prev_end = max(prev_end or 0, e)
        prev_name = n
    tb = sum(busy.values())
    print("window %.1f ms: kernels %.1f ms (%.1f %%), idle %.1f ms" % (wall / 1e6, tb / 1e6, 100 * tb / wall,
                                                                   (wall - tb) / 1e6))
    for n, v in sorted(busy.items(), key=lambda kv: -kv[1]):
        print("  %-34s %6d calls %9.2f ms  avg %8.1f us" % (n, calls[n], v / 1e6, v / calls[n] / 1e3))
    print("gaps (total ms, count, median us):")
    for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1]))[:16]:
        v = np.array(v)
        print("  %-60s %8.2f ms %6d  med %6.1f us  p90 %6.1f us  max %8.1f us" %
              (k, v.sum() / 1e6, len(v), np.median(v) / 1e3, np.percentile(v, 90) / 1e3, v.max() / 1e3))


if __name__ == '__main__':
    main()
