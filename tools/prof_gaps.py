#!/usr/bin/env python3
"""Inter-kernel gaps from a rocprofv3 --kernel-trace CSV: for consecutive dispatches (by start
time) report, per kernel name, the average idle time on the device BEFORE the kernel starts.
usage: prof_gaps.py <dir> [name-filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for f in sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)):
    rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'])
            for r in csv.DictReader(open(f))]
    rows.sort()
    gap = defaultdict(list)
    dur = defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        if flt in n1:
            gap[n1[:60]].append(s1 - e0)
            dur[n1[:60]].append(e1 - s1)
    print('##', os.path.relpath(f, d))
    for k in sorted(gap, key=lambda k: -sum(dur[k]))[:12]:
        g = sorted(gap[k])
        print('  %-60s n=%-6d dur_avg=%8.1f us  gap_before: avg=%7.1f med=%7.1f p90=%7.1f us' % (
            k, len(g), sum(dur[k]) / len(g) / 1e3, sum(g) / len(g) / 1e3, g[len(g) // 2] / 1e3,
            g[int(len(g) * 0.9)] / 1e3))
    if os.environ.get('GAPS_TOP'):
        big = sorted(((s1 - e0, i, n0[:40], n1[:40]) for i, ((s0, e0, n0), (s1, e1, n1))
                      in enumerate(zip(rows, rows[1:]))), reverse=True)[:int(os.environ['GAPS_TOP'])]
        for g, i, a, b in big:
            print('   gap %9.1f us at #%d  %s -> %s' % (g / 1e3, i, a, b))
