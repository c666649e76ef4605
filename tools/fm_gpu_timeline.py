#!/usr/bin/env python3
"""Device timeline of a find_matches run from a rocprofv3 --kernel-trace CSV directory: per sweep
launch (one round) its start, duration, the idle time of the device before it and what ran in
between; then totals.      python tools/fm_gpu_timeline.py <dir> [every]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
every = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
sweeps = [i for i, r in enumerate(rows) if 'knn2sym_kernel' in r[2]]
t0 = rows[sweeps[0]][0]
busy = idle = 0.0
print('round  start ms   sweep ms  other kernels ms  device idle ms before this sweep')
for n, i in enumerate(sweeps):
    prev = sweeps[n - 1] if n else None
    other = gap = 0.0
    if prev is not None:
        end = rows[prev][1]
        for s, e, _k in rows[prev + 1:i]:
            other += (e - s) / 1e6
            gap += max(0, s - end) / 1e6
            end = max(end, e)
        gap += max(0, rows[i][0] - end) / 1e6
    busy += (rows[i][1] - rows[i][0]) / 1e6 + other
    idle += gap
    if n % every == 0 or n == len(sweeps) - 1:
        print('%5d %9.1f %9.2f %12.2f %12.2f' % (n, (rows[i][0] - t0) / 1e6, (rows[i][1] - rows[i][0]) / 1e6, other, gap))
print('rounds %d: device busy %.2f s, idle between kernels %.2f s, span %.2f s'
      % (len(sweeps), busy / 1e3, idle / 1e3, (rows[sweeps[-1]][1] - t0) / 1e9))
