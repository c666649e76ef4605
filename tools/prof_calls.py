#!/usr/bin/env python3
"""Per-launch durations from a rocprofv3 --kernel-trace CSV directory, in launch order:
    prof_calls.py <dir> <kernel name substring> [max rows]
(name, grid, start relative to the first match, duration) -- the --stats averages mix launches
of very different sizes (pyramid octaves, latched no-op iterations)."""
import csv
import glob
import os
import sys


def main(d, pat, limit=200):
    rows = []
    for f in sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)):
        for r in csv.DictReader(open(f)):
            if pat in r.get('Kernel_Name', ''):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                             r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Grid_Size_Y', ''),
                             r.get('Stream_Id', '')))
    rows.sort()
    if not rows:
        print('no launches match', pat)
        return
    t0 = rows[0][0]
    for s, e, name, gx, gy, q in rows[:limit]:
        short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        print('%-34s grid %8s x %-6s stream %-3s start %10.1f us  dur %8.1f us' % (short[:34], gx, gy, q, (s - t0) / 1e3, (e - s) / 1e3))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 200)
