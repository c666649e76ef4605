#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats / counter collection) into small text
summaries that are committed under profiles/.  usage: prof_summary.py <dir> <out.txt>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, out):
    lines = []
    for f in sorted(glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)):
        lines.append('## %s' % os.path.relpath(f, d))
        rows = list(csv.DictReader(open(f)))
        for r in rows:                       # every kernel of the run (round 2 cut this to 25)
            lines.append('  %-110s calls=%-6s total_ns=%-14s avg_ns=%-12s pct=%s' % (
                r.get('Name', '')[:110], r.get('Calls'), r.get('TotalDurationNs'),
                r.get('AverageNs'), r.get('Percentage')))
    for f in sorted(glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)):
        lines.append('## %s' % os.path.relpath(f, d))
        acc = defaultdict(lambda: defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r.get('Kernel_Name', '')
            acc[k][r.get('Counter_Name')].append(float(r.get('Counter_Value', 0)))
            meta[k] = (r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('SGPR_Count'),
                       r.get('LDS_Block_Size'), r.get('Scratch_Size'), r.get('Workgroup_Size'),
                       r.get('Grid_Size'))
        for k in sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values())):
            lines.append('  kernel %s' % k[:100])
            lines.append('    vgpr/agpr/sgpr/lds/scratch/wg/grid = %s' % (meta[k],))
            for c, v in sorted(acc[k].items()):
                lines.append('    %-28s dispatches=%-5d sum=%-18.6g avg=%.6g' % (
                    c, len(v), sum(v), sum(v) / len(v)))
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
