#!/usr/bin/env python3
"""one table per tools/sift_ab.sh setting: SIFT kernels, avg us per launch, launches per frame,
us per frame, HBM bytes per frame (2 x FETCH_SIZE + WRITE_SIZE, KiB counters)"""
import os
import re
import sys

name = sys.argv[1]
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
SIFT = ('gray_up2x', 'blur_strip', 'downsample', 'extrema', 'pyramid_tail', 'refine', 'orient',
        'descriptor', 'desc_bucket', 'sort_')
FRAMES = 3                                        # tools/sift_time.py runs three detects


def base(n):
    n = re.sub(r'^(kernel )?(void )?(\(anonymous namespace\)::)?', '', n.strip())
    return re.sub(r'\(.*$', '', n)


t = {}
for line in open(os.path.join(out, 'r4_sift_ab_%s_stats.txt' % name)):
    m = re.match(r'\s+(.*?)\s+calls=(\d+)\s+total_ns=(\d+)', line)
    if m and any(k in m.group(1) for k in SIFT):
        b = base(m.group(1))
        c, ns = t.get(b, (0, 0))
        t[b] = (c + int(m.group(2)), ns + int(m.group(3)))
pmc = {}
for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    cur = None
    if not os.path.exists(os.path.join(out, 'r4_sift_ab_%s_%s.txt' % (name, ctr))):
        continue                                 # (PASSES=stats)
    for line in open(os.path.join(out, 'r4_sift_ab_%s_%s.txt' % (name, ctr))):
        if line.strip().startswith('kernel'):
            cur = base(line)
        m = re.match(r'\s+%s\s+dispatches=(\d+)\s+sum=(\S+)' % ctr, line)
        if m and cur and any(k in cur for k in SIFT):
            pmc.setdefault(cur, {}).setdefault(ctr, 0.0)
            pmc[cur][ctr] += float(m.group(2))
print('%-34s %8s %10s %10s %10s' % (name, 'calls/fr', 'us/launch', 'us/frame', 'MB/frame'))
tot_us = tot_mb = 0.0
for b in sorted(t, key=lambda b: -t[b][1]):
    c, ns = t[b]
    p = pmc.get(b, {})
    mb = (2 * p.get('FETCH_SIZE', 0.0) + p.get('WRITE_SIZE', 0.0)) * 1024 / FRAMES / 1e6
    print('%-34s %8.1f %10.1f %10.1f %10.1f' % (b[:34], c / FRAMES, ns / c / 1e3, ns / FRAMES / 1e3, mb))
    tot_us += ns / FRAMES / 1e3
    tot_mb += mb
print('%-34s %8s %10s %10.1f %10.1f' % ('total', '', '', tot_us, tot_mb))
