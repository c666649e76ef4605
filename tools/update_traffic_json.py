#!/usr/bin/env python3
"""profiles/<tag>_knn2sym_traffic.json (what bench.py quotes as roofline.traffic / mfma_busy) from
the PMC summaries tools/collect_evidence.sh wrote:  python tools/update_traffic_json.py [tag]"""
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r6'
prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')


def counter(fname, name, kernel='knn2sym_kernel'):
    """avg per dispatch of counter `name` in the block of `kernel`"""
    cur = None
    for line in open(os.path.join(prof, fname)):
        if line.strip().startswith('kernel'):
            cur = line
        m = re.match(r'\s+%s\s+dispatches=(\d+)\s+sum=(\S+)\s+avg=(\S+)' % name, line)
        if m and cur and kernel in cur:
            return float(m.group(3)), int(m.group(1)), cur.strip()
    raise SystemExit('%s: no %s for %s' % (fname, name, kernel))


fetch, n, kname = counter('%s_knn2sym_pmc_fetch.txt' % tag, 'FETCH_SIZE')
write, _, _ = counter('%s_knn2sym_pmc_write.txt' % tag, 'WRITE_SIZE')
busy, _, _ = counter('%s_knn2sym_pmc_sq.txt' % tag, 'SQ_VALU_MFMA_BUSY_CYCLES')
gui, _, _ = counter('%s_knn2sym_pmc_sq.txt' % tag, 'GRBM_GUI_ACTIVE')
mops, _, _ = counter('%s_knn2sym_pmc_sq.txt' % tag, 'SQ_INSTS_VALU_MFMA_MOPS_I8')
path = os.path.join(prof, '%s_knn2sym_traffic.json' % tag)
d = json.load(open(path)) if os.path.exists(path) else {
    "workload": "counter passes on configs[1] (500 images, 31 launches of <= 4096 image pairs, both directions each): the same kernel instantiation and launch shape as the 2812-image headline (965 such launches); rocprofv3 --pmc segfaults inside the 2812-image run on this pool's image",
    "fetch_correction": "x2 on gfx950 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies 128-B requests at 64 B)",
    "algorithmic_bytes_per_launch": 4024 * 1179648}
d['kernel'] = re.sub(r'^kernel (void )?\(anonymous namespace\)::', '', kname).split('(')[0] + \
    ' (symmetric sweep, form 2: 1024 query rows per workgroup, one wave per SIMD, direct global->LDS staging, fused DPP butterfly)'
d['FETCH_SIZE_avg_per_launch_KB'] = round(fetch)
d['WRITE_SIZE_avg_per_launch_KB'] = round(write)
d['hbm_bytes_per_launch'] = int(round(fetch) * 1024 * 2 + round(write) * 1024)
d['source'] = ('profiles/%s_knn2sym_pmc_fetch.txt, profiles/%s_knn2sym_pmc_write.txt (separate rocprofv3 --pmc passes of '
               'bench.py --images 500 --steps 1 --warmup 0 --no-ba --no-sift --no-cpu-baseline --verify-pairs 0 --no-e2e --no-survey)' % (tag, tag))
d['mfma_busy'] = round(busy / (gui / 8 * 1024), 4)
d['mfma_busy_source'] = ('profiles/%s_knn2sym_pmc_sq.txt: SQ_VALU_MFMA_BUSY_CYCLES %.4g / (GRBM_GUI_ACTIVE '
                         '%.4g / 8 XCDs x 1024 SIMDs), the same 500-image bench command; the pipe executes '
                         'ONE pass per distance matrix (%.4g MFMA_MOPS_I8 per dispatch = 4096 pairs x '
                         '65536 MFMAs)' % (tag, busy, gui, mops))
# refuse a summary of a kernel the library does not launch (IAMX_EXPECT_KERNEL = iamx_knn2sym_kernel_id(2))
want = os.environ.get('IAMX_EXPECT_KERNEL')
if want and not d['kernel'].startswith(want):
    raise SystemExit('%s: PMC passes measured %s, the library launches %s' % (path, d['kernel'].split(' (')[0], want))
json.dump(d, open(path, 'w'), indent=1)
print(path, d['hbm_bytes_per_launch'], d['mfma_busy'], 'over', n, 'dispatches')
