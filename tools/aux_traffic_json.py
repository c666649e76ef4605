#!/usr/bin/env python3
"""profiles/<tag>_ba_sift_traffic.json -- HBM bytes per launch of the BA and SIFT kernels that
bench.py prices against the HBM roofline, from the two PMC summaries tools/collect_evidence.sh
wrote (<tag>_aux_pmc_fetch.txt / <tag>_aux_pmc_write.txt: separate rocprofv3 --pmc FETCH_SIZE and
--pmc WRITE_SIZE passes of the bench command, counters in KiB; FETCH_SIZE x 2 is the gfx950
correction of MI355X_MICROARCH.md's HBM section).      python tools/aux_traffic_json.py [tag]"""
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r6'
prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
SIFT = ('gray_up2x_kernel', 'blur_strip_kernel', 'downsample_kernel', 'extrema_kernel',
        'extrema_multi_kernel',
        'pyramid_tail_kernel', 'refine_kernel', 'orient_kernel', 'descriptor_kernel',
        'desc_bucket_count_kernel', 'desc_bucket_scan_kernel', 'desc_bucket_scatter_kernel',
        'sort_count_kernel', 'sort_offsets_kernel', 'sort_scatter_kernel', 'sort_rank_kernel',
        'sort_compact_kernel', 'sort_gather_kernel')
BA = ('ba_residual_lds_kernel', 'ba_residual_pipe_kernel', 'ba_residual_jac_kernel', 'acc_cam_kernel', 'acc_pt_kernel',
      'schur_points_kernel', 'schur_fwd_kernel', 'schur_pt_kernel', 'schur_adj_kernel',
      'schur_pq_kernel', 'schur_update1_kernel', 'schur_update2_kernel',
      'schur_calib_reduce_kernel', 'schur_factor_kernel',
      'lsmr_fwd_kernel', 'lsmr_adj_kernel', 'lsmr_update3_kernel')


def table(fname, counter):
    """kernel name (namespace and arguments stripped, template arguments kept) -> (dispatches, sum)"""
    out, cur = {}, None
    for line in open(os.path.join(prof, fname)):
        m = re.match(r'\s+kernel (.*)', line)
        if m:
            cur = re.sub(r'^(void )?(\(anonymous namespace\)::|iamx::)*', '', m.group(1).strip())
            cur = re.sub(r'\(.*$', '', cur)
            continue
        m = re.match(r'\s+%s\s+dispatches=(\d+)\s+sum=(\S+)' % counter, line)
        if m and cur:
            n, s = out.get(cur, (0, 0.0))
            out[cur] = (n + int(m.group(1)), s + float(m.group(2)))
    return out


fetch = table('%s_aux_pmc_fetch.txt' % tag, 'FETCH_SIZE')
write = table('%s_aux_pmc_write.txt' % tag, 'WRITE_SIZE')
kernels = {}
for name in sorted(set(fetch) | set(write)):
    base = name.split('<')[0]
    if base not in SIFT and base not in BA:
        continue
    nf, sf = fetch.get(name, (0, 0.0))
    nw, sw = write.get(name, (0, 0.0))
    n = max(nf, nw)
    kernels[name] = {"dispatches": n, "FETCH_SIZE_KiB_avg": round(sf / max(nf, 1), 2),
                     "WRITE_SIZE_KiB_avg": round(sw / max(nw, 1), 2),
                     "hbm_bytes_per_launch": int((2.0 * sf / max(nf, 1) + sw / max(nw, 1)) * 1024)}
# SIFT: one detect = every kernel of the chain; frames = launches of the once-per-frame tail kernel
frames = sum(v["dispatches"] for k, v in kernels.items() if k.startswith('pyramid_tail_kernel'))
sift_total = sum(v["hbm_bytes_per_launch"] * v["dispatches"] for k, v in kernels.items()
                 if k.split('<')[0] in SIFT)
doc = {"source": "profiles/%s_aux_pmc_fetch.txt, profiles/%s_aux_pmc_write.txt (separate rocprofv3 --pmc "
                 "passes of bench.py --images 64 --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 "
                 "--no-e2e --no-sift-full --no-survey)" % (tag, tag),
       "fetch_correction": "x2 on gfx950 (MI355X_MICROARCH.md, HBM section)",
       "kernels": kernels,
       "sift": {"frames": frames, "hbm_bytes_per_frame": int(sift_total / max(frames, 1))}}
path = os.path.join(prof, '%s_ba_sift_traffic.json' % tag)
json.dump(doc, open(path, 'w'), indent=1)
print(path, len(kernels), 'kernels;', frames, 'SIFT frames,',
      doc["sift"]["hbm_bytes_per_frame"], 'B per frame')
