#!/usr/bin/env python3
"""How selective is the symmetric sweep's bound test on REAL descriptors?  Renders a 2 x 3 survey
of 5472 x 3648 frames, detects, matches every pair through the shipped batch path and prints per
ordered pair: rows, candidates (rows that passed the (L, U) bound test and are re-scanned exactly
by symexact_kernel), survivors of the exact metric test -- next to the synthetic descriptors of
bench.py, where candidates ~= survivors.        python tools/cand_rate.py"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from imageanalysis_amd import image as iimg, kernels, synth  # noqa: E402

cache = os.environ.get('IAMX_CAND_CACHE')            # .npz of the detected descriptors (several runs, one render)
if cache and os.path.exists(cache):
    z = np.load(cache)
    des = [z[k] for k in sorted(z.files)]
else:
    tmp = tempfile.mkdtemp(prefix='iamx_cand_')
    names, truth, logged, K = synth.make_rendered_survey(tmp, 2, 3, device='cuda', **synth.FULL_FRAME)
    des = []
    for n in names:
        bgr = iimg._decode_bgr(os.path.join(tmp, 'images', n + '.JPG'))
        kp, d32, d8 = iimg.features_from_bgr(bgr, 0.4, equalize=True, keep_u8=True)
        des.append(np.ascontiguousarray(d8))
        print(n, len(d8), 'keypoints')
    if cache:
        np.savez(cache, **{'d%02d' % i: d for i, d in enumerate(des)})
store = kernels.DescriptorStore.from_arrays(des)
und = [(a, b) for a in range(len(des)) for b in range(a + 1, len(des))]
ordered = np.array(und + [(b, a) for a, b in und], np.int32)
pb = kernels.PairBatch(store, ordered, sym=True)
ws = kernels.PairWorkspace(pb.rows, pb.n_pairs)
thresh = 270.0 * 0.75
for it in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pb.run(ws, thresh)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
# sweep and filter + exact stage apart (events on the launch stream, 10 repetitions)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tt = [0.0, 0.0]
for it in range(10):
    ev[0].record()
    pb.run_sym_sweep(ws)
    ev[1].record()
    pb.run_sym_filter(ws, thresh)
    ev[2].record()
    torch.cuda.synchronize()
    tt[0] += ev[0].elapsed_time(ev[1]) / 10
    tt[1] += ev[1].elapsed_time(ev[2]) / 10
print('sweep %.3f ms, filter + exact stage %.3f ms (IAMX_EXACT_NARROW=%s IAMX_NARROW_ABL=%s IAMX_NARROW_WPE=%s)'
      % (tt[0], tt[1], os.environ.get('IAMX_EXACT_NARROW', '1'), os.environ.get('IAMX_NARROW_ABL', '0'),
         os.environ.get('IAMX_NARROW_WPE', '3')))
cand = ws.seg_count[:pb.n_pairs].cpu().numpy()
surv = ws.surv_cnt[:pb.n_pairs].cpu().numpy()
rows = np.array([len(des[a]) for a, _b in ordered])
print('%d ordered pairs in %.1f ms' % (len(ordered), dt * 1e3))
for (a, b), r, c, s in zip(ordered.tolist(), rows, cand, surv):
    print('  %d -> %d: rows %6d  candidates %6d (%.1f %%)  survivors %5d' % (a, b, r, c, 100.0 * c / r, s))
print('total: rows %d, candidates %d (%.1f %%), survivors %d' % (rows.sum(), cand.sum(), 100.0 * cand.sum() / rows.sum(), surv.sum()))

# narrow exact stage: classes per candidate (the mask symcand_rows_kernel left; layout of the narrow
# workspace: 256 control bytes, pair table, then one uint64 mask per query row at its original position)
if ws.nar is not None and os.environ.get('IAMX_EXACT_NARROW', '1') != '0':
    up = lambda v: (v + 255) // 256 * 256                                       # noqa: E731
    mask_off = 256 + up(4 * pb.n_pairs)
    masks = ws.nar[mask_off:mask_off + 8 * pb.rows].view(torch.int64).cpu().numpy()
    keep = np.unpackbits(ws.keep[:(pb.rows + 7) // 8].cpu().numpy(), bitorder='little')[:pb.rows].astype(bool)
    sel = np.nonzero(keep)[0]
    sel = sel[::max(1, len(sel) // 200000)]
    pc = np.array([bin(int(m) & (2 ** 64 - 1)).count('1') for m in masks[sel]])
    osrc = pb.d_osrc.cpu().numpy().reshape(-1, 2)
    role = np.repeat(osrc[:, 1], np.diff(pb.out_off))[sel]
    for r, name in ((0, 'column direction (8 groups)'), (1, 'row direction (row blocks)')):
        v = pc[role == r]
        if len(v):
            print('classes per candidate, %s: mean %.2f, histogram %s' % (name, v.mean(), np.bincount(v)[:12].tolist()))
