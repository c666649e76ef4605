#!/usr/bin/env python3
"""Device SIFT (csrc/sift.hip) against the oracle (oracle/sift_oracle.py), row by row in OpenCV's
output order: counts, duplicates removed, bit-equal keypoint fields, descriptor byte differences.
    python tools/sift_parity.py [whole]"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
from imageanalysis_amd import kernels          # noqa: E402
from oracle import sift_oracle as so           # noqa: E402
from test_sift_gpu import texture              # noqa: E402

cases = [((200, 260), 0), ((167, 301), 3), ((240, 180), 5), ((400, 520), 0), ((600, 800), 3)]
if len(sys.argv) > 1:
    cases.append(((1459, 2189), 21))
for shape, seed in cases:
    gray = so.bgr_to_gray(texture(shape[0], shape[1], seed))
    t0 = time.time()
    kps, des, removed = so.detect_and_compute(gray, return_removed=True)
    t1 = time.time()
    kp, octv, d = kernels.sift_detect(gray)
    print('%s: oracle %d (removed %d, %.1f s)  device %d (removed %d)' % (
        shape, len(kps), removed, t1 - t0, len(kp), kernels.sift_detect.last_removed))
    if len(kp) != len(kps):
        # align on (x, y, size, angle) bit patterns
        key = lambda a: [tuple(r) for r in np.asarray(a[:, :4], np.float32).view(np.int32)]
        ka, kb = key(kps), key(kp)
        sa, sb = set(ka), set(kb)
        print('   only in oracle: %d, only on device: %d' % (len(sa - sb), len(sb - sa)))
        ia = [i for i, k in enumerate(ka) if k in sb]
        ib = {k: i for i, k in enumerate(kb)}
        ib = [ib[ka[i]] for i in ia]
        kps, des, kp, octv, d = kps[ia], des[ia], kp[ib], octv[ib], d[ib]
    f32 = kps[:, :5].astype(np.float32)
    same = (f32.view(np.int32) == kp.view(np.int32))
    print('   rows with all five fields bit-equal: %d / %d; per field %s; octave equal %d' % (
        same.all(1).sum(), len(kp), same.sum(0).tolist(), (kps[:, 5].astype(np.int64) == octv).sum()))
    bad = ~same.all(1)
    if bad.any():
        print('   max abs diff of the differing rows', np.abs(f32[bad].astype(np.float64) - kp[bad]).max(0))
    diff = np.abs(des.astype(int) - d.astype(int))
    print('   descriptor bytes differing: %d of %d (%.5f %%), max %d, rows affected %d' % (
        (diff != 0).sum(), diff.size, 100.0 * (diff != 0).mean(), diff.max(), (diff != 0).any(1).sum()))
