#!/usr/bin/env python3
"""Host -> HBM ingest of descriptors as the drop-in path sees them (Image.des_list: float32
[N,128], integer valued): DeviceMatcher.slot_of + store() for a batch of images."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from imageanalysis_amd import matcher  # noqa: E402


class Img(object):
    def __init__(self, name, des, xy):
        self.name, self.des_list = name, des
        self.kp_list = None
        self._iamx_xy = (None, xy)


rng = np.random.default_rng(0)
n_img, n_kp = 256, 4096
imgs = [Img('I%04d' % i, rng.integers(0, 256, (n_kp, 128)).astype(np.float32),
            rng.uniform(0, 3000, (n_kp, 2)).astype(np.float32)) for i in range(n_img)]
dm = matcher.DeviceMatcher()
torch.cuda.synchronize()
for lo in (0, 64, 128):                     # three growth steps, like find_matches batches
    t0 = time.perf_counter()
    for im in imgs[lo:lo + 64 if lo < 128 else n_img]:
        dm.slot_of(im)
    t1 = time.perf_counter()
    dm.store()
    dm.keypoints()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    k = (64 if lo < 128 else n_img - 128)
    print('%3d images: register %.1f ms (keys), store+keypoints %.1f ms -> %.2f ms / image, %.2f GB/s of f32 descriptors'
          % (k, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3 / k, k * n_kp * 512 / (t2 - t0) / 1e9))
