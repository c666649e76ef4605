#!/usr/bin/env python3
"""The symmetric sweep alone on bench.py's dense-overlap store (12 x 16384 rows, 66 pairs): ms per
launch, best of 5 x 10.  IAMX_LIB selects the library (A/B of sweep variants on one box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from imageanalysis_amd import kernels  # noqa: E402

n_img, rows = 12, 16384
rng = np.random.default_rng(5)
g = rng.gamma(0.6, 1.0, size=(n_img, rows, 128))
g /= np.linalg.norm(g, axis=2, keepdims=True)
des = [np.clip(np.rint(np.minimum(x, 0.2) / np.linalg.norm(np.minimum(x, 0.2), axis=1, keepdims=True) * 512.0), 0, 255).astype(np.uint8)
       for x in g]
store = kernels.DescriptorStore.from_arrays(des)
und = [(a, b) for a in range(n_img) for b in range(a + 1, n_img)]
pb = kernels.PairBatch(store, np.array(und + [(b, a) for a, b in und], np.int32), sym=True)
ws = kernels.PairWorkspace(pb.rows, pb.n_pairs)
for _ in range(3):
    pb.run_sym_sweep(ws)
best = 1e9
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        pb.run_sym_sweep(ws)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
print('%s: sweep %.4f ms per launch of %d pairs x %d rows' % (os.path.basename(kernels._lib.LIB_PATH), best, len(und), rows))
