"""The symmetric form on bench.py's dense-overlap workload, nothing else (for counter passes)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch

import bench
from imageanalysis_amd import kernels

dev = torch.device('cuda:0')
n_img, rows = 12, 16384
g = torch.Generator(device=dev)
g.manual_seed(77)
alpha = torch.full((rows, bench.DIM), 0.6, device=dev)


def fresh(n):
    x = torch._standard_gamma(alpha[:n], generator=g)
    x = x / x.norm(dim=1, keepdim=True)
    x = x.clamp(max=0.2)
    x = x / x.norm(dim=1, keepdim=True)
    return (x * 512.0).round().clamp(0, 255)


base = fresh(rows)
imgs = [base.to(torch.uint8)]
for _ in range(n_img - 1):
    src = torch.randint(0, rows, (rows,), generator=g, device=dev)
    im = (base[src] + torch.randint(-3, 4, (rows, bench.DIM), generator=g, device=dev)).clamp(0, 255)
    new = torch.rand(rows, generator=g, device=dev) < 0.45
    im[new] = fresh(rows)[new]
    imgs.append(im.to(torch.uint8))
store = kernels.DescriptorStore.from_arrays([im.cpu().numpy() for im in imgs])
und = [(a, b) for a in range(n_img) for b in range(a + 1, n_img)]
ordered = np.array(und + [(b, a) for a, b in und], np.int32)
pb = kernels.PairBatch(store, ordered, sym=True)
ws = kernels.PairWorkspace(pb.rows, pb.n_pairs)
for _ in range(4):
    pb.run_knn2_fast(ws)
    pb.run_filter_fast(ws, bench.MAX_DISTANCE * bench.MATCH_RATIO)
torch.cuda.synchronize()
print("candidates", int(ws.seg_count[:pb.n_pairs].sum().item()))
