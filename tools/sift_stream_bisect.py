"""bench.py times 2.8 ms per SIFT detection on the launch stream, tools/sift_stream_time.py 1.7 ms
with the same kernels and buffers; 40 extra streams change nothing (tools/sift_stream_clutter.py).
This script measures the single-stream detection after each thing bench.py has done by the time it
reaches its SIFT section, to find the one that costs 17 us per dependent kernel.
    python tools/sift_stream_bisect.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from imageanalysis_amd import kernels, synth

dev = torch.device('cuda', 0)
img = synth.make_survey_image(seed=100, device=dev)
scaled = kernels.equalize_resize(img, 0.4)
L = kernels.lib()
h, w = scaled.shape[0], scaled.shape[1]
need = int(L.iamx_sift_workspace_bytes(h, w))
cap = 400000


def buffers():
    return (torch.empty(need, dtype=torch.uint8, device=dev), torch.empty((cap, 8), dtype=torch.float32, device=dev),
            torch.empty((cap, 128), dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))


b = buffers()


def detect(bb):
    kernels.check(L.iamx_sift_detect(kernels._ptr(scaled), h, w, 3, 0.04, 10.0, 1.6, kernels._ptr(bb[0]), need,
                                     kernels._ptr(bb[1]), kernels._ptr(bb[2]), cap, kernels._ptr(bb[3]),
                                     torch.cuda.current_stream().cuda_stream), 'iamx_sift_detect')


def measure(tag, bb=None, n=20):
    bb = bb or b
    for _ in range(3):
        detect(bb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        detect(bb)
    e1.record()
    torch.cuda.synchronize()
    print("%-64s %.3f ms per detect" % (tag, e0.elapsed_time(e1) / n), flush=True)


measure("(a) fresh process")
pin = torch.empty(1 << 28, dtype=torch.uint8).pin_memory()
measure("(b) + 256 MB of page-locked host memory")
big = [torch.empty(8 << 30, dtype=torch.uint8, device=dev) for _ in range(5)]
measure("(c) + 40 GB of device memory allocated")
del big
measure("(d) ... released to the caching allocator (not to the driver)")
torch.cuda.empty_cache()
measure("(e) ... empty_cache()")
# the matching section's kernels (code objects, workspaces)
rng = np.random.default_rng(0)
arrs = [rng.integers(0, 256, (4096, 128), dtype=np.uint8) for _ in range(8)]
store = kernels.DescriptorStore.from_arrays(arrs)
und = [(a, c) for a in range(8) for c in range(a + 1, 8)]
pb = kernels.PairBatch(store, np.array(und + [(c, a) for a, c in und], np.int32))
ws = kernels.PairWorkspace(pb.rows, pb.n_pairs)
pb.run(ws, 202.5)
torch.cuda.synchronize()
measure("(f) + a symmetric matching batch has run")
measure("(g) fresh buffer set (like bench.py's)", bb=buffers())
from threadpoolctl import threadpool_limits
with threadpool_limits(limits=1, user_api='blas'):
    measure("(h) inside threadpool_limits(1)")
import bench
class A: pass
args = A(); args.ba_iters = 0
bench.ba_bench(0, 1, dev, None, args)
torch.cuda.synchronize()
measure("(i) + bench.ba_bench has run")
torch.cuda.empty_cache()
measure("(j) ... empty_cache()")
s2 = kernels.sift_detect(scaled, cap=400000)
measure("(k) + one kernels.sift_detect (sort, pinned download)")
