"""Device time of whole SIFT detections on the launch stream (GPU box): hipEvents around N calls of
iamx_sift_detect on one resident detect image (what bench.py's sift.roofline is timed on), and
the same with 8 detector threads in flight (what image.prefetch does).
    python tools/sift_stream_time.py [n]            IAMX_SIFT_GRAPH=1 for the captured-graph form"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imageanalysis_amd import kernels, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
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


def detect(b, stream):
    kernels.check(L.iamx_sift_detect(kernels._ptr(scaled), h, w, 3, 0.04, 10.0, 1.6, kernels._ptr(b[0]), need,
                                     kernels._ptr(b[1]), kernels._ptr(b[2]), cap, kernels._ptr(b[3]),
                                     stream.cuda_stream), 'iamx_sift_detect')


b0 = buffers()
st = torch.cuda.current_stream()
for _ in range(3):
    detect(b0, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(n):
    detect(b0, st)
e1.record()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
alg = 469.0 * h * w
print("graph=%s  one stream: %.3f ms per detect on the stream (host enqueue %.3f ms), %d keypoints, "
      "%.1f GB/s algorithmic = %.4f of 8 TB/s" % (os.environ.get('IAMX_SIFT_GRAPH') == '1', ms, t_host / n * 1e3,
                                                 int(b0[3].item()), alg / ms / 1e6, alg / ms / 1e6 / 8000.0))
if os.environ.get('IAMX_SIFT_SINGLE') == '1':
    sys.exit(0)
# 8 detectors in flight, one thread + stream + buffer set each
K = 8
bufs = [buffers() for _ in range(K)]
streams = [torch.cuda.Stream() for _ in range(K)]


def worker(k, reps):
    with torch.cuda.stream(streams[k]):
        for _ in range(reps):
            detect(bufs[k], streams[k])


for reps in (2, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(k, reps)) for k in range(K)]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("graph=%s  %d streams: %.3f ms per detect (wall, %d detects), %.1f GB/s algorithmic = %.4f of 8 TB/s"
      % (os.environ.get('IAMX_SIFT_GRAPH') == '1', K, dt / (K * n) * 1e3, K * n, alg / (dt / (K * n)) / 1e9,
         alg / (dt / (K * n)) / 1e9 / 8000.0))
