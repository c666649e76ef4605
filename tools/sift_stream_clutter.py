"""Why does bench.py see 2.8 ms per detection on the launch stream where tools/sift_stream_time.py
sees 1.7 ms with the same kernels?  Hypothesis: a process that has created many HIP streams (the
matching section's side streams, detector slots, prefetch workers) maps them onto a few hardware
queues, and the fork / join between the launch stream and the detector's side stream then
serialises behind other queues' barrier packets.  Measures N detections on the current stream
(a) in a fresh process, (b) with 40 idle streams alive, (c) after those streams have each run a
kernel, (d) on a non-default stream.    python tools/sift_stream_clutter.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imageanalysis_amd import kernels, synth

dev = torch.device('cuda', 0)
img = synth.make_survey_image(seed=100, device=dev)
scaled = kernels.equalize_resize(img, 0.4)
L = kernels.lib()
h, w = scaled.shape[0], scaled.shape[1]
need = int(L.iamx_sift_workspace_bytes(h, w))
cap = 400000
b = (torch.empty(need, dtype=torch.uint8, device=dev), torch.empty((cap, 8), dtype=torch.float32, device=dev),
     torch.empty((cap, 128), dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev))


def detect(stream):
    kernels.check(L.iamx_sift_detect(kernels._ptr(scaled), h, w, 3, 0.04, 10.0, 1.6, kernels._ptr(b[0]), need,
                                     kernels._ptr(b[1]), kernels._ptr(b[2]), cap, kernels._ptr(b[3]),
                                     stream.cuda_stream), 'iamx_sift_detect')


def measure(tag, stream=None, n=20):
    st = stream or torch.cuda.current_stream()
    with torch.cuda.stream(st):
        for _ in range(3):
            detect(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st)
        for _ in range(n):
            detect(st)
        e1.record(st)
        t_host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
    print("%-52s %.3f ms per detect on the stream, host enqueue %.3f ms" % (tag, e0.elapsed_time(e1) / n, t_host * 1e3))


measure("(a) fresh process, default stream")
streams = [torch.cuda.Stream() for _ in range(40)]
measure("(b) + 40 idle streams alive")
x = torch.zeros(1 << 20, device=dev)
for s in streams:
    with torch.cuda.stream(s):
        x.add_(1)
torch.cuda.synchronize()
measure("(c) + each of them has run a kernel")
measure("(d) same, on a non-default stream", stream=torch.cuda.Stream())
hp = torch.cuda.Stream(priority=-1)
measure("(e) same, on a high-priority stream", stream=hp)
del streams
import gc; gc.collect(); torch.cuda.synchronize()
measure("(f) the 40 streams released, default stream")
