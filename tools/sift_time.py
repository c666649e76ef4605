"""SIFT stage timing on a synthetic 20 MP image (GPU box): python tools/sift_time.py [scale]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from imageanalysis_amd import kernels, synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4
bgr = synth.make_survey_image()
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    scaled = kernels.equalize_resize(bgr, scale)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    kp, octv, desc = kernels.sift_detect(scaled, cap=1500000)
    t2 = time.perf_counter()
    print("iter %d: prep %.1f ms, sift %.1f ms (incl. host sort), %d keypoints, detect image %s" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(kp), tuple(scaled.shape)))
