"""SIFT stage timing on a synthetic 20 MP image (GPU box): python tools/sift_time.py [scale]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from imageanalysis_amd import kernels

def synth_image(h=3648, w=5472, seed=0):
    g = torch.Generator(device='cuda'); g.manual_seed(seed)
    img = torch.zeros((1, 1, h, w), device='cuda')
    for s in (2, 4, 8, 16, 32, 64):
        n = torch.randn((1, 1, h // s + 2, w // s + 2), generator=g, device='cuda')
        up = torch.nn.functional.interpolate(n, scale_factor=s, mode='bilinear', align_corners=False)[:, :, :h, :w]
        img += up * s ** 0.7
    img = (img - img.min()) / (img.max() - img.min()) * 255
    img = img[0, 0]
    return torch.stack([img, img * 0.9 + 10, img * 0.8 + 20], 2).clamp(0, 255).to(torch.uint8).contiguous()

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4
bgr = synth_image()
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    scaled = kernels.equalize_resize(bgr, scale)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    kp, octv, desc = kernels.sift_detect(scaled, cap=400000)
    t2 = time.perf_counter()
    print("iter %d: prep %.1f ms, sift %.1f ms (incl. host sort), %d keypoints, detect image %s" % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(kp), tuple(scaled.shape)))
