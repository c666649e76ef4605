#!/usr/bin/env python3
"""Achievable HBM bandwidth of a plain device copy as a function of the bytes moved (read +
write): the yardstick for the short BA kernels (125 MB residual, 448 MB residual + Jacobian)."""
import torch

for mb in (64, 125, 250, 448, 883, 2000, 8000):
    n = mb * 1000 * 1000 // 16            # float64 elements per side (read n*8 + write n*8)
    a = torch.empty(n, dtype=torch.float64, device='cuda').normal_()
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print('%5d MB moved: %7.1f us  %6.2f TB/s' % (mb, t * 1e6, 16.0 * n / t / 1e12))
