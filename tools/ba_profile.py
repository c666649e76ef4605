#!/usr/bin/env python3
"""Phase breakdown of the device TRF solve on BASELINE.json configs[4] (synthetic).
usage: python tools/ba_profile.py [iters]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from imageanalysis_amd import ba_solver, synth  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = synth.make_ba_problem()
C, P = len(p['cams0']), len(p['pts0'])
K = p['K']
calib = [K[0, 0], K[1, 1], K[0, 2], K[1, 2], *p['dist']]
prob = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], False, fixed_calib=calib)
x0 = np.hstack([p['cams0'].ravel(), p['pts0'].ravel()])
lb, ub = np.full(x0.size, -np.inf), np.full(x0.size, np.inf)
for j, dlt in ((0, 3.0), (1, 3.0), (2, 9.0)):
    lb[j:C * 7:7] = p['cams0'][:, j] - dlt
    ub[j:C * 7:7] = p['cams0'][:, j] + dlt
for profile in (None, {}):
    prob.profile = profile
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4, max_nfev=iters + 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('profile=%s: %d iterations, %d lsmr its, %.3f s -> %.2f it/s' % (
        profile is not None, res.iterations, res.lsmr_iterations, dt, res.iterations / dt))
    if profile:
        for k, v in sorted(profile.items(), key=lambda kv: -kv[1]):
            print('   %-14s %8.1f ms  (%.1f %%)' % (k, v * 1e3, 100 * v / dt))

if os.environ.get('IAMX_BA_CPROFILE'):
    # where the host spends the solve (the GPU is busy for well under half of it)
    import cProfile
    import pstats
    prob.profile = None
    pr = cProfile.Profile()
    pr.enable()
    res = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4, max_nfev=iters + 1)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(28)
