#!/usr/bin/env python3
"""BASELINE configs[3] with the 8 calibration columns of optimize_calib='global'
(`process.py --cam-calibration`): the bordered Schur inner solver against LSMR on the whole
system -- outer / inner iterations, seconds, end cost.     python tools/ba_calib_rate.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from imageanalysis_amd import ba_solver, synth  # noqa: E402

p = synth.make_ba_problem()
C, P = len(p['cams0']), len(p['pts0'])
K = p['K']
# start 1.5 % off in the focal length and a little in the principal point / distortion
cal0 = np.array([K[0, 0] * 1.015, K[0, 2] + 6.0, K[1, 2] - 4.0, *(np.asarray(p['dist']) * 0.7)])
x0 = np.hstack([p['cams0'].ravel(), p['pts0'].ravel(), cal0])
lb = np.full(x0.size, -np.inf)
ub = np.full(x0.size, np.inf)
for j, dlt in ((0, 3.0), (1, 3.0), (2, 9.0)):
    lb[j:C * 7:7] = p['cams0'][:, j] - dlt
    ub[j:C * 7:7] = p['cams0'][:, j] + dlt
k0 = C * 7 + P * 3
lb[k0:k0 + 3] = [K[0, 0] * 0.8, K[0, 2] * 0.8, K[1, 2] * 0.8]
ub[k0:k0 + 3] = [K[0, 0] * 1.2, K[0, 2] * 1.2, K[1, 2] * 1.2]
lb[k0 + 5:k0 + 7], ub[k0 + 5:k0 + 7] = -0.2, 0.2
from threadpoolctl import threadpool_limits  # noqa: E402
with threadpool_limits(limits=1, user_api='blas'):
    for inner in ('schur', 'lsmr') + (('schur',) if '--twice' in sys.argv else ()):
        prob = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], True)
        prob.inner = inner
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        its = list(prob.inner_iterations)
        print('%-5s: status %d, %d outer iterations in %.3f s (%.1f it/s), inner iterations %d (max %d per solve), '
              'rms residual %.4f px, calib %s' % (inner, res.status, res.iterations, dt, res.iterations / dt,
                                                   sum(its), max(its), np.sqrt(res.cost / len(p['uv'])),
                                                   np.array2string(res.x[k0:], precision=4)))
