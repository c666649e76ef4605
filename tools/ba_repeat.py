import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from imageanalysis_amd import ba_solver, synth
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
for nf in (2, None, None, None, 2, None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4, max_nfev=nf)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('max_nfev=%s: %d iterations, %.4f s -> %.1f it/s' % (nf, res.iterations, dt, res.iterations / dt))
