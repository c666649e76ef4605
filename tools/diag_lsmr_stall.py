#!/usr/bin/env python3
"""Where does the fused LSMR lose wall time?  Times every enqueue call and every state read."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from imageanalysis_amd import ba_solver, synth, _lib  # noqa: E402

p = synth.make_ba_problem()
C, P = len(p['cams0']), len(p['pts0'])
K = p['K']
calib = [K[0, 0], K[1, 1], K[0, 2], K[1, 2], *p['dist']]
prob = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], False, fixed_calib=calib)
x0 = np.hstack([p['cams0'].ravel(), p['pts0'].ravel()])
prob.set_x(x0)
prob.residual_jac()
cn = prob.colnorm()
cn[cn == 0] = 1
d = prob.upload_n(1.0 / cn)
dreg = torch.full((prob.n,), 1e-3, dtype=torch.float64, device='cuda')

L = _lib.lib()
orig = L.iamx_ba_lsmr_iterate
log = []


def timed_iterate(*a):
    t0 = time.perf_counter()
    r = orig(*a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    log.append((t1 - t0, t2 - t1))
    return r


class Wrap(object):
    def __getattr__(self, k):
        return timed_iterate if k == 'iamx_ba_lsmr_iterate' else getattr(L, k)


ba_solver.lib = lambda: Wrap()
for rep in range(3):
    log.clear()
    t0 = time.perf_counter()
    x, istop, itn, nr, nar = ba_solver.lsmr_device_fused(prob, d, dreg, chunk=int(os.environ.get('CHUNK', 64)))
    dt = time.perf_counter() - t0
    print('solve %d: itn=%d istop=%d  %.1f ms total;  enqueue/sync per chunk (ms):' % (rep, itn, istop, dt * 1e3))
    print('   ' + ' '.join('%.1f/%.1f' % (a * 1e3, b * 1e3) for a, b in log))

lb, ub = np.full(x0.size, -np.inf), np.full(x0.size, np.inf)
for j, dlt in ((0, 3.0), (1, 3.0), (2, 9.0)):
    lb[j:C * 7:7] = p['cams0'][:, j] - dlt
    ub[j:C * 7:7] = p['cams0'][:, j] + dlt
for rep in range(2):
    log.clear()
    t0 = time.perf_counter()
    res = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4, max_nfev=4)
    dt = time.perf_counter() - t0
    print('trf %d: %d its, %d lsmr its, %.1f ms; enqueue/sync per chunk (ms):' % (
        rep, res.iterations, res.lsmr_iterations, dt * 1e3))
    print('   ' + ' '.join('%.1f/%.1f' % (a * 1e3, b * 1e3) for a, b in log))

# replay the LAST solve of the TRF run stand-alone (same J, r, d, dreg): data-dependent stall?
saved = {}
orig_lsmr = ba_solver.lsmr


def rec(prob_, d_dev, dreg_dev, **kw):
    saved['d'], saved['dreg'] = d_dev.clone(), dreg_dev.clone()
    saved['r'], saved['Jc'], saved['Jp'] = prob_.r.clone(), prob_.Jc.clone(), prob_.Jp.clone()
    return orig_lsmr(prob_, d_dev, dreg_dev, **kw)


ba_solver.lsmr = rec
res = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4, max_nfev=4)
ba_solver.lsmr = orig_lsmr
prob.r.copy_(saved['r']); prob.Jc.copy_(saved['Jc']); prob.Jp.copy_(saved['Jp'])
for rep in range(3):
    log.clear()
    x, istop, itn, nr, nar = ba_solver.lsmr_device_fused(prob, saved['d'], saved['dreg'])
    print('replay %d: itn=%d; chunks: %s' % (rep, itn, ' '.join('%.1f/%.1f' % (a * 1e3, b * 1e3) for a, b in log)))
print('dreg min/max', float(saved['dreg'].min()), float(saved['dreg'].max()),
      ' d min/max', float(saved['d'].min()), float(saved['d'].max()))
