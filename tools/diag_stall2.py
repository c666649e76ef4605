#!/usr/bin/env python3
"""Which host-side activity before an LSMR solve makes its first chunk stall ~80 ms?"""
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


def solve():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, istop, itn, nr, nar = ba_solver.lsmr_device_fused(prob, d, dreg, maxiter=128)
    return (time.perf_counter() - t0) * 1e3


def nothing():
    pass


def sleep50():
    time.sleep(0.05)


def numpy_churn():
    for _ in range(20):
        a = np.random.rand(834305)
        b = a * 2 + 1
        del a, b


def uploads():
    for _ in range(5):
        prob.upload(np.random.rand(prob.n))


def downloads():
    for _ in range(5):
        prob.download(prob.x, prob.n)


def pageable_uploads():
    for _ in range(5):
        torch.from_numpy(np.random.rand(prob.n)).to('cuda')


def pageable_downloads():
    for _ in range(5):
        prob.x.cpu().numpy()


def small_kernels():
    for _ in range(20):
        prob.dot(prob.x, prob.x, prob.n, False)


def empty_alloc():
    ys = [torch.empty(prob.m, dtype=torch.float64, device='cuda') for _ in range(3)]
    del ys



def jac():
    prob.residual_jac()


def grad():
    prob.grad()


def colnorm():
    prob.colnorm()


def gram():
    prob.gram(np.ones(prob.n), [np.random.rand(prob.n)])


def fun():
    prob.set_x(x0)
    prob.residual()
    prob.cost_of_r(prob.r)


def qr_host():
    from scipy.linalg import qr
    S = np.random.rand(prob.n, 2)
    qr(S, mode='economic')


def host_blas():
    a = np.random.rand(prob.n)
    for _ in range(20):
        np.dot(a, a)
        np.linalg.norm(a)


def solve():
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    x, istop, itn, nr, nar = ba_solver.lsmr_device_fused(prob, d, dreg, maxiter=384)
    return (time.perf_counter() - t0) * 1e3


solve(); solve()
for name, fn in [('nothing', nothing), ('jac', jac), ('grad', grad), ('colnorm', colnorm), ('gram', gram),
                 ('fun', fun), ('qr_host', qr_host), ('host_blas', host_blas), ('nothing', nothing)]:
    ts = []
    for _ in range(3):
        fn()
        ts.append(solve())
    print('%-20s solve(384 its) ms: %s' % (name, ' '.join('%.1f' % t for t in ts)))
