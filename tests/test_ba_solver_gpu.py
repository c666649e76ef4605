"""GPU: K4 operator kernels, device LSMR and the device TRF driver."""
import glob
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO
from test_host_logic import _scene

pytestmark = pytest.mark.gpu
BA_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'ba_*.npz')))


def _problem(path):
    import torch
    from imageanalysis_amd import ba_solver, optimizer
    g = np.load(path)
    proj, inp = _scene(path)
    opt = optimizer.Optimizer('/nonexistent')
    opt.setup(proj, inp['groups'], 0, inp['matches'], cam_calib=bool(g['cam_calib']))
    wc = bool(g['cam_calib'])
    K, dc = opt.K, opt.distCoeffs
    prob = ba_solver.DeviceBA(opt.n_cameras, opt.n_points, opt.camera_indices, opt.point_indices,
                              g['points_2d'], wc,
                              fixed_calib=[K[0, 0], K[1, 1], K[0, 2], K[1, 2], *dc])
    return g, opt, prob


@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
def test_k4_operators_vs_csr(path):
    import torch
    g, opt, prob = _problem(path)
    x0 = g['x0']
    args = (opt.n_cameras, opt.n_points, opt.by_camera_point_indices, opt.by_camera_points_2d)
    J = opt.jac(x0, *args)
    prob.set_x(x0)
    prob.residual_jac()
    rng = np.random.default_rng(0)
    v = rng.normal(size=prob.n)
    u = rng.normal(size=prob.m)
    y = torch.empty(prob.m, dtype=torch.float64, device='cuda')
    prob.jv(prob.upload_n(v), y)               # n-vectors cross in the reference's order
    ref = J @ v
    assert np.abs(prob.download_m(y) - ref).max() <= 1e-12 * np.abs(ref).max()   # m-vectors too
    out = torch.empty(prob.n, dtype=torch.float64, device='cuda')
    prob.jtv(prob.upload_m(u), out)
    ref = J.T @ u
    assert np.abs(prob.download_n(out) - ref).max() <= 1e-12 * np.abs(ref).max()
    # the private point / observation orders really are permutations of the reference's
    assert np.array_equal(prob.download_n(prob.upload_n(v)), v)
    assert np.array_equal(prob.download_m(prob.upload_m(u)), u)
    assert np.abs(prob.download_m(prob.r) - g['f0']).max() <= 1e-9 * np.abs(g['f0']).max()
    cn = prob.colnorm()
    ref = np.sqrt(np.asarray(J.power(2).sum(axis=0)).ravel())
    assert np.abs(cn - ref).max() <= 1e-12 * ref.max()
    gr = prob.grad()
    ref = J.T @ g['f0']
    assert np.abs(gr - ref).max() <= 1e-9 * np.abs(ref).max()
    assert abs(prob.cost_of_r(prob.r) - 0.5 * g['f0'] @ g['f0']) <= 1e-12 * (g['f0'] @ g['f0'])


@pytest.mark.parametrize('path', BA_CASES[:2], ids=os.path.basename)
def test_device_lsmr_equals_scipy_lsmr(path):
    import torch
    from scipy.sparse import diags, vstack
    from scipy.sparse.linalg import lsmr
    from imageanalysis_amd import ba_solver
    g, opt, prob = _problem(path)
    x0 = g['x0']
    args = (opt.n_cameras, opt.n_points, opt.by_camera_point_indices, opt.by_camera_points_2d)
    J = opt.jac(x0, *args)
    prob.set_x(x0)
    prob.residual_jac()
    rng = np.random.default_rng(1)
    d = 1.0 / np.maximum(np.sqrt(np.asarray(J.power(2).sum(axis=0)).ravel()), 1e-9)
    dreg = rng.uniform(0.01, 0.1, prob.n)
    A = vstack([J @ diags(d), diags(dreg)]).tocsr()
    b = np.concatenate([g['f0'], np.zeros(prob.n)])
    # same recurrences => after the same number of iterations the iterates agree to rounding
    # (for a few iterations: Golub-Kahan bidiagonalisation amplifies rounding differences
    #  exponentially on this ill-conditioned operator, 3e-3 after 40 steps)
    for k in (1, 2, 5, 10):
        ref = lsmr(A, b, atol=0, btol=0, conlim=0, maxiter=k)
        x, istop, itn, normr, normar = ba_solver.lsmr_device(
            prob, prob.upload_n(d), prob.upload_n(dreg),
            atol=0, btol=0, conlim=0, maxiter=k)
        assert istop == ref[1] == 7 and itn == ref[2] == k
        assert np.abs(x - ref[0]).max() <= 1e-10 * np.abs(ref[0]).max()
        assert abs(normr - ref[3]) <= 1e-10 * ref[3]
    # with the default tolerances both stop on the same test with equally good solutions
    ref = lsmr(A, b, atol=1e-6, btol=1e-6, conlim=1e8)
    x, istop, itn, normr, normar = ba_solver.lsmr_device(
        prob, prob.upload_n(d), prob.upload_n(dreg))
    assert istop == ref[1] and abs(itn - ref[2]) <= max(3, ref[2] // 10)
    res_dev, res_ref = np.linalg.norm(A @ x - b), np.linalg.norm(A @ ref[0] - b)
    assert abs(res_dev - res_ref) <= 1e-4 * res_ref


@pytest.mark.parametrize('path', [p for p in BA_CASES if not bool(np.load(p)['cam_calib'])],
                         ids=os.path.basename)
def test_fused_lsmr_equals_stepwise_and_scipy(path):
    """iamx_ba_lsmr_iterate (scalars on the device, no host sync) against scipy's lsmr and the
    stepwise device form: same iterates for small k, same stop test / iteration count."""
    import torch
    from scipy.sparse import diags, vstack
    from scipy.sparse.linalg import lsmr
    from imageanalysis_amd import ba_solver
    g, opt, prob = _problem(path)
    x0 = g['x0']
    args = (opt.n_cameras, opt.n_points, opt.by_camera_point_indices, opt.by_camera_points_2d)
    J = opt.jac(x0, *args)
    prob.set_x(x0)
    prob.residual_jac()
    rng = np.random.default_rng(2)
    d = 1.0 / np.maximum(np.sqrt(np.asarray(J.power(2).sum(axis=0)).ravel()), 1e-9)
    dreg = rng.uniform(0.01, 0.1, prob.n)
    A = vstack([J @ diags(d), diags(dreg)]).tocsr()
    b = np.concatenate([g['f0'], np.zeros(prob.n)])
    dd, dr = prob.upload_n(d), prob.upload_n(dreg)
    for k in (1, 2, 5, 10):
        ref = lsmr(A, b, atol=0, btol=0, conlim=0, maxiter=k)
        x, istop, itn, normr, normar = ba_solver.lsmr_device_fused(
            prob, dd, dr, atol=0, btol=0, conlim=0, maxiter=k, chunk=4)
        assert istop == ref[1] == 7 and itn == ref[2] == k
        assert np.abs(x - ref[0]).max() <= 1e-9 * np.abs(ref[0]).max()
        assert abs(normr - ref[3]) <= 1e-10 * ref[3]
        assert abs(normar - ref[4]) <= 1e-8 * max(ref[4], 1e-300)
    ref = lsmr(A, b, atol=1e-6, btol=1e-6, conlim=1e8)
    x, istop, itn, normr, normar = ba_solver.lsmr_device_fused(prob, dd, dr)
    xs, istop_s, itn_s, _, _ = ba_solver.lsmr_device(prob, dd, dr)
    assert istop == ref[1] == istop_s
    assert abs(itn - ref[2]) <= max(3, ref[2] // 10) and abs(itn - itn_s) <= max(3, itn_s // 10)
    res_dev, res_ref = np.linalg.norm(A @ x - b), np.linalg.norm(A @ ref[0] - b)
    assert abs(res_dev - res_ref) <= 1e-4 * res_ref
    # deterministic: fixed reduction trees, no atomics
    x2 = ba_solver.lsmr_device_fused(prob, dd, dr)[0]
    assert np.array_equal(x, x2)


@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
def test_device_trf_reaches_reference_minimum(path):
    from imageanalysis_amd import optimizer
    g = np.load(path)
    proj, inp = _scene(path)
    opt = optimizer.Optimizer('/nonexistent')
    opt.solver = 'device'
    opt.setup(proj, inp['groups'], 0, inp['matches'], cam_calib=bool(g['cam_calib']))
    ret = opt.run()
    res = opt.result
    cost = 0.5 * float(res.fun @ res.fun)
    ref = float(g['cost_final'])
    assert abs(cost - ref) / ref < 2e-2, (cost, ref)
    assert abs(np.mean(np.abs(res.fun)) - np.mean(np.abs(g['f_final']))) < 0.02
    assert res.njev <= 3 * 8 and res.status in (1, 2, 3, 4)
    lo, up = opt._bounds()
    assert np.all(res.x >= np.asarray(lo) - 1e-12) and np.all(res.x <= np.asarray(up) + 1e-12)
    assert ret[0].shape == (opt.n_cameras, 7)


@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
def test_device_resident_trf_logic_equals_host_logic(path):
    """the outer iteration with its n-vectors on the device (default) against the same
    iteration in numpy with SciPy's helper functions (host_logic): same trajectory"""
    from imageanalysis_amd import ba_solver
    out = []
    for host_logic in (False, True):
        g, opt, prob = _problem(path)
        prob.host_logic = host_logic
        # (the logic is what is compared: the inner solves run to a tight tolerance, so that the
        #  truncation of the default forcing term -- which amplifies rounding differences between
        #  the two implementations of the same step selection -- stays out of it)
        prob.schur_eta, prob.schur_qtol = 1e-6, 0.0
        lo, up = opt._bounds()
        res = ba_solver.trf_device(prob, opt._x0(), np.asarray(lo, float), np.asarray(up, float),
                                   ftol=1e-4)
        out.append(res)
    a, b = out
    # Both follow trf.py step by step; their reductions round differently, and with the inexact
    # (atol = btol = 1e-6) LSMR solves that moves individual iterates in the 4th digit.  What
    # must agree is what ftol = 1e-4 promises: the minimum reached, how fast, and feasibility.
    assert a.status == b.status and abs(a.njev - b.njev) <= 2 and abs(a.nfev - b.nfev) <= 2
    # (the calibration columns are weakly determined together with the camera heights: the two
    #  implementations stop a few ftol apart there)
    assert abs(a.cost - b.cost) <= (1e-3 if prob.with_calib else 2e-4) * b.cost
    lo, up = np.asarray(lo, float), np.asarray(up, float)
    assert np.all(a.x >= lo) and np.all(a.x <= up)
    assert np.array_equal(a.active_mask != 0, b.active_mask != 0)


def _two_rank_solve(rank, world, port, path, outdir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)                         # both ranks share the one GPU of the box
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from imageanalysis_amd import optimizer
    g = np.load(path)
    proj, inp = _scene(path)
    opt = optimizer.Optimizer('/nonexistent')
    opt.solver = 'device-lsmr'
    opt.setup(proj, inp['groups'], 0, inp['matches'], cam_calib=bool(g['cam_calib']))
    # record what the solve all-reduces (element counts) and that it runs the fused phase path
    from imageanalysis_amd import ba_solver, dist as D
    sizes = []
    plain = D.allreduce_sum_

    inside = [False]                                  # inside the fused LSMR iterations
    real_fused = ba_solver.lsmr_device_fused

    def fused(*a, **k):
        inside[0] = True
        try:
            return real_fused(*a, **k)
        finally:
            inside[0] = False
    ba_solver.lsmr_device_fused = fused

    def counting(t, group=None):
        # (reductions outside the LSMR iterations -- the TRF scalars -- are recorded negative)
        sizes.append(int(t.numel()) if inside[0] else -int(t.numel()))
        return plain(t, group)
    D.allreduce_sum_ = ba_solver._dist.allreduce_sum_ = counting
    phases = []
    real_phase = ba_solver.lib().iamx_ba_lsmr_phase

    class _Lib(object):                               # ctypes functions cannot be patched in place
        def __getattr__(self, name):
            if name == 'iamx_ba_lsmr_phase':
                def call(*a):
                    phases.append(int(a[-3]))        # (..., phase, parity, stream)
                    return real_phase(*a)
                return call
            return getattr(ba_solver._lib.lib(), name)
    ba_solver.lib = lambda: _Lib()
    opt.run()
    np.save(os.path.join(outdir, 'x_r%d.npy' % rank), opt.result.x)
    np.save(os.path.join(outdir, 'f_r%d.npy' % rank), opt.result.fun)
    np.save(os.path.join(outdir, 'red_r%d.npy' % rank), np.array(sizes, np.int64))
    np.save(os.path.join(outdir, 'ph_r%d.npy' % rank), np.array(phases, np.int64))
    dist.barrier()
    dist.destroy_process_group()


def test_device_trf_two_ranks_point_sharded(tmp_path):
    """observations sharded by point over 2 ranks (gloo, same GPU): same solution as 1 rank."""
    import torch.multiprocessing as mp
    from imageanalysis_amd import optimizer
    path = [p for p in BA_CASES if p.endswith('ba_mid.npz')][0]
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_solve, args=(2, port, path, str(tmp_path)), nprocs=2, join=True)
    g = np.load(path)
    proj, inp = _scene(path)
    opt = optimizer.Optimizer('/nonexistent')
    opt.solver = 'device-lsmr'
    opt.setup(proj, inp['groups'], 0, inp['matches'])
    opt.run()
    x0 = np.load(tmp_path / 'x_r0.npy')
    x1 = np.load(tmp_path / 'x_r1.npy')
    assert np.array_equal(x0, x1)
    f0 = np.load(tmp_path / 'f_r0.npy')
    assert f0.shape == opt.result.fun.shape
    c1, c2 = 0.5 * f0 @ f0, 0.5 * opt.result.fun @ opt.result.fun
    assert abs(c1 - c2) / c2 < 1e-5      # (the partial sums of 1 and 2 ranks round differently)
    assert np.abs(x0 - opt.result.x).max() < 1e-5 * np.abs(opt.result.x).max()
    # the two ranks ran the fused LSMR through iamx_ba_lsmr_phase: phases 0, 1, 2 in turn, and per
    # iteration exactly two all-reduces -- 2 scalars and the camera part of J^T u + 1 scalar; the
    # point part of the n-vectors is never reduced inside the iteration
    C, n = opt.n_cameras, opt.result.x.size
    for r in range(2):
        ph = np.load(tmp_path / ('ph_r%d.npy' % r))
        assert len(ph) >= 3 * 64 and np.array_equal(ph, np.tile([0, 1, 2], len(ph) // 3))
        red = np.load(tmp_path / ('red_r%d.npy' % r))
        iters = len(ph) // 3
        assert (red == 2).sum() == iters and (red == 7 * C + 1).sum() == iters
        # full n-vectors are reduced only O(1) times per outer iteration (gradient, init, x)
        assert (np.abs(red) >= n - 7 * C).sum() <= 8 * (opt.result.njev + 2)
