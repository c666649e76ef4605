"""GPU: BA residual / Jacobian kernels against the oracle and the reference's golden vectors.
Floating point: tolerance 1e-5 relative (BASELINE.json north_star); we assert far tighter."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

BA_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'ba_*.npz')))
REL_TOL = 1e-5          # north_star tolerance
TIGHT = 1e-10           # what float64 on the device actually delivers


def _dev(a, dt=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dt is not None:
        t = t.to(dt)
    return t.cuda()


def _problem(g):
    C, P = int(g['n_cameras']), int(g['n_points'])
    x = g['x0']
    K = g['K']
    if bool(g['cam_calib']):
        cal = x[C * 7 + P * 3:]
        calib = np.array([cal[0], cal[0], cal[1], cal[2], *cal[3:8]])
    else:
        calib = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], *g['dist']])
    return C, P, x, calib


@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
def test_residual_golden(path):
    from imageanalysis_amd import kernels
    g = np.load(path)
    C, P, x, calib = _problem(g)
    r = kernels.ba_residual(_dev(x[:C * 7]), _dev(x[C * 7:C * 7 + P * 3]),
                            _dev(g['camera_indices']), _dev(g['point_indices']),
                            _dev(g['points_2d']), _dev(calib)).cpu().numpy()
    scale = np.abs(g['f0']).max()
    err = np.abs(r - g['f0']).max() / scale
    assert err < TIGHT < REL_TOL
    # and at the reference's converged solution
    xf = g['x_final']
    if bool(g['cam_calib']):
        cal = xf[C * 7 + P * 3:]
        calib = np.array([cal[0], cal[0], cal[1], cal[2], *cal[3:8]])
    r = kernels.ba_residual(_dev(xf[:C * 7]), _dev(xf[C * 7:C * 7 + P * 3]),
                            _dev(g['camera_indices']), _dev(g['point_indices']),
                            _dev(g['points_2d']), _dev(calib)).cpu().numpy()
    assert np.abs(r - g['f_final']).max() / scale < TIGHT


@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
def test_jacobian_vs_reference_finite_differences(path):
    """Analytic 2x(7+3[+8]) blocks == scipy's sparse finite differences of the REFERENCE's own
    fun (3-point scheme), entry by entry."""
    from imageanalysis_amd import kernels
    g = np.load(path)
    C, P, x, calib = _problem(g)
    wc = bool(g['cam_calib'])
    r, Jc, Jp, Jk = kernels.ba_residual_jac(_dev(x[:C * 7]), _dev(x[C * 7:C * 7 + P * 3]),
                                            _dev(g['camera_indices']), _dev(g['point_indices']),
                                            _dev(g['points_2d']), _dev(calib), with_calib=wc)
    assert np.abs(r.cpu().numpy() - g['f0']).max() / np.abs(g['f0']).max() < TIGHT
    Jc, Jp = Jc.cpu().numpy(), Jp.cpu().numpy()
    import scipy.sparse as sp
    n = x.size
    O = g['camera_indices'].size
    J3 = sp.csr_matrix((g['J3_data'], g['J3_indices'], g['J3_indptr']), shape=(2 * O, n)).toarray()
    ci, pi = g['camera_indices'], g['point_indices']
    ref_c = np.empty((O, 2, 7))
    ref_p = np.empty((O, 2, 3))
    for o in range(O):
        ref_c[o] = J3[2 * o:2 * o + 2, ci[o] * 7:ci[o] * 7 + 7]
        ref_p[o] = J3[2 * o:2 * o + 2, C * 7 + pi[o] * 3:C * 7 + pi[o] * 3 + 3]
    # finite differences are good to ~1e-7 relative to the largest entry of a row
    sc = np.abs(ref_c).max()
    assert np.abs(Jc - ref_c).max() / sc < 1e-6
    sp_ = np.abs(ref_p).max()
    assert np.abs(Jp - ref_p).max() / sp_ < 1e-6
    if wc:
        Jk = Jk.cpu().numpy()
        ref_k = J3[:, C * 7 + P * 3:].reshape(O, 2, 8)
        for k in range(8):
            s = max(np.abs(ref_k[:, :, k]).max(), 1e-12)
            assert np.abs(Jk[:, :, k] - ref_k[:, :, k]).max() / s < 1e-5, k


def test_residual_large_vs_c_oracle():
    """200k observations, random gather pattern, vs the plain-C oracle."""
    from imageanalysis_amd import kernels
    from oracle import cpu_ref
    rng = np.random.default_rng(7)
    C, P, O = 300, 30000, 200000
    cams = np.zeros((C, 7))
    cams[:, :3] = rng.normal(0, 50, (C, 3)) * [1, 1, 0.05] + [0, 0, -100]
    # nadir cameras (ypr = heading,-90,0 <=> q ~ (c,0,-c,0)), unnormalised quaternions
    cams[:, 3:] = np.array([0.7071, 0, -0.7071, 0]) * rng.uniform(0.5, 2.0, (C, 1)) \
        + rng.normal(0, 0.02, (C, 4))
    pts = rng.normal(0, 40, (P, 3)) * [1, 1, 0.05]
    ci = np.sort(rng.integers(0, C, O)).astype(np.int32)
    pi = rng.integers(0, P, O).astype(np.int32)
    uv = rng.uniform(0, 5000, (O, 2))
    calib = np.array([3666.6665, 3666.6665, 2736.0, 1824.0, -0.12, 0.083, -0.0016, -0.00096, -0.012])
    r = kernels.ba_residual(_dev(cams), _dev(pts), _dev(ci), _dev(pi), _dev(uv),
                            _dev(calib)).cpu().numpy()
    rr = cpu_ref.ba_residual(cams, pts, ci, pi, uv, calib[:4], calib[4:])
    assert np.abs(r - rr).max() / np.abs(rr).max() < TIGHT


def test_residual_prepared_equals_plain():
    import torch
    from imageanalysis_amd import ba_solver, synth
    p = synth.make_ba_problem(rows=6, cols=8, n_points=4000, n_obs=30000, seed=3,
                              dist=(-0.12, 0.083, -0.0016, -0.00096, -0.012))
    K = p['K']
    calib = [K[0, 0], K[1, 1], K[0, 2], K[1, 2], *p['dist']]
    prob = ba_solver.DeviceBA(len(p['cams0']), len(p['pts0']), p['cam_idx'], p['pt_idx'], p['uv'],
                              False, fixed_calib=calib)
    prob.set_x(np.hstack([p['cams0'].ravel(), p['pts0'].ravel()]))
    a = prob.residual().clone().cpu().numpy()
    prob.residual_jac()
    b = prob.r.cpu().numpy()
    assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max()
    # ... and, back in the reference's observation order, equal the plain C-ABI residual
    from imageanalysis_amd import kernels
    dev = torch.device('cuda')
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    plain = kernels.ba_residual(t(p['cams0']), t(p['pts0']), t(p['cam_idx']), t(p['pt_idx']),
                                t(p['uv']), t(np.array(calib))).cpu().numpy()
    assert np.abs(prob.download_m(prob.r) - plain).max() <= 1e-9 * np.abs(plain).max()


@pytest.mark.parametrize('n_obs,order', [(1, 'sorted'), (2, 'sorted'), (3, 'sorted'), (129, 'sorted'),
                                         (5001, 'sorted'), (5001, 'shuffled'), (70001, 'wide'),
                                         (200000, 'sorted')])
@pytest.mark.parametrize('form', ['pipe', 'lds'])
def test_residual_prepared_forms_any_order_and_ragged_sizes(n_obs, order, form, monkeypatch):
    """iamx_ba_residual_prepared -- the persistent pipelined walk (default) and the
    one-chain-per-workgroup form -- against the plain-C oracle on odd / tiny observation counts and
    on orders that are NOT camera-major: shuffled (every 128-observation chunk spans hundreds of
    cameras: the walk redoes those chunks per observation) and 'wide' (camera-major, but 12 cameras
    per chunk: more than the walk's LDS slice holds)."""
    import ctypes
    import torch
    from imageanalysis_amd import _lib
    from imageanalysis_amd.kernels import _ptr, stream_ptr
    from oracle import cpu_ref
    monkeypatch.setenv('IAMX_BA_RESIDUAL', form)
    rng = np.random.default_rng(n_obs)
    C, P = 300, 20000
    cams = np.zeros((C, 7))
    cams[:, :3] = rng.normal(0, 50, (C, 3)) * [1, 1, 0.05] + [0, 0, -100]
    cams[:, 3:] = np.array([0.7071, 0, -0.7071, 0]) * rng.uniform(0.5, 2.0, (C, 1)) + rng.normal(0, 0.02, (C, 4))
    pts = rng.normal(0, 40, (P, 3)) * [1, 1, 0.05]
    ci = rng.integers(0, C, n_obs).astype(np.int32)
    if order == 'sorted':
        ci = np.sort(ci)
    elif order == 'wide':
        ci = (np.arange(n_obs) // 11 % C).astype(np.int32)
    pi = rng.integers(0, P, n_obs).astype(np.int32)
    uv = rng.uniform(0, 5000, (n_obs, 2))
    calib = np.array([3666.6665, 3666.6665, 2736.0, 1824.0, -0.12, 0.083, -0.0016, -0.00096, -0.012])
    d = [_dev(a) for a in (cams, pts, ci, pi, uv, calib)]
    r = torch.full((2 * n_obs + 8,), 777.0, dtype=torch.float64, device='cuda')
    _lib.check(_lib.lib().iamx_ba_residual_prepared(_ptr(d[0]), C, _ptr(d[1]), P, _ptr(d[2]), _ptr(d[3]),
                                                    _ptr(d[4]), n_obs, _ptr(d[5]), None, _ptr(r),
                                                    stream_ptr()), 'iamx_ba_residual_prepared')
    got = r.cpu().numpy()
    rr = cpu_ref.ba_residual(cams, pts, ci, pi, uv, calib[:4], calib[4:])
    assert np.abs(got[:2 * n_obs] - rr).max() / np.abs(rr).max() < TIGHT
    assert np.all(got[2 * n_obs:] == 777.0)              # nothing written past the end


def test_config3_full_size_residual_and_operator():
    """BASELINE configs[3] at its full size (2812 cameras, ~270 k points, ~1.96 M observations):
    the residual against the plain-C oracle, and J v / J^T u of the device kernels against a SciPy
    CSR matrix assembled from the analytic blocks, plus one fused (matrix-free) LSMR iteration
    count against the stepwise form -- the sizes bench.py times, checked."""
    import scipy.sparse as sp
    import torch
    from imageanalysis_amd import ba_solver, synth
    from oracle import cpu_ref
    p = synth.make_ba_problem()
    C, P, O = len(p['cams0']), len(p['pts0']), len(p['cam_idx'])
    assert C == 2812 and O > 1900000
    K = p['K']
    calib = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2], *p['dist']])
    prob = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], False, fixed_calib=calib)
    x0 = np.hstack([p['cams0'].ravel(), p['pts0'].ravel()])
    prob.set_x(x0)
    r = prob.download_m(prob.residual())
    rr = cpu_ref.ba_residual(p['cams0'], p['pts0'], p['cam_idx'], p['pt_idx'], p['uv'],
                             calib[:4], calib[4:])
    assert np.abs(r - rr).max() / np.abs(rr).max() < TIGHT
    # J in the reference's row / column order from the blocks of the plain C-ABI entry point
    from imageanalysis_amd import kernels
    rj, Jc, Jp, _ = kernels.ba_residual_jac(_dev(p['cams0']), _dev(p['pts0']), _dev(p['cam_idx']),
                                            _dev(p['pt_idx']), _dev(p['uv']), _dev(calib))
    assert np.abs(rj.cpu().numpy() - rr).max() / np.abs(rr).max() < TIGHT
    Jc, Jp = Jc.cpu().numpy(), Jp.cpu().numpy()
    rows = np.repeat(np.arange(2 * O), 10)
    ci, pi = p['cam_idx'].astype(np.int64), p['pt_idx'].astype(np.int64)
    cols = np.concatenate([ci[:, None] * 7 + np.arange(7), C * 7 + pi[:, None] * 3 + np.arange(3)], 1)
    cols = np.repeat(cols, 2, axis=0).reshape(-1)            # both rows of an observation
    vals = np.concatenate([Jc, Jp], 2).reshape(-1)
    J = sp.csr_matrix((vals, (rows, cols)), shape=(2 * O, prob.n))
    rng = np.random.default_rng(11)
    v = rng.normal(size=prob.n)
    u = rng.normal(size=2 * O)
    prob.residual_jac()
    y = torch.empty(prob.m, dtype=torch.float64, device='cuda')
    prob.jv(prob.upload_n(v), y)
    want = J @ v
    assert np.abs(prob.download_m(y) - want).max() / np.abs(want).max() < 1e-11
    out = torch.empty(prob.n, dtype=torch.float64, device='cuda')
    prob.jtv(prob.upload_m(u), out)
    want = J.T @ u
    assert np.abs(prob.download_n(out) - want).max() / np.abs(want).max() < 1e-11
    # matrix-free fused LSMR == the stepwise LSMR on the stored blocks (same iterates, same stop)
    d = 1.0 / np.sqrt(np.asarray(J.multiply(J).sum(0)).ravel())
    d_dev = prob.upload_n(d).clone()
    dreg = prob.upload_n(np.full(prob.n, 1e-3)).clone()
    xa, istop_a, itn_a, nr_a, nar_a = ba_solver.lsmr_device_fused(prob, d_dev, dreg, maxiter=12)
    xb, istop_b, itn_b, nr_b, nar_b = ba_solver.lsmr_device(prob, d_dev, dreg, maxiter=12)
    assert itn_a == itn_b == 12
    assert np.abs(xa - xb).max() <= 1e-9 * np.abs(xb).max()
    assert abs(nr_a - nr_b) <= 1e-10 * nr_b
