"""GPU: iamx_ba_accumulate (U, V, g_c, g_p) and the Schur-complement subproblem solver
(csrc/ba_schur.hip) against numpy / SciPy on the reference-derived BA goldens and at
BASELINE configs[3] size."""
import glob
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO
from test_host_logic import _scene
from test_ba_solver_gpu import _problem

pytestmark = pytest.mark.gpu
BA_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'ba_*.npz')))
PLAIN = [p for p in BA_CASES if not bool(np.load(p)['cam_calib'])]


def _blocks(prob):
    """Jc [O,2,7], Jp [O,2,3], r [O,2], cam, pt (internal order) on the host"""
    O = prob.O
    Jc = prob.download(prob.Jc, O * 14).reshape(O, 2, 7)
    Jp = prob.download(prob.Jp, O * 6).reshape(O, 2, 3)
    r = prob.download(prob.r, 2 * O).reshape(O, 2)
    return Jc, Jp, r, prob.cam_idx.cpu().numpy().astype(np.int64), prob.pt_idx.cpu().numpy().astype(np.int64)


@pytest.mark.parametrize('path', PLAIN, ids=os.path.basename)
def test_accumulate_equals_numpy_blocks(path):
    g, opt, prob = _problem(path)
    prob.set_x(g['x0'])
    prob.residual_jac()
    a = prob.accumulate()
    C, P = prob.C, prob.P
    Jc, Jp, r, cam, pt = _blocks(prob)
    U = np.zeros((C, 7, 7)); V = np.zeros((P, 3, 3)); gc = np.zeros((C, 7)); gp = np.zeros((P, 3))
    np.add.at(U, cam, np.einsum('oki,okj->oij', Jc, Jc))
    np.add.at(V, pt, np.einsum('oki,okj->oij', Jp, Jp))
    np.add.at(gc, cam, np.einsum('oki,ok->oi', Jc, r))
    np.add.at(gp, pt, np.einsum('oki,ok->oi', Jp, r))
    got_U = prob.download(a['U'], C * 49).reshape(C, 7, 7)
    got_V = prob.download(a['V'], P * 9).reshape(P, 3, 3)
    got_g = prob.download(a['g'], prob.n)
    assert np.abs(got_U - U).max() <= 1e-12 * np.abs(U).max()
    assert np.abs(got_V - V).max() <= 1e-12 * np.abs(V).max()
    assert np.array_equal(got_U, got_U.transpose(0, 2, 1)) and np.array_equal(got_V, got_V.transpose(0, 2, 1))
    ref_g = np.concatenate([gc.ravel(), gp.ravel()])
    assert np.abs(got_g - ref_g).max() <= 1e-12 * np.abs(ref_g).max()
    # ... and they are what the outer iteration reads: J^T r and the column sums of J.^2 (the
    # reference's order) equal the products with the CSR Jacobian
    args = (opt.n_cameras, opt.n_points, opt.by_camera_point_indices, opt.by_camera_points_2d)
    J = opt.jac(g['x0'], *args)
    ref = J.T @ g['f0']
    assert np.abs(prob.download_n(prob.grad_dev()) - ref).max() <= 1e-9 * np.abs(ref).max()
    ref = np.asarray(J.power(2).sum(axis=0)).ravel()
    assert np.abs(prob.download_n(prob.colsq_dev()) - ref).max() <= 1e-12 * ref.max()


@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
def test_schur_step_equals_direct_solve(path):
    """the step of the Schur-complement solver, iterated to convergence, is the least-squares
    solution of SciPy's subproblem  min ||[J D; Dreg] p - [r; 0]||  (trf.py:303-314) -- also with
    the 8 calibration columns of optimize_calib='global' (the bordered form)"""
    from scipy.sparse import diags, vstack
    from scipy.sparse.linalg import spsolve
    from imageanalysis_amd import ba_solver
    g, opt, prob = _problem(path)
    x0 = g['x0']
    args = (opt.n_cameras, opt.n_points, opt.by_camera_point_indices, opt.by_camera_points_2d)
    J = opt.jac(x0, *args).tocsr()
    prob.set_x(x0)
    prob.residual_jac()
    rng = np.random.default_rng(1)
    cn = np.sqrt(np.asarray(J.power(2).sum(axis=0)).ravel())
    d = (1.0 / cn) * rng.uniform(0.5, 2.0, prob.n)
    dreg = rng.uniform(1e-3, 3e-2, prob.n)
    A = vstack([J @ diags(d), diags(dreg)]).tocsc()
    rhs = A.T @ np.concatenate([g['f0'], np.zeros(prob.n)])
    ref = spsolve((A.T @ A).tocsc(), rhs)
    dd, dr = prob.upload_n(d), prob.upload_n(dreg)
    step, istop, itn, _rz, _ = ba_solver.schur_solve(prob, dd, dr, eta=1e-13, maxiter=2000)
    assert istop in (1, 3) and itn > 0          # converged (3: p.Sp underflowed at the solution)
    assert np.abs(step - ref).max() <= 2e-7 * np.abs(ref).max(), (itn, np.abs(step - ref).max())
    # the forcing term bounds the work: a loose tolerance stops early with a descent direction
    step2, istop2, itn2, _, _ = ba_solver.schur_solve(prob, dd, dr, eta=0.1)
    assert istop2 == 1 and 0 < itn2 < itn
    gh = d * (J.T @ g['f0'])
    assert step2 @ gh > 0                         # (p solves A^T A p = A^T [r; 0] = g_h)
    # bitwise reproducible (no atomics)
    step3 = ba_solver.schur_solve(prob, dd, dr, eta=0.1)[0]
    assert np.array_equal(step2, step3)


@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
@pytest.mark.parametrize('solver', ['device', 'device-lsmr'])
def test_both_inner_solvers_reach_reference_minimum(path, solver):
    from imageanalysis_amd import optimizer
    g = np.load(path)
    proj, inp = _scene(path)
    opt = optimizer.Optimizer('/nonexistent')
    opt.solver = solver
    opt.setup(proj, inp['groups'], 0, inp['matches'], cam_calib=bool(g['cam_calib']))
    opt.run()
    res = opt.result
    cost = 0.5 * float(res.fun @ res.fun)
    ref = float(g['cost_final'])
    assert abs(cost - ref) / ref < 2e-2, (cost, ref)
    assert abs(np.mean(np.abs(res.fun)) - np.mean(np.abs(g['f_final']))) < 0.02
    assert res.njev <= 3 * 8 and res.status in (1, 2, 3, 4)
    # (the calibration mode has the Schur path too: the bordered form)
    assert res.inner_solver == ('lsmr' if solver == 'device-lsmr' else 'schur')


def test_config3_schur_inner_iterations_and_end_state():
    """BASELINE configs[3] at full size: the Schur solver needs <= 80 inner iterations per outer
    iteration (LSMR on the whole system: ~400) and both inner solvers end at the same cost"""
    from imageanalysis_amd import ba_solver, synth
    p = synth.make_ba_problem()
    C, P = len(p['cams0']), len(p['pts0'])
    K = p['K']
    calib = [K[0, 0], K[1, 1], K[0, 2], K[1, 2], *p['dist']]
    x0 = np.hstack([p['cams0'].ravel(), p['pts0'].ravel()])
    lb = np.full(x0.size, -np.inf)
    ub = np.full(x0.size, np.inf)
    for j, dlt in ((0, 3.0), (1, 3.0), (2, 9.0)):
        lb[j:C * 7:7] = p['cams0'][:, j] - dlt
        ub[j:C * 7:7] = p['cams0'][:, j] + dlt
    out = {}
    for inner in ('schur', 'lsmr'):
        prob = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], False, fixed_calib=calib)
        prob.inner = inner
        res = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4)
        out[inner] = (res, list(prob.inner_iterations))
    res, its = out['schur']
    assert res.status in (1, 2, 3, 4)
    assert max(its) <= 80 and np.mean(its) <= 40, its
    mre = np.sqrt(2 * res.cost / (2 * len(p['uv'])))
    assert mre < 0.46                                   # pixel noise 0.5 px, fitted: the noise floor
    # the reference's formulation (LSMR on the whole system, atol = btol = 1e-6) run to its own
    # ftol stop: ~10x the inner iterations per outer iteration, and an end state that is no better
    res_l, its_l = out['lsmr']
    assert res_l.status in (1, 2, 3, 4)
    assert np.mean(its_l) > 5 * np.mean(its)
    assert res.cost <= 1.005 * res_l.cost, (res.cost, res_l.cost)
    assert abs(np.sqrt(res.cost / len(p['uv'])) - np.sqrt(res_l.cost / len(p['uv']))) < 0.02


def test_config3_with_calibration_columns_bordered_schur():
    """`process.py --cam-calibration` at BASELINE configs[3] size: optimize_calib='global' adds 8
    dense columns (scripts/lib/optimizer.py:142-169,181-189).  The bordered Schur solve takes a
    fraction of the inner iterations of LSMR on the whole system and ends at least as low."""
    from imageanalysis_amd import ba_solver, synth
    p = synth.make_ba_problem()
    C, P = len(p['cams0']), len(p['pts0'])
    K = p['K']
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
    out = {}
    for inner in ('schur', 'lsmr'):
        prob = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], True)
        prob.inner = inner
        res = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4, max_nfev=40 if inner == 'lsmr' else None)
        out[inner] = (res, list(prob.inner_iterations))
    res, its = out['schur']
    res_l, its_l = out['lsmr']
    assert res.status in (1, 2, 3, 4) and max(its) <= 150 and res.iterations <= 25, (its, res.iterations)
    assert np.sqrt(res.cost / len(p['uv'])) < 0.46              # the 0.5 px noise floor, fitted
    assert np.mean(its_l) > 3 * np.mean(its)
    assert res.cost <= 1.005 * res_l.cost, (res.cost, res_l.cost)
    assert np.all(res.x >= lb) and np.all(res.x <= ub)


def _two_rank_schur(rank, world, port, path, outdir):
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
    opt.solver = 'device'
    opt.setup(proj, inp['groups'], 0, inp['matches'], cam_calib=bool(g['cam_calib']))
    from imageanalysis_amd import ba_solver, dist as D
    sizes = []
    plain = D.allreduce_sum_

    def counting(t, group=None):
        sizes.append(int(t.numel()))
        return plain(t, group)
    D.allreduce_sum_ = ba_solver._dist.allreduce_sum_ = counting
    plain_op = D.allreduce_

    def counting_op(t, op, group=None):
        sizes.append(int(t.numel()))
        return plain_op(t, op, group)
    D.allreduce_ = ba_solver._dist.allreduce_ = counting_op
    inner = []
    real = ba_solver.schur_solve

    def solve(prob, *a, **k):
        out = real(prob, *a, **k)
        inner.append(out[2])
        return out
    ba_solver.schur_solve = solve
    opt.run()
    np.save(os.path.join(outdir, 'x_r%d.npy' % rank), opt.result.x)
    np.save(os.path.join(outdir, 'f_r%d.npy' % rank), opt.result.fun)
    np.save(os.path.join(outdir, 'red_r%d.npy' % rank), np.array(sizes, np.int64))
    np.save(os.path.join(outdir, 'inner_r%d.npy' % rank), np.array(inner, np.int64))
    dist.barrier()
    dist.destroy_process_group()


def test_schur_two_ranks_point_sharded_and_its_reduce_schedule(tmp_path):
    """observations sharded by point over 2 ranks (gloo, same GPU): same solution as 1 rank; the
    inner solve all-reduces C x 35 doubles once and C x 7 doubles per CG iteration (chunks of 4
    are enqueued ahead: a few no-op iterations behind the latched stop are reduced as well); the
    outer iteration reduces the camera part of the gradient and of the column sums (C x 7 each)
    and scalars -- the point parts stay on their ranks and are put together once, at the end"""
    import torch.multiprocessing as mp
    from imageanalysis_amd import optimizer
    path = [p for p in BA_CASES if p.endswith('ba_mid.npz')][0]
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_schur, args=(2, port, path, str(tmp_path)), nprocs=2, join=True)
    g = np.load(path)
    proj, inp = _scene(path)
    opt = optimizer.Optimizer('/nonexistent')
    opt.solver = 'device'
    opt.setup(proj, inp['groups'], 0, inp['matches'])
    opt.run()
    x0 = np.load(tmp_path / 'x_r0.npy')
    x1 = np.load(tmp_path / 'x_r1.npy')
    assert np.array_equal(x0, x1)
    f0 = np.load(tmp_path / 'f_r0.npy')
    c1, c2 = 0.5 * f0 @ f0, 0.5 * opt.result.fun @ opt.result.fun
    assert abs(c1 - c2) / c2 < 1e-3
    C, n = opt.n_cameras, opt.result.x.size
    for r in range(2):
        red = np.load(tmp_path / ('red_r%d.npy' % r))
        inner = np.load(tmp_path / ('inner_r%d.npy' % r))
        solves = len(inner)
        njev = opt.result.njev
        assert solves >= 2 and (red == 35 * C).sum() == solves
        n_q = int((red == 7 * C).sum())                    # CG iterations + gradient / column sums
        assert inner.sum() + 2 <= n_q <= inner.sum() + 8 * solves + 2 * (njev + 3)
        # no n-vector and no point part inside the outer loop: x, g, the active set once at the end
        big = red[red > 35 * C]
        assert big.tolist() == [n - 7 * C] * 3 and (red == n).sum() == 0
        assert np.all(red[-3:] == n - 7 * C) or np.all(np.sort(red[-8:])[-3:] == n - 7 * C)


def _two_rank_config3(rank, world, port, outdir):
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)                         # both ranks share the one GPU of the box
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from imageanalysis_amd import ba_solver, dist as D, synth
    p = synth.make_ba_problem()
    C, P = len(p['cams0']), len(p['pts0'])
    K = p['K']
    calib = [K[0, 0], K[1, 1], K[0, 2], K[1, 2], *p['dist']]
    x0 = np.hstack([p['cams0'].ravel(), p['pts0'].ravel()])
    lb = np.full(x0.size, -np.inf)
    ub = np.full(x0.size, np.inf)
    for j, dlt in ((0, 3.0), (1, 3.0), (2, 9.0)):
        lb[j:C * 7:7] = p['cams0'][:, j] - dlt
        ub[j:C * 7:7] = p['cams0'][:, j] + dlt
    sizes = []
    plain = D.allreduce_sum_

    def counting(t, group=None):
        sizes.append(int(t.numel()))
        return plain(t, group)
    D.allreduce_sum_ = ba_solver._dist.allreduce_sum_ = counting
    plain_op = D.allreduce_

    def counting_op(t, op, group=None):
        sizes.append(int(t.numel()))
        return plain_op(t, op, group)
    D.allreduce_ = ba_solver._dist.allreduce_ = counting_op
    prob = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], False, fixed_calib=calib,
                              rank=rank, world=world)
    res = ba_solver.trf_device(prob, x0, lb, ub, ftol=1e-4)
    np.save(os.path.join(outdir, 'x_r%d.npy' % rank), res.x)
    np.save(os.path.join(outdir, 'stat_r%d.npy' % rank),
            np.array([res.cost, res.njev, res.status, prob.O, sum(prob.inner_iterations),
                      len(prob.inner_iterations), prob.pt_lo, prob.pt_hi]))
    np.save(os.path.join(outdir, 'red_r%d.npy' % rank), np.array(sizes, np.int64))
    dist.barrier()
    dist.destroy_process_group()


def test_config3_full_size_two_ranks_point_sharded(tmp_path):
    """BASELINE configs[3] (2812 cameras, 271 k points, 1.96 M observations) with the
    observations sharded by point over 2 ranks (gloo, both on the one GPU): every rank ends with
    the same parameters, at the cost one rank reaches, and all-reduces C x 35 doubles per outer
    iteration + C x 7 per CG iteration -- never an n-vector inside the inner solve"""
    import torch.multiprocessing as mp
    from imageanalysis_amd import ba_solver, synth
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_config3, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    x0, x1 = np.load(tmp_path / 'x_r0.npy'), np.load(tmp_path / 'x_r1.npy')
    assert np.array_equal(x0, x1)
    st0, st1 = np.load(tmp_path / 'stat_r0.npy'), np.load(tmp_path / 'stat_r1.npy')
    p = synth.make_ba_problem()
    C, P, O = len(p['cams0']), len(p['pts0']), len(p['cam_idx'])
    n = 7 * C + 3 * P
    assert st0[3] + st1[3] == O and abs(st0[3] - st1[3]) < 0.02 * O        # balanced by observations
    assert st0[6] == 0 and st0[7] == st1[6] and st1[7] == P                # contiguous point blocks
    assert st0[0] == st1[0] and st0[2] in (1, 2, 3, 4)
    assert np.sqrt(st0[0] / O) < 0.46                                      # the noise floor (rms px)
    # the single-rank solve of the same problem: same end state (partial sums round differently)
    K = p['K']
    prob = ba_solver.DeviceBA(C, P, p['cam_idx'], p['pt_idx'], p['uv'], False,
                              fixed_calib=[K[0, 0], K[1, 1], K[0, 2], K[1, 2], *p['dist']])
    xs = np.hstack([p['cams0'].ravel(), p['pts0'].ravel()])
    lb = np.full(xs.size, -np.inf)
    ub = np.full(xs.size, np.inf)
    for j, dlt in ((0, 3.0), (1, 3.0), (2, 9.0)):
        lb[j:C * 7:7] = p['cams0'][:, j] - dlt
        ub[j:C * 7:7] = p['cams0'][:, j] + dlt
    one = ba_solver.trf_device(prob, xs, lb, ub, ftol=1e-4)
    assert abs(st0[0] - one.cost) < 5e-3 * one.cost
    for r in range(2):
        red = np.load(tmp_path / ('red_r%d.npy' % r))
        st = np.load(tmp_path / ('stat_r%d.npy' % r))
        solves, inner = int(st[5]), int(st[4])
        njev = int(st[1])
        assert (red == 35 * C).sum() == solves
        assert inner + 2 <= (red == 7 * C).sum() <= inner + 8 * solves + 2 * (njev + 3)
        # inside the outer loop nothing larger than the C x 35 camera blocks crosses the ranks:
        # the point parts of x, the gradient and the active set are put together once, at the end
        assert red[red > 35 * C].tolist() == [3 * P] * 3 and (red == n).sum() == 0
        assert set(np.unique(red)) <= {1, 2, 3, 4, 5, 6, 7, 8, 35 * C, 7 * C, 3 * P}
