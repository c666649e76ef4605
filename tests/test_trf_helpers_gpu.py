"""GPU: the n-vector kernels of the device-resident TRF loop (csrc/trf_vec.hip, driven through
ba_solver.VecOps) against scipy/optimize/_lsq/common.py itself -- the helpers
scipy.optimize.least_squares(method='trf') runs between two LSMR solves and that the reference
reaches through scripts/lib/optimizer.py:352-399.  Element-wise results are compared bit for
bit (same IEEE operations in the same order), reductions to round-off."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(seed, n=40000):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 5, n)
    lb = np.where(rng.random(n) < 0.4, x - rng.uniform(0, 3, n), -np.inf)
    ub = np.where(rng.random(n) < 0.4, x + rng.uniform(0, 3, n), np.inf)
    on_lo = rng.random(n) < 0.05
    on_up = (rng.random(n) < 0.05) & ~on_lo
    x = np.where(on_lo & np.isfinite(lb), lb, x)           # some points exactly on a bound
    x = np.where(on_up & np.isfinite(ub), ub, x)
    g = rng.normal(0, 1, n)
    g[rng.random(n) < 0.1] = 0.0
    s = rng.normal(0, 1, n)
    s[rng.random(n) < 0.1] = 0.0
    return x, lb, ub, g, s


@pytest.fixture(scope='module')
def V():
    from imageanalysis_amd import _lib
    from imageanalysis_amd.ba_solver import VecOps
    return VecOps(_lib.require_gpu())


def _t(V, *arrs):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(a, np.float64)).to(V.dev) for a in arrs]


def _h(t):
    return t.cpu().numpy()


@pytest.mark.parametrize('seed', range(4))
def test_cl_scaling_vector_and_scaled_quantities(V, seed):
    from scipy.optimize._lsq import common
    x, lb, ub, g, _ = _case(seed)
    v, dv = common.CL_scaling_vector(x, g, lb, ub)
    tx, tlb, tub, tg = _t(V, x, lb, ub, g)
    tv, tdv = V.cl_scaling(tx, tg, tlb, tub)
    assert np.array_equal(_h(tv), v) and np.array_equal(_h(tdv), dv)
    assert V.absmax(tg, tv) == np.linalg.norm(g * v, ord=np.inf)
    # trf.py trf_bounds: v[dv != 0] *= scale_inv; d = v**0.5 * scale; diag_h = g*dv*scale; g_h = d*g
    scale_inv = np.random.default_rng(seed).uniform(0.1, 30, len(x))
    scale = 1 / scale_inv
    v2 = v.copy()
    v2[dv != 0] *= scale_inv[dv != 0]
    d = v2 ** 0.5 * scale
    td, tdh, tgh, tv2 = V.trf_scale(tv, tdv, tg, _t(V, scale_inv)[0], want_v=True)
    assert np.array_equal(_h(tv2), v2) and np.array_equal(_h(td), d)
    assert np.array_equal(_h(tdh), g * dv * scale) and np.array_equal(_h(tgh), d * g)
    assert np.array_equal(_h(V.sqrt_shift(tv, 0.25)), np.sqrt(v + 0.25))


@pytest.mark.parametrize('seed', range(4))
def test_step_size_to_bound_and_reflection(V, seed):
    import torch
    from scipy.optimize._lsq import common
    x, lb, ub, _, s = _case(seed)
    x = common.make_strictly_feasible(x, lb, ub)
    step, hits = common.step_size_to_bound(x, s, lb, ub)
    tx, tlb, tub, ts = _t(V, x, lb, ub, s)
    tstep, thits = V.step_size_to_bound(tx, ts, tlb, tub, want_hits=True)
    assert tstep == step and np.array_equal(_h(thits).astype(int), hits)
    p_h = np.random.default_rng(seed).normal(size=len(x))
    r_h = p_h.copy()
    r_h[hits.astype(bool)] *= -1                                   # trf.py select_step
    assert np.array_equal(_h(V.reflect(tx, ts, tlb, tub, tstep, _t(V, p_h)[0])), r_h)
    # no finite bound in the way: an infinite step
    step, hits = common.step_size_to_bound(x, s, np.full_like(x, -np.inf), np.full_like(x, np.inf))
    tstep, thits = V.step_size_to_bound(tx, ts, torch.full_like(tx, -np.inf), torch.full_like(tx, np.inf),
                                        want_hits=True)
    assert tstep == step == np.inf and np.array_equal(_h(thits).astype(int), hits)


@pytest.mark.parametrize('seed', range(4))
def test_make_strictly_feasible_and_in_bounds(V, seed):
    from scipy.optimize._lsq import common
    x, lb, ub, _, s = _case(seed)
    step = 0.3 * s * (np.random.default_rng(seed).random(len(x)) < 0.2)
    moved = np.clip(x + step, lb, ub)                              # some points onto the bounds
    want = common.make_strictly_feasible(moved, lb, ub, rstep=0)
    tm, tlb, tub = _t(V, moved, lb, ub)
    got = V.strictly_feasible(tm, tlb, tub)
    assert np.array_equal(_h(got), want)
    # x + step inside the kernel
    tx, tstep = _t(V, x, step)
    assert np.array_equal(_h(V.strictly_feasible(tx, tlb, tub, tstep)),
                          common.make_strictly_feasible(x + step, lb, ub, rstep=0))
    assert V.in_bounds(got, tlb, tub) == bool(common.in_bounds(want, lb, ub)) is True
    assert V.in_bounds(tx, tlb, tub, tstep) == bool(common.in_bounds(x + step, lb, ub))
    out = want.copy()
    k = int(np.nonzero(np.isfinite(ub))[0][0])
    out[k] = ub[k] + 1.0
    assert V.in_bounds(_t(V, out)[0], tlb, tub) == bool(common.in_bounds(out, lb, ub)) is False
    # a degenerate box (lb == ub): the midpoint, like SciPy
    lb2, ub2 = lb.copy(), ub.copy()
    lb2[k] = ub2[k] = 1.25
    x2 = want.copy()
    x2[k] = 1.25
    got2 = V.strictly_feasible(*_t(V, x2, lb2, ub2))
    assert np.array_equal(_h(got2), common.make_strictly_feasible(x2, lb2, ub2, rstep=0))


@pytest.mark.parametrize('seed', range(4))
def test_start_point_and_first_radius(V, seed):
    """trf_bounds before its loop: make_strictly_feasible(x0, lb, ub) with the default rstep, and
    Delta = norm(x0 * scale_inv / v**0.5)"""
    from scipy.optimize._lsq import common
    x, lb, ub, g, _ = _case(seed)
    rng = np.random.default_rng(50 + seed)
    # points on, just inside (within rstep) and outside their bounds
    ubf = np.where(np.isfinite(ub), ub, 0.0)
    x = np.where((rng.random(len(x)) < 0.1) & np.isfinite(ub),
                 ubf - rng.uniform(0, 2e-10, len(x)) * np.maximum(1, np.abs(ubf)), x)
    x = np.where(rng.random(len(x)) < 0.05, x + 10.0, x)
    k = int(np.nonzero(np.isfinite(ub) & np.isfinite(lb))[0][0])
    lb[k] = ub[k] = x[k] = -2.5                                    # a degenerate box
    want = common.make_strictly_feasible(x, lb, ub)
    tx, tlb, tub, tg = _t(V, x, lb, ub, g)
    got = V.feasible_start(tx, tlb, tub, 1e-10)
    assert np.array_equal(_h(got), want)
    assert (want != x).sum() > 100
    v, dv = common.CL_scaling_vector(want, g, lb, ub)
    scale_inv = rng.uniform(0.1, 30, len(x))
    v[dv != 0] *= scale_inv[dv != 0]
    t = want * scale_inv / v ** 0.5
    tv, tdv = V.cl_scaling(got, tg, tlb, tub)
    tt = V.scaled_start(got, _t(V, scale_inv)[0], tv, tdv)
    assert np.array_equal(_h(tt), t, equal_nan=True)
    fin = np.isfinite(t)                     # (the degenerate box has v = 0)
    tf = _t(V, np.where(fin, t, 0.0))[0]
    assert fin.sum() >= len(t) - 1
    assert np.isclose(np.sqrt(V.dots((tf, tf))[0]), np.linalg.norm(t[fin]), rtol=1e-13)


@pytest.mark.parametrize('seed', range(4))
def test_find_active_constraints(V, seed):
    from scipy.optimize._lsq import common
    x, lb, ub, _, _ = _case(seed)
    rng = np.random.default_rng(100 + seed)
    near = rng.random(len(x)) < 0.2
    lbf = np.where(np.isfinite(lb), lb, 0.0)
    x = np.where(near & np.isfinite(lb), lbf + rng.uniform(0, 2e-8, len(x)) * np.maximum(1, np.abs(lbf)), x)
    want = common.find_active_constraints(x, lb, ub, rtol=1e-8)
    got = V.active_constraints(*_t(V, x, lb, ub), 1e-8)
    assert np.array_equal(_h(got).astype(int), want)
    assert (want != 0).any() and (want == 0).any()


def test_linear_combinations_products_sums_and_jac_scale(V):
    rng = np.random.default_rng(9)
    n = 123457                                                     # not a multiple of anything
    a, b, c, w = (rng.normal(size=n) for _ in range(4))
    ta, tb, tc, tw = _t(V, a, b, c, w)
    assert np.array_equal(_h(V.lincomb(2.5, ta)), 2.5 * a)
    assert np.array_equal(_h(V.lincomb(2.5, ta, -0.5, tb)), 2.5 * a + -0.5 * b)
    assert np.array_equal(_h(V.lincomb(2.5, ta, -0.5, tb, 3.0, tc)), 2.5 * a + -0.5 * b + 3.0 * c)
    assert np.array_equal(_h(V.mul(ta, tb)), 1.0 * a * b) and np.array_equal(_h(V.mul(ta, s=-2.0)), -2.0 * a)
    out = V.mul(ta, tb, s=0.5, out=V.new(ta))
    assert np.array_equal(_h(out), 0.5 * a * b)
    got = V.dots((ta, tb), (ta, tw, tb), (tc, tc), (ta, ta), (tb, tb), (tw, tw), (ta, tc), (tb, tc))
    want = [a @ b, (a * w) @ b, c @ c, a @ a, b @ b, w @ w, a @ c, b @ c]
    assert np.allclose(got, want, rtol=1e-12, atol=1e-9)
    assert V.dots((ta, tb)) == V.dots((ta, tb))                    # fixed tree: reproducible
    assert V.dots((ta, tb), n=1000)[0] == pytest.approx(a[:1000] @ b[:1000], rel=1e-12)
    assert V.absmax(ta) == np.abs(a).max() and V.absmax(ta, tb) == np.abs(a * b).max()
    bad = a.copy()
    bad[77777] = np.nan
    assert np.isnan(V.absmax(_t(V, bad)[0]))
    # compute_jac_scale: first call (zeros -> 1), later calls keep the maximum
    colsq = rng.uniform(0, 4, n)
    colsq[::7] = 0.0
    import torch
    si = V.jac_scale(_t(V, colsq)[0], torch.empty(n, dtype=torch.float64, device=V.dev), first=True)
    want = colsq ** 0.5
    want[want == 0] = 1
    assert np.array_equal(_h(si), want)
    colsq2 = rng.uniform(0, 4, n)
    V.jac_scale(_t(V, colsq2)[0], si, first=False)
    assert np.array_equal(_h(si), np.maximum(want, colsq2 ** 0.5))
