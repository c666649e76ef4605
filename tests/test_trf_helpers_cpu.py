"""CPU: the torch restatements of SciPy's O(n) trust-region helpers used by the device-resident
TRF loop (imageanalysis_amd/ba_solver.py _trf_device) against scipy/optimize/_lsq/common.py
itself, on CPU tensors -- the same code runs on device tensors in the solver."""
import numpy as np
import pytest
import torch
from scipy.optimize._lsq import common

from imageanalysis_amd import ba_solver as bs


def _case(seed, n=4000):
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


def _t(*arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)) for a in arrs]


@pytest.mark.parametrize('seed', range(5))
def test_cl_scaling_vector(seed):
    x, lb, ub, g, _ = _case(seed)
    v, dv = common.CL_scaling_vector(x, g, lb, ub)
    tx, tlb, tub, tg = _t(x, lb, ub, g)
    tv, tdv = bs._cl_scaling_dev(tx, tg, tlb, tub, torch.isfinite(tlb), torch.isfinite(tub))
    assert np.array_equal(tv.numpy(), v) and np.array_equal(tdv.numpy(), dv)


@pytest.mark.parametrize('seed', range(5))
def test_step_size_to_bound(seed):
    x, lb, ub, _, s = _case(seed)
    x = common.make_strictly_feasible(x, lb, ub)
    step, hits = common.step_size_to_bound(x, s, lb, ub)
    tx, tlb, tub, ts = _t(x, lb, ub, s)
    tstep, thits = bs._step_size_to_bound_dev(tx, ts, tlb, tub)
    assert tstep == step and np.array_equal(thits.numpy().astype(int), hits)
    # no finite bound in the way: an infinite step, no hits
    step, hits = common.step_size_to_bound(x, s, np.full_like(x, -np.inf), np.full_like(x, np.inf))
    tstep, thits = bs._step_size_to_bound_dev(tx, ts, torch.full_like(tx, -np.inf), torch.full_like(tx, np.inf))
    assert tstep == step == np.inf and np.array_equal(thits.numpy().astype(int), hits)


@pytest.mark.parametrize('seed', range(5))
def test_make_strictly_feasible_and_in_bounds(seed):
    x, lb, ub, _, s = _case(seed)
    x = x + 0.3 * s * (np.random.default_rng(seed).random(len(x)) < 0.2)   # some points outside
    x = np.clip(x, lb, ub)                                                  # -> onto the bounds
    want = common.make_strictly_feasible(x, lb, ub, rstep=0)
    tx, tlb, tub = _t(x, lb, ub)
    got = bs._strictly_feasible_dev(tx, tlb, tub, torch.isfinite(tlb), torch.isfinite(tub))
    assert np.array_equal(got.numpy(), want)
    assert bs._in_bounds_dev(got, tlb, tub) == bool(common.in_bounds(want, lb, ub)) is True
    out = want.copy()
    k = int(np.nonzero(np.isfinite(ub))[0][0])
    out[k] = ub[k] + 1.0
    assert bs._in_bounds_dev(torch.from_numpy(out), tlb, tub) == bool(common.in_bounds(out, lb, ub)) is False
    # a degenerate box (lb == ub): the midpoint, like SciPy
    lb2, ub2 = lb.copy(), ub.copy()
    lb2[k] = ub2[k] = 1.25
    x2 = want.copy()
    x2[k] = 1.25
    got2 = bs._strictly_feasible_dev(torch.from_numpy(x2), *_t(lb2, ub2), torch.isfinite(torch.from_numpy(lb2)),
                                     torch.isfinite(torch.from_numpy(ub2)))
    assert np.array_equal(got2.numpy(), common.make_strictly_feasible(x2, lb2, ub2, rstep=0))


@pytest.mark.parametrize('seed', range(5))
def test_find_active_constraints(seed):
    x, lb, ub, _, _ = _case(seed)
    rng = np.random.default_rng(100 + seed)
    near = rng.random(len(x)) < 0.2
    lbf = np.where(np.isfinite(lb), lb, 0.0)
    x = np.where(near & np.isfinite(lb), lbf + rng.uniform(0, 2e-8, len(x)) * np.maximum(1, np.abs(lbf)), x)
    want = common.find_active_constraints(x, lb, ub, rtol=1e-8)
    tx, tlb, tub = _t(x, lb, ub)
    got = bs._active_constraints_dev(tx, tlb, tub, torch.isfinite(tlb), torch.isfinite(tub), 1e-8)
    assert np.array_equal(got.numpy().astype(int), want)
    assert (want != 0).any() and (want == 0).any()
