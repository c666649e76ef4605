"""Grid-based motion statistics filter (GMS) -- host side, vectorised numpy.

Takes the place of cv2.xfeatures2d.matchGMS(size, size, kp1, kp2, matches,
withRotation=True, withScale=False, thresholdFactor=5.0) in the reference's
scripts/lib/matcher.py:285.  Algorithm as in the reference's own pure-python port
scripts/lib/archive/gms_matcher.py:74-285 (20x20 grids, 4 half-cell shifted left grids,
8 rotation patterns of the 3x3 neighbourhood, threshold factor x sqrt(mean cell count)),
computed for all 400 cells at once instead of cell by cell.  SURVEY.md 8f rank 3 moves this
to the device; per pair it is <= 2000 matches, so the host version is not on the M1 clock.
"""
import math

import numpy as np

GRID = 20
NCELL = GRID * GRID

_ROT = np.array([[1, 2, 3, 4, 5, 6, 7, 8, 9],
                 [4, 1, 2, 7, 5, 3, 8, 9, 6],
                 [7, 4, 1, 8, 5, 2, 9, 6, 3],
                 [8, 7, 4, 9, 5, 1, 6, 3, 2],
                 [9, 8, 7, 6, 5, 4, 3, 2, 1],
                 [6, 9, 8, 3, 5, 7, 2, 1, 4],
                 [3, 6, 9, 2, 5, 8, 1, 4, 7],
                 [2, 3, 6, 1, 5, 9, 4, 7, 8]], np.int64) - 1
_SCALES = (1.0, 0.5, 1.0 / math.sqrt(2.0), math.sqrt(2.0), 2.0)


def _nb9(gw, gh):
    idx = np.arange(gw * gh)
    x, y = idx % gw, idx // gw
    nb = -np.ones((gw * gh, 9), np.int64)
    for yi in (-1, 0, 1):
        for xi in (-1, 0, 1):
            xx, yy = x + xi, y + yi
            ok = (xx >= 0) & (xx < gw) & (yy >= 0) & (yy < gh)
            nb[ok, xi + 4 + yi * 3] = (xx + yy * gw)[ok]
    return nb


_NB_LEFT = _nb9(GRID, GRID)
_NB_CACHE = {(GRID, GRID): _NB_LEFT}


def _nb(gw, gh):
    if (gw, gh) not in _NB_CACHE:
        _NB_CACHE[(gw, gh)] = _nb9(gw, gh)
    return _NB_CACHE[(gw, gh)]


def gms_inlier_mask(xy1, xy2, size1, size2, pairs, with_rotation=True, with_scale=False,
                    threshold_factor=5.0):
    """xy1/xy2: [N,2] float32 keypoint pixels; size = (width, height); pairs [n,2] int
    (queryIdx, trainIdx) in the order handed to matchGMS.  Returns a bool mask [n]."""
    pairs = np.asarray(pairs, np.int64).reshape(-1, 2)
    n = len(pairs)
    if n == 0:
        return np.zeros(0, bool)
    lp = np.asarray(xy1, np.float64)[pairs[:, 0]] / np.array(size1, np.float64)
    rp = np.asarray(xy2, np.float64)[pairs[:, 1]] / np.array(size2, np.float64)

    # the four left grids (plain, +x half cell, +y half cell, both); -1 = outside
    lgs = []
    for ox, oy in ((0.0, 0.0), (0.5, 0.0), (0.0, 0.5), (0.5, 0.5)):
        x = np.floor(lp[:, 0] * GRID + ox).astype(np.int64)
        y = np.floor(lp[:, 1] * GRID + oy).astype(np.int64)
        lgs.append(np.where((x >= GRID) | (y >= GRID), -1, x + y * GRID))

    best, best_n, last = None, 0, np.zeros(n, bool)
    for s in (range(5) if with_scale else (0,)):
        gwr, ghr = int(GRID * _SCALES[s]), int(GRID * _SCALES[s])
        nr = gwr * ghr
        nb_r = _nb(gwr, ghr)
        rg = (np.floor(rp[:, 0] * gwr).astype(np.int64)
              + np.floor(rp[:, 1] * ghr).astype(np.int64) * gwr)
        per_grid = []
        for lg in lgs:
            ok = (lg >= 0) & (rg >= 0)
            stats = np.bincount(lg[ok] * nr + rg[ok], minlength=NCELL * nr).reshape(NCELL, nr)
            cnt = stats.sum(1)
            per_grid.append((lg, stats, cnt, np.argmax(stats, axis=1)))
        for r in (range(8) if with_rotation else (0,)):
            mask = np.zeros(n, bool)
            for lg, stats, cnt, jbest in per_grid:
                rr = nb_r[jbest][:, _ROT[r]]                       # [400,9] right neighbours
                valid = (_NB_LEFT >= 0) & (rr >= 0)
                ll = np.where(valid, _NB_LEFT, 0)
                score = np.where(valid, stats[ll, np.where(valid, rr, 0)], 0).sum(1)
                tot = np.where(valid, cnt[ll], 0).sum(1).astype(np.float64)
                npair = np.maximum(valid.sum(1), 1)
                thresh = threshold_factor * np.sqrt(tot / npair)
                cell = np.where(cnt == 0, -1, np.where(score < thresh, -2, jbest))
                mask |= (lg >= 0) & (cell[np.maximum(lg, 0)] == rg)
            last = mask
            c = int(mask.sum())
            if c > best_n:
                best, best_n = mask, c
    return best if best is not None else last
