"""ORACLE / TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's pair
matching path in numpy.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product (imageanalysis_amd/) never does.

Pinned against tests/golden/match_*.npz, which were produced by running the
reference's own scripts/lib/matcher.py here (oracle/gen_golden.py).  The
third-party arithmetic under it (cv2 knnMatch, cv2 matchGMS) is not available
in this container: knnMatch is restated as exact brute force (SURVEY.md section 0
fact 5) and GMS follows the reference's archived pure-Python port, so those two
are "parity pinned to the reference's Python, cv2 natives unpinned".

Every function cites the reference lines it follows (paths relative to
/root/reference/).
"""
import math

import numpy as np

MAX_DISTANCE_SIFT = 270.0          # scripts/lib/matcher.py:53
MYMAX = 2000                       # scripts/lib/matcher.py:265


# ---------------------------------------------------------------------------
# scripts/lib/matcher.py:203-216 raw_matches -> the_matcher.knnMatch(des1, des2, k=2)
# ---------------------------------------------------------------------------
def knn2_l2(des_q, des_t):
    """Exact 2-NN in squared L2 over integer-valued descriptors.

    Returns (idx[nq,2] int32, d2[nq,2] int32), nearest first; equal distances
    resolved by the lowest train index (the oracle's fixed rule; cv2's own tie
    order is unpinned).  cv2's DMatch.distance is float32(sqrt(float32(d2))).
    """
    q = np.asarray(des_q).astype(np.float64)
    t = np.asarray(des_t).astype(np.float64)
    nq, nt = q.shape[0], t.shape[0]
    assert nt >= 2
    nt2 = (t * t).sum(1)
    idx = np.empty((nq, 2), np.int32)
    d2o = np.empty((nq, 2), np.int32)
    step = 1024
    for s in range(0, nq, step):
        qs = q[s:s + step]
        d2 = (qs * qs).sum(1)[:, None] + nt2[None, :] - 2.0 * (qs @ t.T)   # exact integers
        i0 = np.argmin(d2, axis=1)                     # first (= lowest index) minimum
        r = np.arange(qs.shape[0])
        m0 = d2[r, i0].copy()
        d2[r, i0] = np.inf
        i1 = np.argmin(d2, axis=1)
        idx[s:s + step, 0] = i0
        idx[s:s + step, 1] = i1
        d2o[s:s + step, 0] = m0
        d2o[s:s + step, 1] = d2[r, i1]
    return idx, d2o


def distances_f32(d2):
    """cv2 L2 distance as the reference sees it: float32 sqrt of the exact sum."""
    return np.sqrt(np.asarray(d2).astype(np.float32))


# ---------------------------------------------------------------------------
# scripts/lib/matcher.py:253-269 quality metric, stable sort, threshold, clip
# ---------------------------------------------------------------------------
def metric_filter(idx, d2, match_ratio, max_distance=MAX_DISTANCE_SIFT, mymax=MYMAX):
    """Returns the ordered [(queryIdx, trainIdx)] list handed to matchGMS."""
    dist = distances_f32(d2).astype(np.float64)         # python floats of f32 values
    if np.any(dist[:, 1] == 0.0):
        raise ZeroDivisionError("float division by zero")       # matcher.py:255
    ratio = dist[:, 0] / dist[:, 1]
    metric = dist[:, 0] * ratio
    order = np.argsort(metric, kind='stable')           # sorted(..., key=metric) is stable
    keep = order[metric[order] < max_distance * match_ratio]
    keep = keep[:mymax]
    return np.stack([keep.astype(np.int32), idx[keep, 0].astype(np.int32)], axis=1)


def lowe_stats(d2, match_ratio):
    """scripts/lib/matcher.py:221-235 (log-only statistics)."""
    dist = distances_f32(d2).astype(np.float64)
    good = dist[:, 0] <= dist[:, 1] * match_ratio
    return dict(avg=float(dist[:, 0].mean()), count_good=int(good.sum()),
                max_good=float(dist[good, 0].max()) if good.any() else 0.0)


# ---------------------------------------------------------------------------
# GMS -- scripts/lib/archive/gms_matcher.py:74-285 (the live call at
# scripts/lib/matcher.py:285: withRotation=True, withScale=False, thresholdFactor=5.0)
# ---------------------------------------------------------------------------
_ROT = np.array([[1, 2, 3, 4, 5, 6, 7, 8, 9],
                 [4, 1, 2, 7, 5, 3, 8, 9, 6],
                 [7, 4, 1, 8, 5, 2, 9, 6, 3],
                 [8, 7, 4, 9, 5, 1, 6, 3, 2],
                 [9, 8, 7, 6, 5, 4, 3, 2, 1],
                 [6, 9, 8, 3, 5, 7, 2, 1, 4],
                 [3, 6, 9, 2, 5, 8, 1, 4, 7],
                 [2, 3, 6, 1, 5, 9, 4, 7, 8]]) - 1      # gms_matcher.py:29-60
_SCALES = [1.0, 0.5, 1.0 / math.sqrt(2.0), math.sqrt(2.0), 2.0]   # :63


def _neighbours(gw, gh):
    """gms_matcher.py:110-125 get_nb9: 3x3 neighbourhood, -1 outside."""
    nb = -np.ones((gw * gh, 9), np.int64)
    for idx in range(gw * gh):
        x, y = idx % gw, idx // gw
        for yi in (-1, 0, 1):
            for xi in (-1, 0, 1):
                xx, yy = x + xi, y + yi
                if 0 <= xx < gw and 0 <= yy < gh:
                    nb[idx, xi + 4 + yi * 3] = xx + yy * gw
    return nb


def gms_inlier_mask(xy1, xy2, size1, size2, pairs, with_rotation=True, with_scale=False,
                    threshold_factor=5.0, port_negative_index=False):
    """Boolean inlier mask over `pairs` ([n,2] (queryIdx, trainIdx)).

    port_negative_index=True reproduces one quirk of the archived Python port
    (gms_matcher.py:199: a left cell index of -1 indexes the LAST cell through
    Python's negative indexing).  The default skips such matches the way
    OpenCV's C++ does; it only matters for keypoints in the outermost half cell.
    """
    pairs = np.asarray(pairs, np.int64).reshape(-1, 2)
    n = len(pairs)
    # gms_matcher.py:96-101 NormalizePoints (python float arithmetic on the f32 pts)
    p1 = np.asarray(xy1, np.float64) / np.array([size1[0], size1[1]], np.float64)
    p2 = np.asarray(xy2, np.float64) / np.array([size2[0], size2[1]], np.float64)
    lp = p1[pairs[:, 0]]
    rp = p2[pairs[:, 1]]
    gwl = ghl = 20                                                  # :83
    nl = gwl * ghl
    nb_l = _neighbours(gwl, ghl)

    def run(rot, gwr, ghr, nb_r):                                   # :179-201
        nr = gwr * ghr
        mask = np.zeros(n, bool)
        rg = np.zeros(n, np.int64)
        for grid_type in (1, 2, 3, 4):
            # :219-236 GetGridIndexLeft / :238-241 GetGridIndexRight
            ox = 0.5 if grid_type in (2, 4) else 0.0
            oy = 0.5 if grid_type in (3, 4) else 0.0
            x = np.floor(lp[:, 0] * gwl + ox).astype(np.int64)
            y = np.floor(lp[:, 1] * ghl + oy).astype(np.int64)
            lg = np.where((x >= gwl) | (y >= ghl), -1, x + y * gwl)
            if grid_type == 1:
                rg = (np.floor(rp[:, 0] * gwr).astype(np.int64)
                      + np.floor(rp[:, 1] * ghr).astype(np.int64) * gwr)
            # :203-217 AssignMatchPairs
            stats = np.zeros((nl, nr), np.int64)
            ok = (lg >= 0) & (rg >= 0)
            np.add.at(stats, (lg[ok], rg[ok]), 1)
            cnt_l = stats.sum(1)
            # :243-285 VerifyCellPairs
            cell = -np.ones(nl, np.int64)
            for i in range(nl):
                if cnt_l[i] == 0:
                    continue
                j = int(np.argmax(stats[i]))        # first maximum, strict '>' scan
                cell[i] = j
                score, thresh, numpair = 0, 0.0, 0
                for k in range(9):
                    ll = nb_l[i, k]
                    rr = nb_r[j, _ROT[rot, k]]
                    if ll == -1 or rr == -1:
                        continue
                    score += stats[ll, rr]
                    thresh += cnt_l[ll]
                    numpair += 1
                thresh = threshold_factor * math.sqrt(thresh / numpair)
                if score < thresh:
                    cell[i] = -2
            # :196-199 mark inliers
            if port_negative_index:
                mask |= cell[lg] == rg              # numpy -1 -> last cell, like the port
            else:
                mask |= (lg >= 0) & (cell[np.maximum(lg, 0)] == rg)
        return mask

    best, best_n, last = None, 0, np.zeros(n, bool)
    scales = range(5) if with_scale else (0,)
    rots = range(8) if with_rotation else (0,)
    for s in scales:                                                # :127-177
        gwr = int(gwl * _SCALES[s])
        ghr = int(ghl * _SCALES[s])
        nb_r = _neighbours(gwr, ghr)
        for r in rots:
            last = run(r, gwr, ghr, nb_r)
            c = int(last.sum())
            if c > best_n:
                best, best_n = last, c
    return best if best is not None else last


# ---------------------------------------------------------------------------
# scripts/lib/matcher.py:157-182 filter_duplicates
# ---------------------------------------------------------------------------
def filter_duplicates(xy1, xy2, pairs):
    used1, used2, out = set(), set(), []
    for a, b in np.asarray(pairs).reshape(-1, 2).tolist():
        k1 = "%.2f-%.2f" % (float(xy1[a][0]), float(xy1[a][1]))
        k2 = "%.2f-%.2f" % (float(xy2[b][0]), float(xy2[b][1]))
        if k1 in used1 or k2 in used2:
            continue
        used1.add(k1)
        used2.add(k2)
        out.append([a, b])
    return np.array(out, np.int32).reshape(-1, 2)


# ---------------------------------------------------------------------------
# scripts/lib/matcher.py:187-200 filter_cross_check
# ---------------------------------------------------------------------------
def filter_cross_check(pairs1, pairs2):
    rev = set((int(a), int(b)) for a, b in np.asarray(pairs2).reshape(-1, 2).tolist())
    new1 = [[int(a), int(b)] for a, b in np.asarray(pairs1).reshape(-1, 2).tolist()
            if (int(b), int(a)) in rev]
    new2 = [[b, a] for a, b in new1]
    return (np.array(new1, np.int32).reshape(-1, 2), np.array(new2, np.int32).reshape(-1, 2))


# ---------------------------------------------------------------------------
# scripts/lib/matcher.py:218-300 basic_pair_matches
# ---------------------------------------------------------------------------
def basic_pair_matches(des1, xy1, des2, xy2, match_ratio, min_pairs, size, stages=None):
    empty = np.zeros((0, 2), np.int32)
    if des1 is None or des2 is None or len(des1) <= 1 or len(des2) <= 1:   # :205-210
        raise ZeroDivisionError("division by zero")     # :232 sum / len(matches) on []
    idx, d2 = knn2_l2(des1, des2)
    thresh = metric_filter(idx, d2, match_ratio)
    if stages is not None:
        stages['knn_idx'], stages['knn_d2'], stages['pregms'] = idx, d2, thresh
    if len(thresh) < min_pairs:                         # :271
        return empty
    mask = gms_inlier_mask(xy1, xy2, size, size, thresh)            # :285
    post = thresh[mask]
    if stages is not None:
        stages['postgms'] = post
    out = filter_duplicates(xy1, xy2, post)             # :294
    if len(out) < min_pairs:                            # :296
        return empty
    return out


# ---------------------------------------------------------------------------
# scripts/lib/matcher.py:304-347 bidirectional_pair_matches
# ---------------------------------------------------------------------------
def bidirectional_pair_matches(des1, xy1, des2, xy2, match_ratio, min_pairs, size):
    p1 = basic_pair_matches(des1, xy1, des2, xy2, match_ratio, min_pairs, size)
    if len(p1) >= min_pairs:
        p2 = basic_pair_matches(des2, xy2, des1, xy1, match_ratio, min_pairs, size)
    else:
        p2 = np.zeros((0, 2), np.int32)
    return filter_cross_check(p1, p2)
