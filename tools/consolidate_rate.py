#!/usr/bin/env python3
"""The consolidation stage of scripts/process.py:305-331 (merge_duplicates, check_for_pair_dups,
check_for_1vn_dups, make_match_structure, link_matches) on a synthetic survey of the shape of the
512-frame configs[4] run -- a grid of frames, every world point seen by the frames around it,
every overlapping pair's matches as find_matches leaves them (array-backed MatchPairs) -- HOST
code only, no GPU needed: per-function seconds (and a cProfile with --profile).
    python tools/consolidate_rate.py [rows cols [points_per_frame]] [--profile] [--dup=0.15]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis_amd import match_cleanup  # noqa: E402
from imageanalysis_amd.hostlib.image_pose import PoseProject  # noqa: E402
from imageanalysis_amd.keypoints import KeyPointList  # noqa: E402
from imageanalysis_amd.matchpairs import MatchPairs  # noqa: E402


def build(rows, cols, per_frame, seed=0, dup=0.0):
    """frames on a grid with 70 % overlap along a row and 60 % between rows: a world point falls
    into ~8 frames; a pair of frames keeps ~55 % of its common points as matches"""
    rng = np.random.default_rng(seed)
    W, H = 5472.0, 3648.0
    sx, sy = 0.30 * W, 0.40 * H                       # frame spacing in ground pixels
    n_img = rows * cols
    names = ['S%04d' % k for k in range(n_img)]
    proj = PoseProject(names)
    origin = np.array([[c * sx, r * sy] for r in range(rows) for c in range(cols)])
    # world points: uniform over the covered area, at the density that gives per_frame per frame
    area = ((cols - 1) * sx + W) * ((rows - 1) * sy + H)
    n_pts = int(per_frame * area / (W * H))
    pts = np.stack([rng.uniform(0, (cols - 1) * sx + W, n_pts), rng.uniform(0, (rows - 1) * sy + H, n_pts)], 1)
    kp_of = []                                         # per image: world point ids in keypoint order
    for k in range(n_img):
        inside = np.nonzero((pts[:, 0] >= origin[k, 0]) & (pts[:, 0] < origin[k, 0] + W) &
                            (pts[:, 1] >= origin[k, 1]) & (pts[:, 1] < origin[k, 1] + H))[0]
        inside = rng.permutation(inside)
        kp_of.append(inside)
        im = proj.image_list[k]
        im.set_camera_pose([origin[k, 1] * 0.05, origin[k, 0] * 0.05, -100.0], 0.0, -90.0, 0.0)
        xy = (pts[inside] - origin[k]).astype(np.float32)
        if dup > 0 and len(xy) > 1:
            # SIFT gives a location with several dominant orientations several keypoints: the same
            # pixel more than once (what merge_duplicates is for)
            twins = rng.random(len(xy)) < dup
            xy[twins] = xy[rng.integers(0, len(xy), int(twins.sum()))]
        z = np.zeros(len(xy), np.float32)
        im.kp_list = KeyPointList(xy[:, 0].copy(), xy[:, 1].copy(), z + 3.0, z, z + 0.05, z.astype(np.int32))
        im.uv_list = xy
        im.match_list = {}
    where = np.full(n_pts, -1, np.int64)
    n_pairs = n_matches = 0
    for a in range(n_img):
        where[kp_of[a]] = np.arange(len(kp_of[a]))
        ra, ca = divmod(a, cols)
        for b in range(a + 1, n_img):
            rb, cb = divmod(b, cols)
            if abs(rb - ra) > 2 or abs(cb - ca) > 3:
                continue
            idx_b = np.nonzero(where[kp_of[b]] >= 0)[0]
            if len(idx_b) < 25:
                continue
            keep = idx_b[rng.random(len(idx_b)) < 0.55]
            pairs = np.stack([where[kp_of[b][keep]], keep], 1).astype(np.int32)
            proj.image_list[a].match_list[names[b]] = MatchPairs(pairs)
            proj.image_list[b].match_list[names[a]] = MatchPairs(np.ascontiguousarray(pairs[:, ::-1]))
            n_pairs += 1
            n_matches += len(pairs)
        where[kp_of[a]] = -1
    return proj, n_pairs, n_matches


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    rows, cols = (int(args[0]), int(args[1])) if len(args) >= 2 else (16, 32)
    per_frame = int(args[2]) if len(args) >= 3 else 37000
    t0 = time.time()
    dup = max([float(a[6:]) for a in sys.argv if a.startswith('--dup=')] or [0.15])
    proj, n_pairs, n_matches = build(rows, cols, per_frame, dup=dup)
    print('%d frames, %d pairs with matches, %d matches (built in %.1f s)'
          % (rows * cols, n_pairs, n_matches, time.time() - t0))
    from imageanalysis_amd import _lib
    _lib.lib()                          # (loading the library is not part of the stage)
    prof = cProfile.Profile() if '--profile' in sys.argv else None
    if prof:
        prof.enable()
    total = 0.0
    direct = None
    for name, fn in (('merge_duplicates', lambda: match_cleanup.merge_duplicates(proj)),
                     ('check_for_pair_dups', lambda: match_cleanup.check_for_pair_dups(proj)),
                     ('check_for_1vn_dups', lambda: match_cleanup.check_for_1vn_dups(proj)),
                     ('make_match_structure', lambda: match_cleanup.make_match_structure(proj)),
                     ('link_matches', lambda: match_cleanup.link_matches(proj, direct))):
        t = time.perf_counter()
        r = fn()
        dt = time.perf_counter() - t
        total += dt
        if name == 'make_match_structure':
            direct = r
        print('  %-22s %6.3f s' % (name, dt))
    if prof:
        prof.disable()
    print('  %-22s %6.3f s;  %d chains' % ('consolidate', total, len(r)))
    if prof:
        pstats.Stats(prof).sort_stats('tottime').print_stats(25)


if __name__ == '__main__':
    main()
