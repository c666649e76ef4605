#!/usr/bin/env python3
"""The HOST side of matcher.find_matches() on an all-pairs survey, without a GPU: the device
batch is replaced by a stand-in that hands back synthetic round results of the shape the kernels
deliver (pairs closer than ~110 m carry a few hundred matches and a surface / yaw record, the
others nothing), so that what python spends per round -- scheduling, booking, smart records,
.match pickles, the final save -- can be profiled on any machine.

    python tools/fm_host_profile.py [rows cols] [--profile]      (38 74 = the configs[2] survey)"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis_amd import image as iimg, matcher  # noqa: E402
from imageanalysis_amd._deps import getNode  # noqa: E402
from imageanalysis_amd.hostlib import camera  # noqa: E402
from imageanalysis_amd.keypoints import KeyPointList  # noqa: E402
from imageanalysis_amd.matchpairs import MatchPairs  # noqa: E402

W, H, F = 5472, 3648, 3666.6665
AGL, SPACING = 100.0, 20.0


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    rows, cols = (int(args[0]), int(args[1])) if len(args) >= 2 else (20, 40)
    rng = np.random.default_rng(7)
    tmp = tempfile.mkdtemp(prefix='iamx_fmh_')
    an = os.path.join(tmp, 'ImageAnalysis')
    os.makedirs(os.path.join(an, 'meta'))
    os.makedirs(os.path.join(an, 'cache'))
    getNode('/config/directories', True).setString('project_dir', tmp)
    matcher.detector_node.setString('detector', 'SIFT')
    matcher.detector_node.setFloat('scale', 0.4)
    matcher.matcher_node.setFloat('match_ratio', 0.75)
    matcher.matcher_node.setInt('min_pairs', 25)
    matcher.matcher_node.setString('schedule', 'all-pairs')
    camera.set_K(F, F, W / 2.0, H / 2.0)
    camera.set_dist_coeffs([0.0] * 5)
    camera.set_image_params(W, H)
    n_img = rows * cols
    names = ['S%04d' % i for i in range(n_img)]

    class Proj(object):
        analysis_dir = an

        def findIndexByName(self, name):
            return names.index(name) if name in names else None

        def save_images_info(self):
            pass

    proj = Proj()
    proj.image_list = []
    kp = KeyPointList(np.arange(8.0), np.arange(8.0), np.full(8, 3.0), np.zeros(8), np.ones(8), np.zeros(8, np.int32))
    for r in range(rows):
        for c in range(cols):
            k = c if r % 2 == 0 else cols - 1 - c
            ned = np.array([r * SPACING, k * SPACING, -AGL]) + rng.normal(0, 0.3, 3)
            im = iimg.Image(an, names[len(proj.image_list)])
            im.set_camera_pose(ned.tolist(), (0.0 if r % 2 == 0 else 180.0) + rng.normal(0, 1.0), -90.0, 0.0)
            im.kp_list = kp
            im.des_list = np.zeros((4096, 128), np.float32)[:8]
            getNode('/smart', True).getChild(im.name, True).setFloat('tri_surface_m', 0.0)
            proj.image_list.append(im)
    ned = np.array([im.get_camera_pose()[0] for im in proj.image_list])

    matcher.max_distance, matcher.min_pairs = 270.0, 25.0
    matcher.the_matcher = object()                   # no device matcher: the stand-in below
    t_launch = [0.0]

    def launch(view, ratio, **kw):
        """a round's results as _finish_batch_arrays() delivers them"""
        t = time.perf_counter()
        pi, pj = view.pi, view.pj
        n = len(pi)
        d = np.linalg.norm(ned[pj] - ned[pi], axis=1)
        hit = np.nonzero(d < 112.0)[0]
        R = matcher._RoundResult()
        R.n = n
        R.n_fwd = np.full(n, 3, np.int64)
        R.n_rev = np.full(n, 2, np.int64)
        R.cc = np.zeros(n, np.int64)
        R.quiet = np.ones(n, bool)
        R.quiet[hit] = False
        cnt = np.maximum(30, (1700 * (1.0 - d[hit] / 118.0) ** 1.5).astype(np.int64))
        R.cc[hit] = cnt
        R.n_fwd[hit] = cnt + 40
        R.n_rev[hit] = cnt + 35
        tot = int(cnt.sum())
        fwd_all = rng.integers(0, 4096, (tot, 2), dtype=np.int32)
        rev_all = np.ascontiguousarray(fwd_all[:, ::-1])
        off = np.concatenate([[0], np.cumsum(cnt)]).tolist()
        R.hits = []
        for t_, k in enumerate(hit.tolist()):
            surf = (-1.0 + 0.01 * t_, 0.9, float(d[k]), None, None, (1.5, float(d[k]), 30.0, 0.7), (-1.2, float(d[k]), 210.0, 0.6))
            R.hits.append((k, MatchPairs.of_array(fwd_all[off[t_]:off[t_ + 1]]),
                           MatchPairs.of_array(rev_all[off[t_]:off[t_ + 1]]), surf))
        t_launch[0] += time.perf_counter() - t
        return R

    matcher._launch_batch = launch
    matcher._finish_batch_arrays = lambda h: h
    prof = cProfile.Profile() if '--profile' in sys.argv else None
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    matcher.find_matches(proj, camera.get_K(), strategy='traditional', transform='homography', sort=True)
    if prof:
        prof.disable()
    dt = time.perf_counter() - t0
    n_pairs = n_img * (n_img - 1) // 2
    linked = sum(len(v) > 0 for im in proj.image_list for v in im.match_list.values()) // 2
    total = sum(len(v) for im in proj.image_list for v in im.match_list.values()) // 2
    print('host side of find_matches: %d pairs in %.2f s (%.2f s of it in the stand-in that makes the results); '
          '%d pairs with matches, %d matches -> %.1f us of host time per pair with matches'
          % (n_pairs, dt, t_launch[0], linked, total, (dt - t_launch[0]) / max(linked, 1) * 1e6))
    if prof:
        pstats.Stats(prof).sort_stats('tottime').print_stats(32)


if __name__ == '__main__':
    main()
