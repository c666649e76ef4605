#!/usr/bin/env python3
"""End-to-end time of the drop-in BA entry points at BASELINE config 4 scale: a synthetic
matches_grouped list (2812 cameras, ~270 k features, ~1.96 M observations) through
Optimizer.setup() and Optimizer.run() -- python list handling included, not the solver alone.

    python tools/optimizer_rate.py [--profile]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis_amd import optimizer, synth  # noqa: E402
from imageanalysis_amd.hostlib import camera  # noqa: E402
from imageanalysis_amd.hostlib import transforms as tf  # noqa: E402
from imageanalysis_amd.hostlib.image_pose import PoseProject  # noqa: E402


def main():
    t0 = time.time()
    prob = synth.make_ba_problem()
    C, P = len(prob['cams0']), len(prob['pts0'])
    names = ['I%04d' % i for i in range(C)]
    proj = PoseProject(names)
    K = np.asarray(prob['K'], float)
    camera.set_K(K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    camera.set_dist_coeffs([0.0] * 5)
    camera.set_image_params(5472, 3648)
    for im, c in zip(proj.image_list, prob['cams0']):
        e = tf.euler_from_quaternion(c[3:7], 'rzyx')
        im.set_camera_pose(c[:3].tolist(), *[float(np.degrees(a)) for a in e])
    order = np.argsort(prob['pt_idx'], kind='stable')
    pi, ci, uv = prob['pt_idx'][order], prob['cam_idx'][order], prob['uv'][order]
    ptr = np.searchsorted(pi, np.arange(P + 1))
    cl, uvl, pts = ci.tolist(), uv.tolist(), prob['pts0'].tolist()
    matches = [[pts[p], 0] + [[cl[k], uvl[k]] for k in range(ptr[p], ptr[p + 1])] for p in range(P)]
    print('synthetic project: %d cameras, %d features, %d observations (%.1f s to build)'
          % (C, len(matches), len(cl), time.time() - t0))
    prof = cProfile.Profile() if '--profile' in sys.argv else None
    opt = optimizer.Optimizer('/tmp')
    if prof:
        prof.enable()
    t0 = time.perf_counter()
    opt.setup(proj, [names], 0, matches, optimized=False, cam_calib=False)
    t1 = time.perf_counter()
    opt.run()
    t2 = time.perf_counter()
    if prof:
        prof.disable()
    res = opt.result
    print('setup %.2f s, run %.2f s: njev %d nfev %d cost %.4g -> %.2f TRF iterations/s through '
          'Optimizer.run()' % (t1 - t0, t2 - t1, res.njev, res.nfev, res.cost, res.njev / (t2 - t1)))
    t3 = time.perf_counter()
    opt.update_camera_poses(proj)
    print('update_camera_poses %.2f s' % (time.perf_counter() - t3))
    if prof:
        pstats.Stats(prof).sort_stats('cumulative').print_stats(30)


if __name__ == '__main__':
    main()
