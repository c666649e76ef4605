#!/usr/bin/env python3
"""End-to-end rate of the drop-in entry point matcher.find_matches() (not the kernels alone):
a synthetic nadir survey on a lawn-mower grid, keypoints = projections of shared ground points
(so overlapping pairs carry geometrically consistent matches that pass GMS and feed the surface
estimate), descriptors = a per-ground-point base vector + integer noise, clutter on top.

    python tools/find_matches_rate.py [rows cols [kpts]] [--profile] [--objects]
    (38 74 4096 = the 2812-image survey of BASELINE configs[2])

Prints pairs/s through find_matches (all-pairs schedule), and where the host time goes."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from imageanalysis_amd import image as iimg, matcher  # noqa: E402
from imageanalysis_amd._deps import getNode  # noqa: E402
from imageanalysis_amd.hostlib import camera  # noqa: E402

W, H, F = 5472, 3648, 3666.6665
AGL, SPACING = 100.0, 20.0


def base_descriptor(ids, rng_seed=99):
    """SIFT-like marginals (bench.py synth_descriptors) per ground point id"""
    out = np.empty((len(ids), 128), np.float32)
    for k, i in enumerate(ids):
        r = np.random.default_rng(rng_seed + int(i))
        x = r.gamma(0.6, 1.0, 128)
        x /= np.linalg.norm(x)
        x = np.minimum(x, 0.2)
        x /= np.linalg.norm(x)
        out[k] = np.clip(np.rint(x * 512.0), 0, 255)
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    rows, cols = (int(args[0]), int(args[1])) if len(args) >= 2 else (16, 25)
    kpts = int(args[2]) if len(args) >= 3 else 4096
    rng = np.random.default_rng(7)
    n_img = rows * cols
    tmp = tempfile.mkdtemp(prefix='iamx_fm_')
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
    camera.set_mount_params(0.0, -90.0, 0.0)

    # ground points: enough that an image sees ~0.8 * kpts of them
    half_w, half_h = 0.5 * W / F * AGL, 0.5 * H / F * AGL
    n0, n1 = -half_h - 5, (rows - 1) * SPACING + half_h + 5
    e0, e1 = -half_w - 5, (cols - 1) * SPACING + half_w + 5
    density = 0.8 * kpts / (4 * half_w * half_h)
    n_pts = int(density * (n1 - n0) * (e1 - e0))
    gnd = np.stack([rng.uniform(n0, n1, n_pts), rng.uniform(e0, e1, n_pts), rng.normal(0, 1.0, n_pts)], 1)
    t0 = time.time()
    base = base_descriptor(np.arange(n_pts))
    print('%d images (%d x %d), %d ground points, descriptors in %.1f s' % (n_img, rows, cols, n_pts, time.time() - t0))

    names = ['S%04d' % i for i in range(n_img)]

    class Proj(object):
        analysis_dir = an

        def findIndexByName(self, name):
            return names.index(name) if name in names else None

        def save_images_info(self):
            pass

    proj = Proj()
    proj.image_list = []
    for r in range(rows):
        for c in range(cols):
            k = c if r % 2 == 0 else cols - 1 - c
            ned = np.array([r * SPACING, k * SPACING, -AGL]) + rng.normal(0, 0.3, 3)
            yaw = (0.0 if r % 2 == 0 else 180.0) + rng.normal(0, 1.0)
            im = iimg.Image(an, names[len(proj.image_list)])
            im.set_pose_from_camera(ned.tolist(), yaw, -90.0 + rng.normal(0, 0.5), rng.normal(0, 0.5))
            # nadir projection with the yaw only (keypoints need to be consistent, not exact)
            cy, sy = np.cos(np.radians(yaw)), np.sin(np.radians(yaw))
            dn, de, dz = gnd[:, 0] - ned[0], gnd[:, 1] - ned[1], gnd[:, 2] - ned[2]
            fwd, right = cy * dn + sy * de, -sy * dn + cy * de
            u = W / 2.0 + F * right / dz
            v = H / 2.0 - F * fwd / dz
            vis = np.nonzero((u >= 1) & (u < W - 1) & (v >= 1) & (v < H - 1))[0]
            if len(vis) > kpts:
                vis = rng.choice(vis, kpts, replace=False)
            n_cl = kpts - len(vis)
            xy = np.concatenate([np.stack([u[vis], v[vis]], 1),
                                 np.stack([rng.uniform(1, W - 1, n_cl), rng.uniform(1, H - 1, n_cl)], 1)])
            des = np.concatenate([np.clip(base[vis] + rng.integers(-6, 7, (len(vis), 128)), 0, 255),
                                  base_descriptor(rng.integers(10 ** 7, 10 ** 8, n_cl))]).astype(np.float32)
            order = rng.permutation(kpts)
            xy, des = xy[order].astype(np.float32), des[order]
            if '--objects' in sys.argv:      # a python object per keypoint, like a cv2 detector's list
                im.kp_list = [iimg.make_keypoint(x, y, 3.0, 0.0, 1.0, 0) for x, y in xy.tolist()]
            else:                            # the array-backed list our detector delivers
                from imageanalysis_amd.keypoints import KeyPointList
                im.kp_list = KeyPointList(xy[:, 0], xy[:, 1], np.full(kpts, 3.0), np.zeros(kpts),
                                          np.ones(kpts), np.zeros(kpts, np.int32))
            im.des_list = des
            getNode('/smart', True).getChild(im.name, True).setFloat('tri_surface_m', 0.0)
            proj.image_list.append(im)
    print('project built in %.1f s' % (time.time() - t0))

    for a in sys.argv:
        if a.startswith('--ppb='):
            matcher.PAIRS_PER_BATCH = int(a[6:])
        if a.startswith('--batch-gb='):
            matcher.BATCH_BYTES = int(float(a[11:]) * (1 << 30))
    matcher.configure()
    K = camera.get_K()
    torch.cuda.synchronize()
    prof = cProfile.Profile() if '--profile' in sys.argv else None
    # phases of the call: set-up until the first launch, the rounds, the final save
    marks = {}
    from imageanalysis_amd import smart as _smart
    orig_launch, orig_save, orig_ssave = matcher._launch_batch, matcher.saveMatches, _smart.save

    trace = [] if '--trace' in sys.argv else None
    matcher._round_trace = trace

    def launch(*a, **k):
        marks.setdefault('first launch', time.perf_counter())
        marks['launches'] = marks.get('launches', 0) + 1
        t = time.perf_counter()
        r = orig_launch(*a, **k)
        if trace is not None:
            trace.append(('launch', time.perf_counter() - t, time.perf_counter() - marks['first launch']))
        return r

    def save(*a, **k):
        marks['rounds done'] = time.perf_counter()
        r = orig_save(*a, **k)
        marks['matches saved'] = time.perf_counter()
        return r

    def ssave(*a, **k):
        r = orig_ssave(*a, **k)
        marks['smart saved'] = time.perf_counter()
        return r
    matcher._launch_batch, matcher.saveMatches, _smart.save = launch, save, ssave
    c0 = os.times()
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    matcher.find_matches(proj, K, strategy='traditional', transform='homography', sort=True)
    if prof:
        prof.disable()
    dt = time.perf_counter() - t0
    c1 = os.times()
    seq = [('first launch', t0)] + [(k, marks[k]) for k in ('first launch', 'rounds done', 'matches saved', 'smart saved') if k in marks]
    print('phases: ' + ', '.join('%s +%.2f s' % (seq[i][0], seq[i][1] - seq[i - 1][1]) for i in range(1, len(seq)))
          + '; end +%.2f s; %d launches' % (t0 + dt - seq[-1][1], marks.get('launches', 0)))
    print('process CPU during find_matches: user %.1f s, system %.1f s in %.2f s wall'
          % (c1.user - c0.user, c1.system - c0.system, dt))
    try:
        from threadpoolctl import threadpool_info
        print('thread pools:', [(d.get('internal_api'), d.get('num_threads')) for d in threadpool_info()])
    except Exception as e:                                  # noqa: BLE001
        print('threadpoolctl:', e)
    n_pairs = n_img * (n_img - 1) // 2
    linked = sum(len(v) > 0 for im in proj.image_list for v in im.match_list.values()) // 2
    total = sum(len(v) for im in proj.image_list for v in im.match_list.values()) // 2
    print('find_matches: %d pairs in %.2f s = %.0f pairs/s; %d pairs with matches, %d matches'
          % (n_pairs, dt, n_pairs / dt, linked, total))
    if trace is not None:
        # per round: host time of the launch, wait for the device, host time of the finish, pairs with matches
        ln = [t for t in trace if t[0] == 'launch']
        fn = [t for t in trace if t[0] == 'finish']
        print('before the first launch: ' + ', '.join('%s +%.3f s' % (t[1], t[2] - t0) for t in trace if t[0] == 'pre'))
        print('round  at s   launch ms   wait ms  finish ms  hits')
        for r in list(range(0, min(40, len(fn)))) + list(range(40, len(fn), 40)):
            print('%5d %6.2f %9.1f %9.1f %9.1f %6d' % (r, ln[r][2], ln[r][1] * 1e3, fn[r][1] * 1e3, fn[r][2] * 1e3, fn[r][3]), *fn[r][4:])
        print('sums: launch %.2f s, wait %.2f s, finish %.2f s' % (sum(t[1] for t in ln), sum(t[1] for t in fn), sum(t[2] for t in fn)))
    if prof:
        pstats.Stats(prof).sort_stats('cumulative').print_stats(28)
        pstats.Stats(prof).sort_stats('tottime').print_stats(40)


if __name__ == '__main__':
    main()
