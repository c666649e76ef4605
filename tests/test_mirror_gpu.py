"""GPU: the drop-in python modules (imageanalysis_amd.matcher / .optimizer) end to end
against the reference's golden vectors and the oracle."""
import glob
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN
from test_host_logic import _image, _scene

pytestmark = pytest.mark.gpu

MATCH_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'match_*.npz')))
BA_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'ba_*.npz')))


# The synthetic strips below copy keypoints from image i-1 to image i with a pixel shift of
# (+250, -120): with the FC6310S camera 100 m above the ground that is a camera step of 3.27 m
# south and 6.82 m west, and the poses say so -- find_matches feeds the yaw error it estimates from
# the matches back into the poses (scripts/lib/matcher.py:990-993), so a strip whose poses
# contradict its pixels has its pairs triangulated to nonsense and discarded (:1001-1005).
def _configure(match_ratio=0.75, min_pairs=25, w=5472, h=3648):
    from imageanalysis_amd import matcher
    from imageanalysis_amd.hostlib import camera
    matcher.detector_node.setString('detector', 'SIFT')
    matcher.detector_node.setFloat('scale', 0.4)
    matcher.matcher_node.setFloat('match_ratio', match_ratio)
    matcher.matcher_node.setInt('min_pairs', min_pairs)
    matcher.matcher_node.__dict__.pop('schedule', None)
    camera.set_image_params(w, h)
    camera.set_K(3666.6665, 3666.6665, 2736.0, 1824.0)     # (the surface triangulation inverts K)
    matcher.configure()
    return matcher


@pytest.mark.parametrize('path', MATCH_CASES, ids=os.path.basename)
def test_pair_matches_equal_reference(path):
    g = np.load(path)
    matcher = _configure(float(g['match_ratio']), int(g['min_pairs']))
    i1, i2 = _image('A', g['des1'], g['xy1']), _image('B', g['des2'], g['xy2'])
    idx, dist = matcher.raw_matches(i1, i2)
    assert np.array_equal(idx, g['knn_fwd_idx']) and np.array_equal(dist, g['knn_fwd_dist'])
    assert np.array_equal(np.array(matcher.basic_pair_matches(i1, i2)).reshape(-1, 2), g['basic_fwd'])
    assert np.array_equal(np.array(matcher.basic_pair_matches(i2, i1)).reshape(-1, 2), g['basic_rev'])
    f, r = matcher.bidirectional_pair_matches(i1, i2)
    assert np.array_equal(np.array(f).reshape(-1, 2), g['bidir_fwd'])
    assert np.array_equal(np.array(r).reshape(-1, 2), g['bidir_rev'])


@pytest.mark.parametrize('path', MATCH_CASES, ids=os.path.basename)
def test_device_post_filter_equals_reference(path):
    """iamx_match_postfilter (sort/clip, GMS, de-dup, gates, cross check on the device) ==
    the reference's bidirectional_pair_matches output == the host filters."""
    g = np.load(path)
    ratio = float(g['match_ratio'])
    matcher = _configure(ratio, int(g['min_pairs']))
    i1, i2 = _image('A', g['des1'], g['xy1']), _image('B', g['des2'], g['xy2'])
    (f, r, _nf, _nr), = matcher._match_batch([(i1, i2)], ratio)
    assert np.array_equal(np.array(f).reshape(-1, 2), g['bidir_fwd'])
    assert np.array_equal(np.array(r).reshape(-1, 2), g['bidir_rev'])
    (f2, r2, _nf2, _nr2), = matcher._match_batch([(i1, i2)], ratio, device_filters=False)
    assert f == f2 and r == r2
    assert all(type(v) is int for pair in f for v in pair)


def test_device_post_filter_dense_duplicates_and_gates():
    """many pairs in one launch: dense motion-consistent matches (GMS keeps), random matches
    (GMS rejects), repeated keypoint positions (de-dup replay), pairs below min_pairs, pairs
    whose reverse direction fails; device lists == host filters on the same survivors."""
    from test_match_gpu import _sift_like
    matcher = _configure(0.75, 25)
    rng = np.random.default_rng(123)
    W, H = 5472, 3648
    base = _sift_like(rng, 1500)
    base_xy = np.stack([rng.uniform(300, W - 900, 1500), rng.uniform(300, H - 600, 1500)], 1)
    imgs = []
    for k in range(8):
        n = int(rng.integers(900, 1500))
        d = _sift_like(rng, n)
        xy = np.stack([rng.uniform(0, W - 1, n), rng.uniform(0, H - 1, n)], 1)
        m = int((0.05, 0.3, 0.6, 0.9, 0.02, 0.5, 0.7, 0.4)[k] * 900)
        src, dst = rng.permutation(1500)[:m], rng.permutation(n)[:m]
        d[dst] = np.clip(base[src].astype(int) + rng.integers(-6, 7, (m, 128)), 0, 255)
        xy[dst] = base_xy[src] + [40.0 * k, -25.0 * k] + rng.normal(0, 0.5, (m, 2))
        if k in (2, 5):                       # same pixel position for many keypoints
            xy[dst[:m // 3]] = np.round(xy[dst[:m // 3]] / 8) * 8
            d[dst[m // 2:m // 2 + 50]] = d[dst[:50]]          # and twin descriptors
        if k == 6:                            # geometry scrambled: matches exist, GMS rejects
            xy[dst] = np.stack([rng.uniform(0, W - 1, m), rng.uniform(0, H - 1, m)], 1)
        imgs.append(_image('D%d' % k, d, np.clip(xy, 0, [W - 1, H - 1]).astype(np.float32)))
    batch = [(imgs[a], imgs[b]) for a in range(8) for b in range(a + 1, 8)]
    dev = matcher._match_batch(batch, 0.75)
    host = matcher._match_batch(batch, 0.75, device_filters=False)
    n_nonempty = 0
    for (a, b), d_, h_ in zip(batch, dev, host):
        assert d_[0] == h_[0] and d_[1] == h_[1], (a.name, b.name)
        assert d_[2:] == h_[2:]
        n_nonempty += len(d_[0]) > 0
    assert 8 <= n_nonempty < len(batch)


def test_find_matches_batched_equals_pairwise_oracle():
    """find_matches over a 7-image strip: every match list == the oracle's bidirectional
    pipeline for that pair; already-matched pairs are skipped, empty ones retried."""
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    from oracle import match_oracle as mo
    from test_match_gpu import _sift_like
    matcher = _configure(0.75, 25)
    rng = np.random.default_rng(17)
    n_img, W, H = 7, 5472, 3648
    names = ['S%02d' % i for i in range(n_img)]
    proj = PoseProject(names)
    sizes = [900, 1100, 640, 1000, 777, 1300, 512]
    des, xy = [], []
    for i, n in enumerate(sizes):
        d = _sift_like(rng, n)
        p = np.stack([rng.uniform(600, W - 600, n), rng.uniform(400, H - 400, n)], 1)
        if i:
            k = min(int(0.45 * n), len(des[i - 1]))
            src = rng.permutation(len(des[i - 1]))[:k]
            dst = rng.permutation(n)[:k]
            d[dst] = np.clip(des[i - 1][src].astype(int) + rng.integers(-5, 6, (k, 128)), 0, 255)
            p[dst] = xy[i - 1][src] + [250.0, -120.0] + rng.normal(0, 0.6, (k, 2))
        des.append(d)
        xy.append(np.clip(p, 0, [W - 1, H - 1]).astype(np.float32))
    for i, im in enumerate(proj.image_list):
        im.set_camera_pose([-3.2727 * i, -6.8182 * i, -100.0], 0.0, -90.0, 0.0)
        fresh = _image(names[i], des[i], xy[i])
        im.des_list, im.kp_list = fresh.des_list, fresh.kp_list
    proj.image_list[0].match_list['S01'] = [[1, 2]]          # "already done" -> skipped
    proj.image_list[1].match_list['S00'] = [[2, 1]]
    proj.image_list[2].match_list['S03'] = []                # empty -> retried
    proj.image_list[3].match_list['S02'] = []
    matcher.find_matches(proj, None, strategy='traditional', transform='gms', sort=True)
    assert proj.image_list[0].match_list['S01'] == [[1, 2]]
    n_nonempty = 0
    for i in range(n_img):
        for j in range(i + 1, n_img):
            a, b = proj.image_list[i], proj.image_list[j]
            if (i, j) == (0, 1):
                continue
            if j - i > 4:
                assert names[j] not in a.match_list
                continue
            f, r = mo.bidirectional_pair_matches(des[i], xy[i], des[j], xy[j], 0.75, 25, (W, H))
            assert np.array_equal(np.array(a.match_list[names[j]]).reshape(-1, 2), f), (i, j)
            assert np.array_equal(np.array(b.match_list[names[i]]).reshape(-1, 2), r), (i, j)
            assert all(type(v) is int for pair in a.match_list[names[j]] for v in pair)
            n_nonempty += len(f) > 0
            assert not a.matches_clean or True
    assert n_nonempty >= 4


def test_find_matches_growing_arena_and_oversize_pairs():
    """several batches: the descriptor / keypoint arenas grow while old slots stay valid; and a
    pair with more survivors than the device sort holds takes the host filters -- same lists."""
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    from oracle import match_oracle as mo
    from test_match_gpu import _sift_like
    matcher = _configure(0.75, 25)
    rng = np.random.default_rng(31)
    n_img, W, H = 6, 5472, 3648
    names = ['G%02d' % i for i in range(n_img)]
    proj = PoseProject(names)
    des, xy = [], []
    for i in range(n_img):
        n = 700 + 50 * i
        d = _sift_like(rng, n)
        p = np.stack([rng.uniform(600, W - 600, n), rng.uniform(400, H - 400, n)], 1)
        if i:
            k = 350
            src, dst = rng.permutation(len(des[i - 1]))[:k], rng.permutation(n)[:k]
            d[dst] = np.clip(des[i - 1][src].astype(int) + rng.integers(-5, 6, (k, 128)), 0, 255)
            p[dst] = xy[i - 1][src] + [250.0, -120.0] + rng.normal(0, 0.6, (k, 2))
        des.append(d)
        xy.append(np.clip(p, 0, [W - 1, H - 1]).astype(np.float32))
    for i, im in enumerate(proj.image_list):
        im.set_camera_pose([-3.2727 * i, -6.8182 * i, -100.0], 0.0, -90.0, 0.0)
        fresh = _image(names[i], des[i], xy[i])
        im.des_list, im.kp_list = fresh.des_list, fresh.kp_list
    old = matcher.PAIRS_PER_BATCH
    matcher.PAIRS_PER_BATCH = 2                     # images join the arena batch by batch
    try:
        matcher.find_matches(proj, None, strategy='traditional', sort=True)
    finally:
        matcher.PAIRS_PER_BATCH = old
    n_nonempty = 0
    for i in range(n_img):
        for j in range(i + 1, min(i + 5, n_img)):
            f, r = mo.bidirectional_pair_matches(des[i], xy[i], des[j], xy[j], 0.75, 25, (W, H))
            a, b = proj.image_list[i], proj.image_list[j]
            assert np.array_equal(np.array(a.match_list[names[j]]).reshape(-1, 2), f), (i, j)
            assert np.array_equal(np.array(b.match_list[names[i]]).reshape(-1, 2), r), (i, j)
            n_nonempty += len(f) > 0
    assert n_nonempty >= 5
    # > 2048 survivors in a direction: radix select of the best 2000 instead of a full sort
    big = [_image('B%d' % k, _sift_like(rng, 5000),
                  np.stack([rng.uniform(0, W - 1, 5000), rng.uniform(0, H - 1, 5000)], 1).astype(np.float32))
           for k in range(2)]
    dev = matcher._match_batch([(big[0], big[1])], 1e6)         # threshold keeps every row
    host = matcher._match_batch([(big[0], big[1])], 1e6, device_filters=False)
    assert dev == host and dev[0][2] == 5000


def test_post_filter_clip_with_ties_at_the_boundary():
    """6000 survivors per direction whose metrics take only a few dozen distinct values: the
    2000-th smallest sits inside a long run of equal metrics, so the device selection must keep
    exactly the first ties in position order (what python's stable sort + [:2000] keeps)."""
    import torch
    from imageanalysis_amd import kernels
    from imageanalysis_amd.kernels import _ptr
    matcher = _configure(0.75, 25)
    rng = np.random.default_rng(9)
    n, W, H, ns = 9000, 5472.0, 3648.0, 6000
    xy1 = np.stack([rng.uniform(0, W - 400, n), rng.uniform(0, H - 1, n)], 1).astype(np.float32)
    xy2 = xy1.copy()
    xy2[:, 0] += 300.0
    L = kernels.lib()
    clip = int(L.iamx_match_postfilter_clip())
    dev = torch.device('cuda')
    t = lambda a, dt_: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt_)
    for trial in range(3):
        q = np.sort(rng.choice(n, ns, replace=False))
        tt = q.copy()
        bad = rng.random(ns) < 0.35
        tt[bad] = rng.integers(0, n, bad.sum())
        metric = rng.choice(np.linspace(5.0, 200.0, (3, 40, 400)[trial]), ns)
        order_r = np.argsort(tt, kind='stable')
        sf = (q.astype(np.int32), tt.astype(np.int32), metric)
        sr = (tt[order_r].astype(np.int32), q[order_r].astype(np.int32), metric[order_r])
        host = []
        for a, b, sv in ((xy1, xy2, sf), (xy2, xy1, sr)):
            pr = matcher._threshold_sort_clip(*sv)
            i1, i2 = _image('X', np.zeros((2, 128), np.uint8), a[:2]), _image('Y', np.zeros((2, 128), np.uint8), b[:2])
            host.append(matcher._post_filter(i1, i2, pr, (a, b)))
        want = matcher.filter_cross_check(host[0], host[1])[0]
        d = dict(off=t(np.array([0, ns, 2 * ns]), torch.int64), cnt=t(np.array([ns, ns]), torch.int32),
                 q=t(np.concatenate([sf[0], sr[0]]), torch.int32), t=t(np.concatenate([sf[1], sr[1]]), torch.int32),
                 m=t(np.concatenate([sf[2], sr[2]]), torch.float64),
                 pairs=t(np.array([[0, 1], [1, 0]]), torch.int32), kp_off=t(np.array([0, n]), torch.int64),
                 xy=t(np.concatenate([xy1, xy2]), torch.float32),
                 key2=t(matcher.kp_key2(np.concatenate([xy1, xy2])), torch.int32),
                 out_cnt=torch.empty(1, dtype=torch.int32, device=dev),
                 out_pairs=torch.empty((1, clip, 2), dtype=torch.int32, device=dev),
                 scratch=torch.empty((1, 2, clip, 2), dtype=torch.int32, device=dev),
                 stat=torch.empty((1, 4), dtype=torch.int32, device=dev),
                 status=torch.empty(1, dtype=torch.int32, device=dev))
        kernels.check(L.iamx_match_postfilter(_ptr(d['off']), _ptr(d['cnt']), _ptr(d['q']), _ptr(d['t']),
                                              _ptr(d['m']), _ptr(d['pairs']), _ptr(d['kp_off']),
                                              _ptr(d['xy']), _ptr(d['key2']), 1, W, H, 25.0, 5.0,
                                              _ptr(d['out_cnt']), _ptr(d['out_pairs']), _ptr(d['scratch']),
                                              _ptr(d['stat']), _ptr(d['status']), kernels.stream_ptr()),
                      'iamx_match_postfilter')
        torch.cuda.synchronize()
        assert int(d['status'][0].item()) == 0
        got = d['out_pairs'][0, :int(d['out_cnt'][0].item())].cpu().numpy()
        assert len(want) > 200 and np.array_equal(got, np.array(want).reshape(-1, 2)), trial


def test_find_matches_zero_division_like_reference():
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    matcher = _configure()
    proj = PoseProject(['Z0', 'Z1'])
    d = np.zeros((40, 128), np.uint8)
    for i, im in enumerate(proj.image_list):
        im.set_camera_pose([0.0, 10.0 * i, -100.0], 0.0, -90.0, 0.0)
        f = _image(im.name, d, np.zeros((40, 2), np.float32))
        im.des_list, im.kp_list = f.des_list, f.kp_list
    with pytest.raises(ZeroDivisionError):       # all distances 0 -> d0/d1 = 0/0 (matcher.py:255)
        matcher.find_matches(proj, None, strategy='traditional')
    # ... and the call after it is not affected: the device flag behind the exception lives in a
    # POOLED workspace that the next batch of the same size draws (it stayed raised until round 4)
    from test_match_gpu import _sift_like
    rng = np.random.default_rng(3)
    ok = PoseProject(['Y0', 'Y1'])
    for i, im in enumerate(ok.image_list):
        im.set_camera_pose([0.0, 10.0 * i, -100.0], 0.0, -90.0, 0.0)
        f = _image(im.name, _sift_like(rng, 40), rng.uniform(100, 3000, (40, 2)).astype(np.float32))
        im.des_list, im.kp_list = f.des_list, f.kp_list
    matcher.find_matches(ok, None, strategy='traditional')
    assert 'Y1' in ok.image_list[0].match_list


@pytest.mark.parametrize('solver', ['device', 'scipy'])
@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
def test_optimizer_fun_jac_run_against_reference(path, solver):
    import scipy.sparse as sp
    from imageanalysis_amd import optimizer
    g = np.load(path)
    proj, inp = _scene(path)
    opt = optimizer.Optimizer('/nonexistent')
    assert opt.solver == 'device'                       # the shipped default
    opt.solver = solver
    opt.setup(proj, inp['groups'], 0, inp['matches'], cam_calib=bool(g['cam_calib']))
    args = (opt.n_cameras, opt.n_points, opt.by_camera_point_indices, opt.by_camera_points_2d)
    scale = np.abs(g['f0']).max()
    f0 = opt.fun(g['x0'], *args)
    assert f0.shape == g['f0'].shape and np.abs(f0 - g['f0']).max() / scale < 1e-10
    J = opt.jac(g['x0'], *args)
    J3 = sp.csr_matrix((g['J3_data'], g['J3_indices'], g['J3_indptr']), shape=J.shape)
    assert np.array_equal(J.indptr, J3.indptr) and np.array_equal(J.indices, J3.indices)
    colmax = np.maximum(np.abs(J3).max(axis=0).toarray().ravel(), 1e-9)
    assert (np.abs((J - J3).toarray()) / colmax[None, :]).max() < 2e-5
    # full solve with the reference's settings (TRF, ftol=1e-4, x_scale='jac', bounds)
    ret = opt.run()
    assert len(ret) == 9 and ret[0].shape == (opt.n_cameras, 7) and ret[1].shape == (opt.n_points, 3)
    cost = 0.5 * float(opt.result.fun @ opt.result.fun)
    ref = float(g['cost_final'])
    assert cost < float(0.5 * g['f0'] @ g['f0']) * 1e-2
    assert abs(cost - ref) / ref < 2e-2, (cost, ref)          # same minimum; FD vs analytic J
    mre = np.mean(np.abs(opt.result.fun))
    assert abs(mre - np.mean(np.abs(g['f_final']))) < 0.02
    # 4b-mre-by-image.py:52-60 call form: fun() with explicit calib params appended
    if not bool(g['cam_calib']):
        opt.optimize_calib = 'global'
        x = np.hstack((opt.camera_params.ravel(), opt.points_3d.ravel(), opt.K[0, 0], opt.K[0, 2],
                       opt.K[1, 2], opt.distCoeffs))
        e = opt.fun(x, *args)
        assert np.abs(e - opt.result.fun).max() < 1e-9


@pytest.mark.parametrize('sort', [True, False], ids=['sorted', 'unsorted'])
def test_smart_json_written_ahead_of_time_equals_the_final_one(tmp_path, sort):
    """find_matches writes smart.json on a helper thread once several rounds in a row brought no
    match (the host idles through the quiet rounds of a distance-sorted survey) and only writes
    it again at the end when a pair with matches was booked later.  Either way the file is the
    one a call without the early write leaves (scripts/lib/matcher.py:1020-1031 writes it at the
    end): sorted schedule = every hit first, the early copy stands; unsorted (image order) =
    hits keep arriving, the end rewrites."""
    import json
    from imageanalysis_amd import smart
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    from test_match_gpu import _sift_like
    matcher = _configure(0.75, 25)
    n_img, W, H = 20, 5472, 3648
    rng = np.random.default_rng(23)
    des, xy = [], []
    for i in range(n_img):
        n = 600
        d = _sift_like(rng, n)
        p = np.stack([rng.uniform(600, W - 600, n), rng.uniform(400, H - 400, n)], 1)
        if i:
            # rows 0..299 of image i = rows 300..599 of image i - 1 (its own fresh rows: only
            # neighbours share features, no chains across three images)
            k = 300
            src, dst = np.arange(k, 2 * k), np.arange(k)
            d[dst] = np.clip(des[i - 1][src].astype(int) + rng.integers(-5, 6, (k, 128)), 0, 255)
            p[dst] = xy[i - 1][src] + [250.0, -120.0] + rng.normal(0, 0.6, (k, 2))
        des.append(d)
        xy.append(np.clip(p, 0, [W - 1, H - 1]).astype(np.float32))
    files = {}
    old = matcher.PAIRS_PER_BATCH, matcher.EARLY_SMART_ROUNDS
    try:
        for mode, after in (('early', 1), ('plain', 10 ** 9)):
            tag = ('E' if mode == 'early' else 'P') + ('s' if sort else 'u')
            names = ['%s%02d' % (tag, i) for i in range(n_img)]
            out = tmp_path / mode
            out.mkdir()
            proj = PoseProject(names, analysis_dir=str(out))
            for i, im in enumerate(proj.image_list):
                im.set_camera_pose([-3.2727 * i, -6.8182 * i, -100.0], 0.0, -90.0, 0.0)
                fresh = _image(names[i], des[i], xy[i])
                im.des_list, im.kp_list = fresh.des_list, fresh.kp_list
            matcher.PAIRS_PER_BATCH, matcher.EARLY_SMART_ROUNDS = 2, after
            before = dict(matcher.early_smart_stats)
            matcher.find_matches(proj, None, strategy='traditional', sort=sort)
            written = matcher.early_smart_stats['written'] - before['written']
            stands = matcher.early_smart_stats['current_at_end'] - before['current_at_end']
            if mode == 'early':
                assert written == 1 and stands == (1 if sort else 0), matcher.early_smart_stats
            else:
                assert written == 0 and stands == 0
            tree = json.load(open(out / 'smart.json'))
            mine = {k.replace(tag, 'X'): json.loads(json.dumps(v).replace(tag, 'X'))
                    for k, v in tree.items() if k.startswith(tag)}
            assert len(mine) >= n_img - 1
            files[mode] = mine
            hits = sum(len(v) > 0 for im in proj.image_list for v in im.match_list.values()) // 2
            assert hits >= n_img - 2
    finally:
        matcher.PAIRS_PER_BATCH, matcher.EARLY_SMART_ROUNDS = old
    assert files['early'] == files['plain']
    assert any('yaw_pairs' in v or 'tri_surface_pairs' in v for v in files['early'].values())


@pytest.mark.parametrize('route', ['never', 'always', 'auto'])
def test_find_matches_dense_routing_gives_the_same_lists(route):
    """A round of find_matches may take the symmetric sweep or the one-direction bound form
    (matcher.DENSE_ROUTE): every match list == the oracle's bidirectional pipeline either way, with
    the arena growing in between (the parity-partitioned copy is rebuilt from the rows already on
    the device) and, in 'auto', rounds of both kinds in one call."""
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    from oracle import match_oracle as mo
    from test_match_gpu import _sift_like
    matcher = _configure(0.75, 25)
    rng = np.random.default_rng(53)
    n_img, W, H = 8, 5472, 3648
    names = ['R%02d' % i for i in range(n_img)]
    proj = PoseProject(names)
    des, xy = [], []
    for i in range(n_img):
        n = 2100 + 130 * i
        d = _sift_like(rng, n)
        p = np.stack([rng.uniform(600, W - 600, n), rng.uniform(400, H - 400, n)], 1)
        if i:
            # the first images overlap heavily (and their unmatched rows sit close to the metric
            # threshold: many candidate rows), the last ones barely
            k = int((0.6 if i < 5 else 0.03) * min(n, len(des[i - 1])))
            src, dst = rng.permutation(len(des[i - 1]))[:k], rng.permutation(n)[:k]
            d[dst] = np.clip(des[i - 1][src].astype(int) + rng.integers(-5, 6, (k, 128)), 0, 255)
            p[dst] = xy[i - 1][src] + [250.0, -120.0] + rng.normal(0, 0.6, (k, 2))
        des.append(d)
        xy.append(np.clip(p, 0, [W - 1, H - 1]).astype(np.float32))
    for i, im in enumerate(proj.image_list):
        im.set_camera_pose([-3.2727 * i, -6.8182 * i, -100.0], 0.0, -90.0, 0.0)
        fresh = _image(names[i], des[i], xy[i])
        im.des_list, im.kp_list = fresh.des_list, fresh.kp_list
    old = (matcher.PAIRS_PER_BATCH, matcher.DENSE_ROUTE, matcher.DENSE_SHARE, matcher.DENSE_PROBE)
    matcher.PAIRS_PER_BATCH, matcher.DENSE_ROUTE = 3, route
    if route == 'auto':
        matcher.DENSE_SHARE, matcher.DENSE_PROBE = 0.0005, 3      # (any candidates at all: dense)
    try:
        matcher.find_matches(proj, None, strategy='traditional', sort=True)
        rounds = list(matcher._route['rounds'])
    finally:
        matcher.PAIRS_PER_BATCH, matcher.DENSE_ROUTE, matcher.DENSE_SHARE, matcher.DENSE_PROBE = old
    assert sum(rounds) >= 6
    if route == 'never':
        assert rounds[1] == 0
    elif route == 'always':
        assert rounds[0] == 0
    else:
        assert rounds[0] >= 2 and rounds[1] >= 2, rounds
    n_nonempty = 0
    for i in range(n_img):
        for j in range(i + 1, min(i + 5, n_img)):
            f, r = mo.bidirectional_pair_matches(des[i], xy[i], des[j], xy[j], 0.75, 25, (W, H))
            a, b = proj.image_list[i], proj.image_list[j]
            assert np.array_equal(np.array(a.match_list[names[j]]).reshape(-1, 2), f), (route, i, j)
            assert np.array_equal(np.array(b.match_list[names[i]]).reshape(-1, 2), r), (route, i, j)
            n_nonempty += len(f) > 0
    assert n_nonempty >= 4
