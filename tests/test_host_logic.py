"""CPU: host-side logic of the matcher / optimizer mirrors against the reference's golden
vectors -- nothing here needs (or may silently replace) a device kernel."""
import glob
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN

MATCH_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'match_*.npz')))
BA_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'ba_*.npz')))


class KP(object):
    def __init__(self, x, y):
        self.pt = (float(x), float(y))


def _image(name, des, xy):
    from imageanalysis_amd.hostlib.image_pose import PoseImage
    im = PoseImage(name)
    im.des_list = des.astype(np.float32)
    im.kp_list = [KP(x, y) for x, y in xy]
    return im


@pytest.mark.parametrize('path', MATCH_CASES, ids=os.path.basename)
def test_gms_dedupe_crosscheck_match_reference(path):
    from imageanalysis_amd import matcher
    from imageanalysis_amd.gms import gms_inlier_mask
    g = np.load(path)
    size = (int(g['width']), int(g['height']))
    i1, i2 = _image('A', g['des1'], g['xy1']), _image('B', g['des2'], g['xy2'])
    fwd = rev = None
    for tag, (a, b) in dict(fwd=(i1, i2), rev=(i2, i1)).items():
        pre = g['pregms_%s' % tag]
        if len(pre) == 0:
            continue
        xa, xb = matcher._kp_xy(a), matcher._kp_xy(b)
        mask = gms_inlier_mask(xa, xb, size, size, pre)
        assert np.array_equal(pre[mask], g['postgms_%s' % tag])
        out = matcher.filter_duplicates(a, b, [list(map(int, p)) for p in pre[mask]])
        assert np.array_equal(np.array(out).reshape(-1, 2), g['basic_%s' % tag])
        if tag == 'fwd':
            fwd = out
        else:
            rev = out
    if fwd is not None and rev is not None:
        f, r = matcher.filter_cross_check(fwd, rev)
        assert np.array_equal(np.array(f).reshape(-1, 2), g['bidir_fwd'])
        assert np.array_equal(np.array(r).reshape(-1, 2), g['bidir_rev'])


def test_product_gms_equals_oracle_on_border_points():
    from imageanalysis_amd.gms import gms_inlier_mask
    from oracle import match_oracle as mo
    rng = np.random.default_rng(0)
    for it in range(8):
        n1, n2 = 400, 420
        xy1 = np.stack([rng.uniform(0, 5472, n1), rng.uniform(0, 3648, n1)], 1).astype(np.float32)
        xy2 = np.stack([rng.uniform(0, 5472, n2), rng.uniform(0, 3648, n2)], 1).astype(np.float32)
        k = int(rng.integers(50, 1500))
        q, t = rng.integers(0, n1, k), rng.integers(0, n2, k)
        xy2[t[:k // 2]] = np.clip(xy1[q[:k // 2]] + [30, -20] + rng.normal(0, 3, (k // 2, 2)),
                                  0, [5471, 3647]).astype(np.float32)
        pairs = np.stack([q, t], 1)
        a = gms_inlier_mask(xy1, xy2, (5472, 3648), (5472, 3648), pairs)
        b = mo.gms_inlier_mask(xy1, xy2, (5472, 3648), (5472, 3648), pairs)
        assert np.array_equal(a, b)


def test_work_list_schedules():
    from imageanalysis_amd import matcher
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    names = ['W%02d' % i for i in range(12)]
    proj = PoseProject(names)
    for i, im in enumerate(proj.image_list):
        im.set_camera_pose([0.0, 20.0 * i, -100.0], 0.0, -90.0, 0.0)
    node = matcher.matcher_node
    for key in ('schedule', 'min_dist', 'max_dist'):
        node.__dict__.pop(key, None)
    w = matcher._work_list(proj, sort=True)
    # HEAD: only |i-j| <= 4 (lib/matcher.py:899), discretised distance, stable sort
    assert sorted((i, j) for d, i, j in w) == [(i, j) for i in range(12) for j in range(i + 1, 12)
                                                if j - i <= 4]
    assert [d for d, i, j in w] == sorted(d for d, i, j in w)
    assert w[0][0] == 26.0 and w[-1][0] == 78.0          # interval = 20 * 1.3
    node.setString('schedule', 'all-pairs')
    assert len(matcher._work_list(proj, sort=False)) == 66
    node.__dict__.pop('schedule')


def _scene(path):
    from imageanalysis_amd._deps import getNode
    from imageanalysis_amd.hostlib import camera
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    with open(path.replace('.npz', '_in.pkl'), 'rb') as f:
        inp = pickle.load(f)
    proj = PoseProject(inp['names'])
    for im, (ned, ypr, quat) in zip(proj.image_list, inp['poses']):
        im.set_camera_pose(ned, ypr[0], ypr[1], ypr[2])
    node = getNode('/config/camera', True)
    node.setLen('K', 9)
    for i, v in enumerate(inp['K']):
        node.setFloatEnum('K', i, v)
    node.__dict__.pop('K_opt', None)
    node.__dict__.pop('dist_coeffs_opt', None)
    camera.set_dist_coeffs(inp['dist'])
    camera.set_image_params(inp['width'], inp['height'])
    return proj, inp


@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
def test_optimizer_setup_matches_reference(path):
    from imageanalysis_amd import optimizer
    g = np.load(path)
    proj, inp = _scene(path)
    opt = optimizer.Optimizer('/nonexistent')
    opt.setup(proj, inp['groups'], 0, inp['matches'], optimized=False,
              cam_calib=bool(g['cam_calib']))
    C, P = int(g['n_cameras']), int(g['n_points'])
    assert (opt.n_cameras, opt.n_points) == (C, P)
    assert np.array_equal(opt.camera_indices, g['camera_indices'])
    assert np.array_equal(opt.point_indices, g['point_indices'])
    assert np.array_equal([opt.camera_map_fwd[i] for i in range(C)], g['camera_map_fwd'])
    assert np.array_equal([opt.feat_map_rev[i] for i in range(P)], g['feat_map_rev'])
    assert np.array_equal([len(a) for a in opt.by_camera_point_indices], g['by_camera_counts'])
    uv = np.concatenate([a.reshape(-1, 2) for a in opt.by_camera_points_2d if len(a)])
    assert np.array_equal(uv, g['points_2d'])
    assert all(a.shape == (len(a), 1, 2) for a in opt.by_camera_points_2d)
    x0 = opt._x0()
    assert np.allclose(x0, g['x0'], rtol=0, atol=1e-12)       # quats come from ypr -> 1e-16
    # sparsity mask == the reference's (nnz and pattern of the FD Jacobian)
    A = opt.bundle_adjustment_sparsity(C, P, opt.camera_indices, opt.point_indices)
    assert A.nnz == int(g['J_sparsity_nnz'])
    csr = A.tocsr()
    csr.sort_indices()
    assert np.array_equal(csr.indptr, g['J2_indptr'])
    assert np.array_equal(csr.indices, g['J2_indices'])
    # rvec/tvec helper
    for c in range(C):
        rvec, tvec = opt.nedquat2rvectvec(x0[c * 7:c * 7 + 3], x0[c * 7 + 3:c * 7 + 7])
        assert np.allclose(np.asarray(rvec).ravel(), g['rvecs'][c], atol=1e-9)
        assert np.allclose(np.asarray(tvec).ravel(), g['tvecs'][c], atol=1e-9)
    lo, up = opt._bounds()
    assert len(lo) == x0.size and lo[0] == x0[0] - 3 and up[2] == x0[2] + 9 and lo[3] == -np.inf


@pytest.mark.parametrize('path', [p for p in BA_CASES], ids=os.path.basename)
def test_pose_writeback_and_refit_match_reference(path):
    """update_camera_poses() + refit() from the reference's converged x* (G6)."""
    from imageanalysis_amd import optimizer
    g = np.load(path)
    with open(path.replace('.npz', '_refit.pkl'), 'rb') as f:
        want = pickle.load(f)
    proj, inp = _scene(path)
    opt = optimizer.Optimizer('/nonexistent')
    matches = inp['matches']
    opt.setup(proj, inp['groups'], 0, matches, cam_calib=bool(g['cam_calib']))
    C, P = opt.n_cameras, opt.n_points
    xf = g['x_final']
    opt.camera_params = xf[:C * 7].reshape(C, 7)
    opt.points_3d = xf[C * 7:C * 7 + P * 3].reshape(P, 3)
    opt.update_camera_poses(proj)
    for im, (ned, ypr, quat), valid in zip(proj.image_list, want['poses_opt'], want['valid']):
        assert bool(im.node.getChild('camera_pose_opt', True).getBool('valid')) == valid
        if valid:
            n2, y2, q2 = im.get_camera_pose(opt=True)
            assert np.allclose(n2, ned, atol=1e-9) and np.allclose(y2, ypr, atol=1e-9)
            assert np.allclose(q2, quat, atol=1e-12)
    opt.refit(proj, matches, inp['groups'], 0)
    for im, (ned, ypr, quat), valid in zip(proj.image_list, want['poses_refit'], want['valid']):
        if valid:
            n2, y2, q2 = im.get_camera_pose(opt=True)
            assert np.allclose(n2, ned, atol=1e-8) and np.allclose(y2, ypr, atol=1e-8)
            assert np.allclose(q2, quat, atol=1e-10)
    for m, wpt in zip(matches, want['matches_points']):
        assert np.allclose(m[0], wpt, atol=1e-8)


def test_transforms_known_answers():
    from imageanalysis_amd.hostlib import transforms as tf
    # doctest values of the reference's archived transformations.py (:1398-1406)
    assert np.allclose(tf.quaternion_matrix([1, 0, 0, 0]), np.identity(4))
    assert np.allclose(tf.quaternion_matrix([0, 1, 0, 0]), np.diag([1, -1, -1, 1]))
    q = tf.quaternion_from_euler(0.3, -1.2, 0.5, 'rzyx')
    assert np.allclose(tf.euler_from_quaternion(q, 'rzyx'), (0.3, -1.2, 0.5))


def test_feature_cache_formats_roundtrip(tmp_path):
    """.feat/.desc/.match files in the reference's formats (lib/image.py:140-228), no GPU."""
    import gzip
    from imageanalysis_amd import image as iimg
    from imageanalysis_amd._deps import getNode
    an = tmp_path / 'ImageAnalysis'
    (an / 'cache').mkdir(parents=True)
    (an / 'meta').mkdir()
    getNode('/config/directories', True).setString('project_dir', str(tmp_path))
    im = iimg.Image(str(an), 'IMG_7')
    im.kp_list = [iimg.make_keypoint(10.25, 20.5, 3.0, 45.0, 0.03, 65793),
                  iimg.make_keypoint(1.0, 2.0, 5.0, 300.0, 0.01, 16711935)]
    im.des_list = np.arange(256, dtype=np.float32).reshape(2, 128) % 200
    im.match_list = {'IMG_8': [[0, 5], [1, 2]], 'IMG_9': []}
    im.save_features(); im.save_descriptors(); im.save_matches()
    from imageanalysis_amd import cacheio
    cacheio.wait()                        # the gzip files are written in the background
    raw = pickle.load(gzip.open(im.features_file, 'rb'))
    assert raw == [((10.25, 20.5), 3.0, 45.0, im.kp_list[0].response, 65793, -1),
                   ((1.0, 2.0), 5.0, 300.0, im.kp_list[1].response, 16711935, -1)]
    assert pickle.load(open(im.match_file, 'rb')) == im.match_list
    im2 = iimg.Image(str(an), 'IMG_7')
    assert im2.load_features() and im2.load_descriptors()
    im2.load_matches()
    assert im2.kp_list[1].pt == (1.0, 2.0) and im2.kp_list[1].octave == 16711935
    assert np.array_equal(im2.des_list, im.des_list) and im2.match_list == im.match_list


def test_kp_xy_cache_does_not_keep_the_keypoint_list_alive():
    """find_matches' periodic flush sets kp_list to None (scripts/lib/matcher.py:1008-1026); the
    cached coordinate array must not hold the KeyPoint objects."""
    import gc
    import weakref
    from imageanalysis_amd import matcher

    class KP(object):
        def __init__(self, x, y):
            self.pt = (x, y)

    class KPList(list):                       # weak-referenceable, like cacheio's lazy sequences
        pass

    class Img(object):
        pass

    for make in (KPList, list):
        im = Img()
        im.kp_list = make(KP(float(i), float(2 * i)) for i in range(100))
        probe = weakref.ref(im.kp_list[0])
        xy = matcher._kp_xy(im)
        assert xy.shape == (100, 2) and xy[7, 1] == 14.0
        assert matcher._kp_xy(im) is xy                      # cached while the list is the same
        im.kp_list = None
        gc.collect()
        assert probe() is None                               # the flush frees the objects
        im.kp_list = make(KP(1.0, 1.0) for _ in range(3))    # a reload gets fresh coordinates
        assert matcher._kp_xy(im).shape == (3, 2)


def test_gather_results_reraises_single_rank():
    from imageanalysis_amd import dist
    assert dist.gather_results([1, 2]) == [[1, 2]]
    with pytest.raises(ZeroDivisionError):
        dist.gather_results([], ZeroDivisionError("float division by zero"))


def test_match_pairs_behave_like_the_reference_lists_and_pickle_like_them(tmp_path):
    """image.match_list values are array-backed (matchpairs.MatchPairs): every list operation
    the reference's readers use gives what the list of [i, j] lists would, the `.match` file
    written from the arrays loads -- with pickle alone -- to plain lists of lists."""
    import pickle
    from imageanalysis_amd.matchpairs import MatchPairs, dumps_match_dict
    rng = np.random.default_rng(5)
    d, ref = {}, {}
    for k in range(40):
        n = int(rng.integers(0, 400))
        a = rng.integers(0, 49000 if k % 3 else 200000, (n, 2)).astype(np.int32)
        d['IMG_%04d' % k] = MatchPairs(a)
        ref['IMG_%04d' % k] = a.tolist()
    d['none'], ref['none'] = [], []
    d['host path'], ref['host path'] = [[4, 5], [6, 7]], [[4, 5], [6, 7]]
    d['neg'], ref['neg'] = MatchPairs(np.array([[-1, 70000]], np.int32)), [[-1, 70000]]
    for blob in (dumps_match_dict(d), pickle.dumps(d), pickle.dumps(d, 2)):
        got = pickle.loads(blob)
        assert got == ref and all(type(v) is list for v in got.values())
        assert all(type(p) is list and type(p[0]) is int for v in got.values() for p in v[:3])
    assert pickle.loads(dumps_match_dict({})) == {}
    m, r = d['IMG_0001'], ref['IMG_0001']
    assert np.shares_memory(np.asarray(m), m.array()) and len(m) == len(r)      # (untouched: a view)
    assert len(m) == len(r) and m[7] == r[7] and m[-1] == r[-1] and m[2:5] == r[2:5]
    assert list(m) == r and m == r and r == m and not (m != r) and [p[1] for p in m] == [p[1] for p in r]
    assert np.asarray(m, np.int64).tolist() == r and m[7] is m[7]
    assert d == ref                                   # dictionaries of them compare like lists
    m.append([1, 2]); r.append([1, 2])
    del m[0]; del r[0]
    m[3] = [9, 9]; r[3] = [9, 9]
    assert m == r and pickle.loads(dumps_match_dict({'k': m})) == {'k': r}
    e = MatchPairs(np.zeros((3, 2), np.int32))
    e[:] = np.array([[5, 6]])
    assert e == [[5, 6]] and len(MatchPairs()) == 0 and not MatchPairs()
    # through the Image object
    from imageanalysis_amd import image as iimg
    im = iimg.Image.__new__(iimg.Image)
    im.match_file = str(tmp_path / 'a.match')
    im.match_list = d
    im.save_matches()
    with open(im.match_file, 'rb') as fp:
        assert pickle.load(fp) == ref


def test_keypoint_list_is_a_list_of_keypoints_and_feat_files_stay_reference_readable(tmp_path):
    """image.kp_list is array-backed (keypoints.KeyPointList): same objects as the list the
    reference builds, `.feat` written from the columns loads with gzip + pickle alone to the
    reference's list of tuples, and a `.feat` written the reference's way loads back."""
    import gzip
    import pickle
    from imageanalysis_amd import cacheio, image as iimg
    from imageanalysis_amd.keypoints import KeyPointList
    from imageanalysis_amd.matcher import _kp_xy
    rng = np.random.default_rng(3)
    n = 5000
    x, y = rng.uniform(0, 5472, n), rng.uniform(0, 3648, n)
    size, angle, resp = rng.uniform(1, 40, n), rng.uniform(0, 360, n), rng.uniform(0, 0.2, n)
    octave = (rng.integers(0, 8, n) | (rng.integers(1, 4, n) << 8) | (rng.integers(0, 255, n) << 16))
    kl = KeyPointList(x, y, size, angle, resp, octave)
    ref = iimg.make_keypoints(x, y, size, angle, resp, octave)
    want = [(kp.pt, kp.size, kp.angle, kp.response, kp.octave, kp.class_id) for kp in ref]
    assert len(kl) == n and pickle.loads(kl.feat_bytes()) == want
    assert np.array_equal(_kp_xy(type('I', (), {'kp_list': kl})()), np.array([k.pt for k in ref], np.float32))
    # through the Image methods, read back the way the reference reads (image.py:140-160)
    im = iimg.Image.__new__(iimg.Image)
    im.features_file = str(tmp_path / 'a.feat')
    im.kp_list = kl
    im.save_features()
    cacheio.wait()
    with gzip.open(im.features_file, 'rb') as fp:
        assert pickle.load(fp) == want
    im.kp_list = None
    assert im.load_features() and isinstance(im.kp_list, KeyPointList) and im.kp_list._objs is None
    got = [(kp.pt, kp.size, kp.angle, kp.response, kp.octave, kp.class_id) for kp in im.kp_list]
    assert got == want and im.kp_list[17].pt == ref[17].pt and im.kp_list[-1].octave == ref[-1].octave
    # a file written by the reference's writer (default pickle protocol, memoised tuples)
    with gzip.open(im.features_file, 'wb') as fp:
        pickle.dump(want, fp)
    assert im.load_features()
    assert [(kp.pt, kp.size, kp.angle, kp.response, kp.octave, kp.class_id) for kp in im.kp_list] == want
    # objects persist once created; an empty file gives an empty list
    im.kp_list[5].size = 99.0
    assert im.kp_list[5].size == 99.0 and pickle.loads(im.kp_list.feat_bytes())[5][1] == 99.0
    assert KeyPointList.from_feat_bytes(pickle.dumps([])) == [] and \
        pickle.loads(KeyPointList([], [], [], [], [], []).feat_bytes()) == []


def test_pairs_per_batch_is_bounded_by_the_device_workspace():
    """find_matches sizes its device batches by what a batch needs at the survey's keypoint
    count: the per-row partial bounds of the symmetric sweep grow with rows^2 per image pair"""
    from imageanalysis_amd import matcher
    old = matcher.PAIRS_PER_BATCH
    try:
        assert matcher.BATCH_BYTES == 24 << 30
        matcher.PAIRS_PER_BATCH = 2048                   # (other tests shrink it)
        assert matcher._pairs_per_batch(4096) == 2048 and matcher._pairs_per_batch(50000) == 512
        matcher.PAIRS_PER_BATCH = 16384
        assert matcher._pairs_per_batch(4096) == 16384 and matcher._pairs_per_batch(50000) == 512
        for rows in (100, 4096, 20000, 50000, 200000):
            n = matcher._pairs_per_batch(rows)
            assert n >= 16 and n & (n - 1) == 0
            assert n == 16 or n * matcher._workspace_bytes_per_pair(rows) <= matcher.BATCH_BYTES
        matcher.PAIRS_PER_BATCH = 3                      # (tests force tiny batches)
        assert matcher._pairs_per_batch(50000) == 3
    finally:
        matcher.PAIRS_PER_BATCH = old


def test_pair_lists_pickled_in_libiamx_load_like_the_reference_lists():
    """matchpairs._pairs_bytes_many (iamx_pickle_pair_lists): narrow and wide indices, empty
    lists, many lists in one call -- each stream is what pickle.loads() turns into [[i, j], ...]"""
    import pickle
    from imageanalysis_amd import matchpairs as mp
    rng = np.random.default_rng(3)
    arrays = [rng.integers(0, 4096, (n, 2)).astype(np.int32) for n in (0, 1, 7, 300, 0, 2000)]
    arrays.append(np.array([[0, 65535], [65536, 1], [2 ** 31 - 1, 70000]], np.int32))      # wide
    arrays.append(np.array([[65535, 65535]], np.int32))                                     # still narrow
    out = mp._pairs_bytes_many(arrays)
    assert len(out) == len(arrays)
    for a, b in zip(arrays, out):
        assert pickle.loads(b'\x80\x02' + bytes(b) + b'.') == a.tolist()
    assert bytes(out[0]) == b']' and len(bytes(out[1])) == 3 + 9 and len(bytes(out[6])) == 3 + 3 * 13
    # through the cache of a MatchPairs: filled in by prepickle(), dropped by an edit
    m = mp.MatchPairs(arrays[3])
    mp.prepickle([m, mp.MatchPairs(arrays[2]), [], mp.MatchPairs()])
    assert m._pk is not None and pickle.loads(b'\x80\x02' + bytes(m.pickled()) + b'.') == arrays[3].tolist()
    m[:] = arrays[2]
    assert pickle.loads(b'\x80\x02' + bytes(m.pickled()) + b'.') == arrays[2].tolist()


def test_smart_round_form_equals_the_pair_by_pair_bookkeeping():
    """smart.record_round + materialize_pending + flush_aggregates (what find_matches uses: a
    round's entries written in one pass, averages once at the end) leaves the SAME property tree
    and the same weighted yaw errors as record_surface_estimate / record_yaw_values pair by pair"""
    from imageanalysis_amd import smart
    from imageanalysis_amd.hostlib import camera

    class Im(object):
        def __init__(self, k):
            self.name = 'Q%03d' % k
            self.ned = [3.0 * k, 17.0 * (k % 5), -100.0 - k]
            self.match_list = {}

        def get_camera_pose(self):
            return (self.ned, 10.0, -90.0, 0.0)

        def get_aircraft_pose(self):
            return (0.0, 0.0, 0.0), (12.0 + len(self.name), 0.0, 0.0)

    camera.set_image_params(5472, 3648)
    rng = np.random.default_rng(4)
    imgs = [Im(k) for k in range(12)]
    pairs = []
    for a in range(12):
        for b in range(a + 1, 12):
            if rng.random() < 0.6:
                dist = float(np.linalg.norm(np.array(imgs[b].ned) - np.array(imgs[a].ned)))
                avg, std = float(rng.normal(-3, 8)), float(abs(rng.normal(6, 12)))
                yf = tuple(float(v) for v in (rng.normal(0, 20), dist, rng.uniform(0, 360), rng.uniform(0, 9)))
                yr = tuple(float(v) for v in (rng.normal(0, 20), dist, rng.uniform(0, 360), rng.uniform(0, 9)))
                pairs.append((imgs[a], imgs[b], avg, std, dist,
                              yf if rng.random() < 0.8 else None, yr if rng.random() < 0.8 else None))
    assert len(pairs) > 25

    def reset():
        smart.smart_node.__dict__.clear()
        smart.load(None)
        for im in imgs:
            smart.smart_node.getChild(im.name, True).setFloat('tri_surface_m', 0.0)

    # (a) pair by pair, averages after every pair (the reference's order of operations)
    reset()
    yaw_a = {}
    for i1, i2, avg, std, dist, yf, yr in pairs:
        smart.record_surface_estimate(i1, i2, avg, std, dist)
        for me, other, yv in ((i1, i2, yf), (i2, i1, yr)):
            y = smart.record_yaw_values(me, other, yv)
            if yv is not None:
                yaw_a[me.name] = y
    tree_a = smart._to_dict(smart.smart_node)

    # (b) the round form: three "rounds", one flush at the end
    reset()
    smart.begin_batch()
    for c0 in range(0, len(pairs), 11):
        smart.record_round(pairs[c0:c0 + 11])
        smart.materialize_pending()
    yaw_b = smart.flush_aggregates()
    tree_b = smart._to_dict(smart.smart_node)
    smart.freeze_poses(False)
    assert tree_b == tree_a
    assert set(yaw_b) >= set(yaw_a)
    for name, y in yaw_a.items():
        assert yaw_b[name] == y
    reset()


def test_pair_schedule_matrix_form_equals_the_gather_form():
    """matcher._work_arrays builds the schedule from the n x n distance matrix for surveys up to
    4096 images: same pairs, same (bit-identical) distances, same stable order as the
    triu_indices + gather form that larger surveys still take, for all three schedules"""
    from imageanalysis_amd import matcher

    class Im(object):
        def __init__(self, ned):
            self.ned = ned

        def get_camera_pose(self):
            return (self.ned, 0.0, -90.0, 0.0)

    class Proj(object):
        pass

    rng = np.random.default_rng(8)
    proj = Proj()
    proj.image_list = [Im([20.0 * (k // 9) + rng.normal(0, 2.0), 20.0 * (k % 9) + rng.normal(0, 2.0),
                           -100.0 + rng.normal(0, 1.0)]) for k in range(70)]
    node = matcher.matcher_node
    try:
        for schedule, extra in (('all-pairs', {}), ('neighbours', {}), ('distance', {'min_dist': 15.0, 'max_dist': 70.0})):
            node.setString('schedule', schedule)
            for k, v in extra.items():
                node.setFloat(k, v)
            for sort in (True, False):
                got = matcher._work_arrays(proj, sort)
                keep = matcher._WORK_MATRIX_MAX
                matcher._WORK_MATRIX_MAX = 0
                try:
                    want = matcher._work_arrays(proj, sort)
                finally:
                    matcher._WORK_MATRIX_MAX = keep
                assert len(got[0]) == len(want[0]) > 0
                for a, b in zip(got, want):
                    assert a.dtype == b.dtype and np.array_equal(a, b)
                if sort:
                    # the order python's stable sorted() gives: by rounded distance, row-major inside
                    dd, ii, jj = got
                    assert (np.diff(dd) >= 0).all()
                    same = np.diff(dd) == 0
                    rm = ii.astype(np.int64) * 100000 + jj
                    assert (np.diff(rm)[same] > 0).all()
    finally:
        node.setString('schedule', 'neighbours')
        for k in ('min_dist', 'max_dist'):
            if node.hasChild(k):
                node.__dict__.pop(k, None)


def test_detector_slots_are_private_per_thread_and_always_returned():
    """kernels.detector_slot: concurrent holders get distinct slots (their device buffers are
    keyed by the slot), at most DETECT_SLOTS at a time, and a slot goes back on an exception"""
    import threading
    import time
    from imageanalysis_amd import kernels

    class Dev(object):
        index = 0

    assert kernels._slot_key(Dev()) == (0, 0)               # outside any slot: the shared slot 0
    held, peak, lock = [], [0], threading.Lock()

    def worker():
        with kernels.detector_slot() as s:
            key = kernels._slot_key(Dev())
            assert key == (0, s.slot) and s.slot >= 1
            with lock:
                held.append(s.slot)
                assert len(set(held)) == len(held)           # nobody else holds this slot now
                peak[0] = max(peak[0], len(held))
            time.sleep(0.02)
            with lock:
                held.remove(s.slot)
        assert kernels._slot_key(Dev()) == (0, 0)

    threads = [threading.Thread(target=worker) for _ in range(3 * kernels.DETECT_SLOTS)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert 1 <= peak[0] <= kernels.DETECT_SLOTS
    with pytest.raises(RuntimeError):
        with kernels.detector_slot():
            raise RuntimeError('inside a slot')
    got = []
    for _ in range(kernels.DETECT_SLOTS):                    # every slot is back in the pool
        s = kernels.detector_slot()
        s.__enter__()
        got.append(s)
    assert sorted(x.slot for x in got) == list(range(1, kernels.DETECT_SLOTS + 1))
    for s in got:
        s.__exit__(None, None, None)


def test_quiet_pair_overwrites_a_stale_entry():
    """the reference assigns match_list[name] = [] for a pair without matches
    unconditionally (scripts/lib/matcher.py:978-979).  A one-sided non-empty entry left by an
    interrupted save must not survive the retry: value replaced in place, key order kept."""
    from imageanalysis_amd.matchpairs import MatchDict, QuietLedger
    names = ['a', 'b', 'c']
    led = QuietLedger(names)
    da, db = MatchDict({'c': [[5, 6]], 'b': [[1, 2]]}), MatchDict()
    da.attach(led, 0)
    db.attach(led, 1)
    led.add([0], [1], [7])                               # pair (a, b) ended without matches
    assert da.entries() == [('c', [[5, 6]]), ('b', [])] or \
        [(k, list(v)) for k, v in da.entries()] == [('c', [[5, 6]]), ('b', [])]
    assert [(k, list(v)) for k, v in db.entries()] == [('a', [])]
    import pickle
    assert pickle.loads(pickle.dumps(da)) == {'c': [[5, 6]], 'b': []}
    assert list(pickle.loads(pickle.dumps(da)).keys()) == ['c', 'b']
    assert dict(db) == {'a': []}


def test_feat_bytes_without_the_library(monkeypatch):
    """KeyPointList.feat_bytes() is file formatting; without libiamx.so (a CPU-only
    tool re-saving a .feat) it writes the same bytes from numpy instead of raising"""
    import pickle
    from imageanalysis_amd import _lib
    from imageanalysis_amd.keypoints import KeyPointList
    rng = np.random.default_rng(4)
    n = 300
    kl = KeyPointList(rng.uniform(0, 5000, n), rng.uniform(0, 3000, n), rng.uniform(1, 9, n),
                      rng.uniform(0, 360, n), rng.uniform(0, 0.1, n), rng.integers(0, 1 << 24, n))
    with_lib = kl.feat_bytes()

    def no_lib():
        raise _lib.IamxError("libiamx.so not found (test)")
    monkeypatch.setattr(_lib, 'lib', no_lib)
    assert kl.feat_bytes() == with_lib
    rows = pickle.loads(with_lib)
    assert len(rows) == n and rows[7][0] == (float(kl.x[7]), float(kl.y[7])) and rows[7][5] == -1


@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
def test_optimizer_setup_and_refit_on_the_array_backed_chains(path):
    """Optimizer.setup() / refit() on match_cleanup.Chains (the arrays link_matches() returns) ==
    the same calls on the reference's list of lists: every array of the setup, the feature maps,
    the points refit() writes back"""
    import copy
    from imageanalysis_amd import match_cleanup, optimizer
    g = np.load(path)
    out = {}
    for form in ('lists', 'arrays'):
        proj, inp = _scene(path)
        matches = copy.deepcopy(inp['matches'])
        if form == 'arrays':
            matches = match_cleanup.Chains.from_lists(matches)
        opt = optimizer.Optimizer('/nonexistent')
        opt.setup(proj, inp['groups'], 0, matches, cam_calib=bool(g['cam_calib']))
        C, P = opt.n_cameras, opt.n_points
        xf = g['x_final']
        setup = dict(ci=opt.camera_indices.copy(), pi=opt.point_indices.copy(), p3=np.array(opt.points_3d),
                     bi=[a.copy() for a in opt.by_camera_point_indices],
                     b2=[a.copy() for a in opt.by_camera_points_2d],
                     fwd=dict(opt.feat_map_fwd), rev=dict(opt.feat_map_rev), cp=np.array(opt.camera_params))
        opt.camera_params = xf[:C * 7].reshape(C, 7)
        opt.points_3d = xf[C * 7:C * 7 + P * 3].reshape(P, 3)
        opt.update_camera_poses(proj)
        opt.refit(proj, matches, inp['groups'], 0)
        if form == 'arrays':
            assert matches.untouched()
        out[form] = (setup, pickle.loads(pickle.dumps(matches)))
    (a, ma), (b, mb) = out['lists'], out['arrays']
    for k in ('ci', 'pi', 'p3', 'cp'):
        assert np.array_equal(a[k], b[k]), k
    assert a['fwd'] == b['fwd'] and a['rev'] == b['rev']
    for x, y in zip(a['bi'] + a['b2'], b['bi'] + b['b2']):
        assert np.array_equal(x, y)
    assert len(ma) == len(mb)
    for x, y in zip(ma, mb):
        assert x[1:] == y[1:]
        assert (x[0] is None) == (y[0] is None)
        if x[0] is not None:
            assert np.allclose(x[0], y[0], rtol=0, atol=0)


def test_2d_trust_region_solver_equals_scipys():
    """ba_solver.solve_tr_2d against scipy/optimize/_lsq/common.py solve_trust_region_2d -- the
    subproblem trf.py solves per trial step (trf.py:328) --: the same case (Newton step or
    boundary) and the same model value, over definite, singular and badly scaled B."""
    from scipy.optimize._lsq.common import solve_trust_region_2d
    from imageanalysis_amd.ba_solver import solve_tr_2d
    rng = np.random.default_rng(3)
    newton = boundary = 0
    for k in range(2000):
        Q = rng.normal(size=(2, 2))
        B = Q @ Q.T * 10 ** rng.uniform(-4, 4)
        if k % 7 == 0:                          # s1 = 0: the second row and column vanish
            B[1, :] = 0
            B[:, 1] = 0
        g = np.array([10 ** rng.uniform(-4, 4), 0.0 if k % 2 else rng.normal()])
        Delta = 10 ** rng.uniform(-4, 4)
        want, want_newton = solve_trust_region_2d(B, g, Delta)
        got, got_newton = solve_tr_2d(B, g, Delta)
        assert got_newton == want_newton
        value = lambda p: 0.5 * p @ B @ p + g @ p          # noqa: E731
        assert abs(value(got) - value(want)) <= 1e-10 * abs(value(want))
        assert np.linalg.norm(got) <= Delta * (1 + 1e-12)
        if not np.allclose(B[1], 0):            # (a unique minimiser)
            assert np.allclose(got, want, rtol=1e-6, atol=1e-9 * Delta)
        newton += got_newton
        boundary += not got_newton
    assert newton > 100 and boundary > 100


def test_round_bulk_routines_equal_numpy():
    """libiamx host routines of a find_matches round (lib/matcher.py:978-979 lists, lib/smart.py:
    117-130 statistics): the threaded copy + column swap, and per-pair mean / std of the packed
    heights BIT-equal to the numpy expressions they replace (np.add.reduceat sums)."""
    import ctypes
    from imageanalysis_amd import _lib
    L = _lib.lib()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)          # noqa: E731
    rng = np.random.default_rng(11)
    for n in (0, 1, 5, 70000, 300001):
        src = rng.integers(0, 50000, (n, 2)).astype(np.int32)
        fwd, rev = np.full_like(src, -1), np.full_like(src, -1)
        _lib.check(L.iamx_pairs_fwd_rev(P(src), n, P(fwd), P(rev), 5), 'iamx_pairs_fwd_rev')
        assert np.array_equal(fwd, src) and np.array_equal(rev, src[:, ::-1])
    # segments of 1 .. 3000 values: below 8, up to 128, and the recursive halves of numpy's sum
    c = np.concatenate([np.arange(1, 300), rng.integers(1, 3000, 400), [1, 1, 7, 8, 9, 127, 128, 129, 2999]])
    c = c.astype(np.int64)
    z = np.ascontiguousarray(rng.normal(-120.0, 35.0, int(c.sum())))
    z[::97] *= 1e6
    starts = np.ascontiguousarray(np.concatenate([[0], np.cumsum(c)[:-1]]), np.int64)
    mean_np = np.add.reduceat(z, starts) / c
    std_np = np.sqrt(np.add.reduceat((z - np.repeat(mean_np, c)) ** 2, starts) / c)
    for threads in (1, 4):
        mean, std = np.empty(len(c)), np.empty(len(c))
        _lib.check(L.iamx_segment_mean_std(P(z), P(starts), P(c), len(c), len(z), P(mean), P(std), threads),
                   'iamx_segment_mean_std')
        assert np.array_equal(mean, mean_np) and np.array_equal(std, std_np)
    # and what the reference computes per pair: np.mean / np.std of the pair's values
    k = 431
    seg = z[starts[k]:starts[k] + c[k]]
    assert np.isclose(mean[k], np.mean(seg), rtol=1e-14) and np.isclose(std[k], np.std(seg), rtol=1e-12)
    bad = np.array([0], np.int64)
    assert L.iamx_segment_mean_std(P(z), P(starts), P(bad), 1, len(z), P(mean), P(std), 1) != 0


def test_quiet_ledger_grows_and_indexes_like_the_naive_form():
    """QuietLedger appends a round's quiet pairs into preallocated columns (growing them when the
    announced capacity was too small) and hands every image its partners in processing order --
    what a dictionary entry per pair and direction, assigned pair after pair
    (scripts/lib/matcher.py:978-979), would hold."""
    from imageanalysis_amd.matchpairs import QuietLedger
    rng = np.random.default_rng(5)
    names = ['q%02d' % k for k in range(37)]
    for capacity in (0, 10, 5000):
        led = QuietLedger(names, capacity=capacity)
        want = {k: [] for k in range(len(names))}
        seq = 0
        for _round in range(40):
            m = int(rng.integers(0, 90))
            i = rng.integers(0, len(names), m)
            j = (i + 1 + rng.integers(0, len(names) - 1, m)) % len(names)
            s = np.arange(seq, seq + m)
            seq += m
            led.add(i, j, s)
            for a, b, t in zip(i.tolist(), j.tolist(), s.tolist()):
                want[a].append((b, t))
                want[b].append((a, t))
        assert len(led) == seq
        for k in range(len(names)):
            other, sq = led.partners_of(k)
            assert list(zip(other.tolist(), sq.tolist())) == want[k]
        led.add([0], [1], [seq])                        # an add after the index was built
        other, sq = led.partners_of(1)
        assert (int(other[-1]), int(sq[-1])) == (0, seq)


def test_huge_page_backed_arrays_behave_like_numpy_arrays():
    """matchpairs.empty_huge: the arrays a round's match lists are views of -- shape, dtype,
    contiguity, writability, a view keeping the memory alive, the plain fallback below 8 MiB"""
    import gc
    from imageanalysis_amd.matchpairs import empty_huge
    small = empty_huge((100, 2), np.int32)
    assert small.shape == (100, 2) and small.dtype == np.int32 and small.flags['OWNDATA']
    big = empty_huge((3_000_000, 2), np.int32)           # 24 MB
    assert big.shape == (3_000_000, 2) and big.dtype == np.int32
    assert big.flags['C_CONTIGUOUS'] and big.flags['WRITEABLE'] and big.ctypes.data % 4096 == 0
    big[:] = np.arange(6_000_000, dtype=np.int32).reshape(-1, 2)
    view = big[1_000_000:1_000_005]
    del big
    gc.collect()
    assert view[:, 0].tolist() == [2_000_000, 2_000_002, 2_000_004, 2_000_006, 2_000_008]
    flat = empty_huge(9_000_000, np.uint8)
    assert flat.shape == (9_000_000,) and flat.dtype == np.uint8
    flat[-1] = 7
    assert int(flat[-1]) == 7


def test_native_kp_keys_and_dup_remap_equal_the_numpy_form():
    """iamx_kp_key2 / iamx_kp_dup_remap (round 5: merge_duplicates' per-image tables in one native
    call) against the numpy statement of the same integers, on random positions, on exact halves
    (x.xx5 values that are float32-representable: round-half-even decides) and on shared pixels"""
    import ctypes
    from imageanalysis_amd import _lib, matcher
    rng = np.random.default_rng(77)
    n = 5000
    x = rng.uniform(0, 5472, (n, 2)).astype(np.float32)
    x[:200] = (rng.integers(0, 40000, (200, 2)) * 0.125).astype(np.float32)         # k/8: many exact .125 / .375 / .625 / .875
    x[200:300] = np.float32(2 ** -17) * rng.integers(0, 3, (100, 2))              # below 2^-16: prints as 0.00
    x[300:320] = [[0.005, 0.015], [0.025, 1.005]] * 10
    def numpy_key(xy):
        v = np.asarray(xy, np.float64)
        m = (v * float(1 << 40)).astype(np.int64) * 100
        q, rem = m >> 40, m & ((1 << 40) - 1)
        half = 1 << 39
        return (q + ((rem > half) | ((rem == half) & ((q & 1) == 1)))).astype(np.int32)
    want = numpy_key(x)
    got = matcher.kp_key2(x)
    assert got.dtype == np.int32 and np.array_equal(got, want)
    # ... and the strings they stand for
    for k in rng.integers(0, n, 300):
        assert ("%.2f-%.2f" % (float(x[k, 0]), float(x[k, 1]))) == "%d.%02d-%d.%02d" % (
            want[k, 0] // 100, want[k, 0] % 100, want[k, 1] // 100, want[k, 1] % 100)
    with pytest.raises(ValueError):
        matcher.kp_key2(np.full((300, 2), 20000.0, np.float32))
    # duplicate remap: three "images" back to back
    counts = [1500, 0, 3500]
    base = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    xy = x.copy()
    twins = rng.permutation(1500)[:300]
    xy[twins] = xy[(twins + 7) % 1500]                     # shared pixels inside image 0
    used = (rng.random(n) < 0.7).astype(np.uint8)
    remap = np.empty(n, np.int32)
    ident = np.zeros(3, np.uint8)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _lib.check(_lib.lib().iamx_kp_dup_remap(P(xy), P(used), P(base), 3, P(remap), P(ident), 2), 'iamx_kp_dup_remap')
    for i, (b, e) in enumerate(zip(base[:-1], base[1:])):
        want_r = np.arange(e - b)
        u = np.nonzero(used[b:e])[0]
        if len(u):
            k2 = numpy_key(xy[b:e][u]).astype(np.int64)
            _uq, first, inv = np.unique((k2[:, 0] << 32) | k2[:, 1], return_index=True, return_inverse=True)
            want_r[u] = u[first[inv]]
        assert np.array_equal(remap[b:e], want_r), i
        assert bool(ident[i]) == bool((want_r == np.arange(e - b)).all())
    assert not ident[0] and ident[1]


def test_device_memory_model_of_the_matching_stage():
    """matcher.device_memory_model (DESIGN section 3 "memory at scale"): only the arena grows with
    the survey; the figures quoted for BASELINE configs[4]'s 10 000 frames"""
    from imageanalysis_amd import matcher
    m = matcher.device_memory_model(10000, 37000)
    rows = (37000 + 127) // 128 * 128
    assert m['arena_bytes'] == 10000 * rows * 284 + 10000 * 37000 * 16
    assert 100e9 < m['arena_bytes'] < 112e9
    assert m['workspace_bytes_each'] <= matcher.BATCH_BYTES
    assert m['peak_bytes'] < 288e9                                  # fits one MI355X
    with_copy = matcher.device_memory_model(10000, 37000, train_layout=True)
    assert with_copy['arena_bytes'] - m['arena_bytes'] == 10000 * rows * 140
    small = matcher.device_memory_model(128, 38000)
    assert small['workspace_bytes_each'] == m['workspace_bytes_each'] or small['pairs_per_batch'] >= m['pairs_per_batch']
    assert small['arena_bytes'] < 1.6e9
