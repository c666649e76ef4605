#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE ONLY.

Generates the golden vectors under tests/golden/ by importing and running the
REFERENCE'S OWN Python (/root/reference/scripts/lib/{matcher,optimizer,image,
camera}.py) in this container, with oracle/shims/ standing in for the
third-party modules that are not installed (cv2, props, props_json, navpy) and
`transformations` taken from the reference's archived copy.  Runs only here
(the GPU box has no /root/reference); the .npz/.pkl outputs are committed.

    python oracle/gen_golden.py            # regenerate everything

What each fixture pins:
  G1 match_*.npz   lib/matcher.py:203-347 raw_matches -> metric filter/sort/clip
                   -> GMS -> filter_duplicates -> cross-check, both directions
  G2 ba_*.npz      lib/optimizer.py:174-279 Optimizer.fun residual vector
                   (+ nedquat2rvectvec :120-126), layout and conventions
  G3 (same files)  scipy approx_derivative of that fun with the reference's
                   sparsity mask (:142-169) -- oracle for the analytic Jacobian
  G4 (same files)  Optimizer.run() end state (x*, cost, njev)
  G5 (same files)  Optimizer.setup() index structures
  G6 ba_*_refit.pkl update_camera_poses()/refit() outputs
"""
import contextlib
import io
import math
import os
import pickle
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, os.path.join(HERE, 'shims'))
sys.path.insert(0, '/root/reference/scripts')
sys.path.append('/root/reference/scripts/lib/archive')      # -> transformations

import numpy as np                                           # noqa: E402
import cv2                                                   # noqa: E402  (the shim)
from props import getNode, root                              # noqa: E402
from lib import camera, matcher, optimizer                   # noqa: E402
from lib import image as ref_image                           # noqa: E402
from scipy.optimize._numdiff import approx_derivative, group_columns  # noqa: E402
import transformations as _tf                                # noqa: E402  (reference's archived copy)


class _Numpy1Compat(object):
    """The archived transformations.py targets numpy 1.x: ``numpy.array(x, copy=False)``
    meant "copy only if needed"; numpy 2 (installed here) raises instead.  Give that
    module the 1.x meaning without touching the reference file."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def array(obj, *a, **k):
        if k.get('copy', True) is False:
            k.pop('copy')
            return np.asarray(obj, *a, **k)
        return np.array(obj, *a, **k)


_tf.numpy = _Numpy1Compat()

W_PX, H_PX = 5472, 3648
FX = 3666.6665
K_FC6310S = [FX, 0.0, 2736.0, 0.0, FX, 1824.0, 0.0, 0.0, 1.0]


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


# ---------------------------------------------------------------------------
# synthetic SIFT-like data
# ---------------------------------------------------------------------------
def sift_like(rng, n):
    g = rng.gamma(0.6, 1.0, size=(n, 128))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    g = np.minimum(g, 0.2)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    return np.clip(np.rint(g * 512.0), 0, 255).astype(np.uint8)


def make_pair(seed, n1, n2, n_true, noise=6, shift=(400.0, -250.0), dup_uv=0,
              tie_rows=0):
    """image 2 = a random subset of image 1's features (noisy descriptors, shifted
    pixels) + unrelated features."""
    rng = np.random.default_rng(seed)
    des1 = sift_like(rng, n1)
    xy1 = np.stack([rng.uniform(450, W_PX - 450, n1), rng.uniform(300, H_PX - 300, n1)], 1)
    des2 = sift_like(rng, n2)
    xy2 = np.stack([rng.uniform(0, W_PX - 1, n2), rng.uniform(0, H_PX - 1, n2)], 1)
    src = rng.permutation(n1)[:n_true]
    dst = rng.permutation(n2)[:n_true]
    nz = rng.integers(-noise, noise + 1, size=(n_true, 128))
    des2[dst] = np.clip(des1[src].astype(np.int64) + nz, 0, 255).astype(np.uint8)
    xy2[dst] = xy1[src] + np.array(shift) + rng.normal(0, 0.7, size=(n_true, 2))
    if dup_uv:
        # SIFT emits the same pixel at several orientations: same pt, other descriptor
        a = rng.permutation(n_true)[:dup_uv]
        for k in a:
            j = int(rng.integers(0, n1))
            xy1[j] = xy1[src[k]]
            des1[j] = np.clip(des1[src[k]].astype(np.int64)
                              + rng.integers(-3, 4, size=128), 0, 255).astype(np.uint8)
    if tie_rows:
        # exact distance ties: duplicate train descriptors at other indices
        a = rng.permutation(n2)[:tie_rows]
        b = rng.permutation(n2)[:tie_rows]
        des2[b] = des2[a]
    xy1 = xy1.astype(np.float32)
    xy2 = np.clip(xy2, 0, [W_PX - 1, H_PX - 1]).astype(np.float32)
    return des1, xy1, des2, xy2


class FakeImage(object):
    """Carries exactly what lib/matcher.py reads from an Image."""

    def __init__(self, name, des_u8, xy):
        self.name = name
        self.des_list = des_u8.astype(np.float32)      # cv2.SIFT output dtype
        self.kp_list = [cv2.KeyPoint(float(x), float(y), 3.0) for x, y in xy]
        self.match_list = {}


def run_match_case(name, seed, n1, n2, n_true, match_ratio=0.75, min_pairs=25, **kw):
    des1, xy1, des2, xy2 = make_pair(seed, n1, n2, n_true, **kw)
    i1 = FakeImage('A', des1, xy1)
    i2 = FakeImage('B', des2, xy2)

    getNode('/config/detector', True).setString('detector', 'SIFT')
    getNode('/config/detector', True).setFloat('scale', 0.4)
    mnode = getNode('/config/matcher', True)
    mnode.setFloat('match_ratio', match_ratio)
    mnode.setInt('min_pairs', min_pairs)
    camera.set_image_params(W_PX, H_PX)
    matcher.configure()

    # record what the reference hands to / gets from matchGMS
    gms_log = []
    real_gms = cv2.xfeatures2d.matchGMS

    def spy(size1, size2, kp1, kp2, matches, **kwargs):
        out = real_gms(size1, size2, kp1, kp2, matches, **kwargs)
        gms_log.append((np.array([[m.queryIdx, m.trainIdx] for m in matches], np.int32).reshape(-1, 2),
                        np.array([[m.queryIdx, m.trainIdx] for m in out], np.int32).reshape(-1, 2)))
        return out
    cv2.xfeatures2d.matchGMS = spy
    try:
        with quiet():
            knn_f = matcher.raw_matches(i1, i2)
            knn_r = matcher.raw_matches(i2, i1)
            basic_f = matcher.basic_pair_matches(i1, i2)
            n_gms_f = len(gms_log)
            basic_r = matcher.basic_pair_matches(i2, i1)
            gms_f = gms_log[0] if n_gms_f else (np.zeros((0, 2), np.int32),) * 2
            gms_r = gms_log[n_gms_f] if len(gms_log) > n_gms_f else (np.zeros((0, 2), np.int32),) * 2
            fwd, rev = matcher.bidirectional_pair_matches(i1, i2)
    finally:
        cv2.xfeatures2d.matchGMS = real_gms

    def knn_arrays(knn):
        idx = np.array([[m[0].trainIdx, m[1].trainIdx] for m in knn], np.int32)
        dist = np.array([[m[0].distance, m[1].distance] for m in knn], np.float32)
        return idx, dist

    kf_idx, kf_dist = knn_arrays(knn_f)
    kr_idx, kr_dist = knn_arrays(knn_r)
    arr = lambda l: np.array(l, np.int32).reshape(-1, 2)
    out = dict(des1=des1, xy1=xy1, des2=des2, xy2=xy2,
               match_ratio=np.float64(match_ratio), min_pairs=np.int32(min_pairs),
               width=np.int32(W_PX), height=np.int32(H_PX),
               knn_fwd_idx=kf_idx, knn_fwd_dist=kf_dist,
               knn_rev_idx=kr_idx, knn_rev_dist=kr_dist,
               pregms_fwd=gms_f[0], postgms_fwd=gms_f[1],
               pregms_rev=gms_r[0], postgms_rev=gms_r[1],
               basic_fwd=arr(basic_f), basic_rev=arr(basic_r),
               bidir_fwd=arr(fwd), bidir_rev=arr(rev))
    np.savez_compressed(os.path.join(GOLD, 'match_%s.npz' % name), **out)
    print('G1 %-10s n1=%d n2=%d pregms=%d/%d basic=%d/%d bidir=%d/%d' % (
        name, n1, n2, len(gms_f[0]), len(gms_r[0]), len(basic_f), len(basic_r),
        len(fwd), len(rev)))


# ---------------------------------------------------------------------------
# BA scenes through the reference's Optimizer
# ---------------------------------------------------------------------------
class FakeProj(object):
    def __init__(self, names, analysis_dir):
        self.analysis_dir = analysis_dir
        self.image_list = [ref_image.Image(analysis_dir, n) for n in names]

    def findIndexByName(self, name):
        for i, im in enumerate(self.image_list):
            if im.name == name:
                return i
        return None

    def findImageByName(self, name):
        for im in self.image_list:
            if im.name == name:
                return im
        return None

    def save_images_info(self):
        pass


def make_ba_scene(seed, rows, cols, n_pts, dist, spacing=30.0, agl=100.0,
                  cam_sigma=1.0, pt_sigma=2.0, px_sigma=0.5, extra_images=2):
    """Nadir cameras on a lawn-mower grid, ground points,
    noisy projections.  Returns the structures process.py would hand to setup()."""
    rng = np.random.default_rng(seed)
    n_cam = rows * cols
    names = ['IMG_%04d' % i for i in range(n_cam + extra_images)]
    with quiet():
        getNode('/config/directories', True).setString('project_dir', '/nonexistent')
        proj = FakeProj(names, '/nonexistent/ImageAnalysis')
    cnode = getNode('/config/camera', True)
    cnode.setLen('K', 9)
    for i, v in enumerate(K_FC6310S):
        cnode.setFloatEnum('K', i, v)
    camera.set_dist_coeffs(list(dist))
    camera.set_image_params(W_PX, H_PX)

    true_cams = []
    for i, im in enumerate(proj.image_list):
        r, c = divmod(i, cols)
        if r % 2:
            c = cols - 1 - c
        ned = np.array([r * spacing, c * spacing, -agl]) + rng.normal(0, 0.3, 3)
        yaw = (0.0 if r % 2 == 0 else 180.0) + rng.normal(0, 3.0)
        pitch = -90.0 + rng.normal(0, 2.0)
        roll = rng.normal(0, 2.0)
        true_cams.append((ned, yaw, pitch, roll))
        im.set_camera_pose(ned.tolist(), yaw, pitch, roll)

    opt = optimizer.Optimizer('/nonexistent')
    K = camera.get_K()
    dc = camera.get_dist_coeffs()
    pts = np.stack([rng.uniform(-40, (rows - 1) * spacing + 40, n_pts),
                    rng.uniform(-60, (cols - 1) * spacing + 60, n_pts),
                    rng.normal(0, 2.0, n_pts)], 1)
    matches = []
    for p in pts:
        obs = []
        for ci, im in enumerate(proj.image_list):
            ned, ypr, quat = im.get_camera_pose()
            rvec, tvec = opt.nedquat2rvectvec(ned, quat)
            uv, _ = cv2.projectPoints(p.reshape(1, 3), rvec, tvec, K, dc)
            u, v = uv.ravel()
            Xc = cv2.Rodrigues(rvec)[0] @ p + np.asarray(tvec).ravel()
            if Xc[2] > 1.0 and 0 <= u < W_PX and 0 <= v < H_PX:
                obs.append([ci, [float(u + rng.normal(0, px_sigma)),
                                 float(v + rng.normal(0, px_sigma))]])
        if len(obs) >= 2:
            guess = (p + rng.normal(0, pt_sigma, 3)).tolist()
            matches.append([guess, 0] + obs)
    # some matches belong to another group / are unassigned, some are too short
    for k in range(0, len(matches), 17):
        matches[k][1] = -1
    for k in range(5, len(matches), 23):
        matches[k][1] = 1
    # perturb the camera poses away from truth (inside the +-3 m bounds)
    for (ned, yaw, pitch, roll), im in zip(true_cams, proj.image_list):
        im.set_camera_pose((ned + rng.normal(0, cam_sigma, 3)).tolist(),
                           yaw + rng.normal(0, 1.0), pitch + rng.normal(0, 1.0),
                           roll + rng.normal(0, 1.0))
    groups = [names[:n_cam], names[n_cam:]]
    return proj, groups, matches


def run_ba_case(name, seed, rows, cols, n_pts, dist, cam_calib=False, solve=True):
    proj, groups, matches = make_ba_scene(seed, rows, cols, n_pts, dist)
    matches_in = pickle.loads(pickle.dumps(matches))
    poses_in = [im.get_camera_pose() for im in proj.image_list]
    opt = optimizer.Optimizer('/nonexistent')
    with quiet():
        opt.setup(proj, groups, 0, matches, optimized=False, cam_calib=cam_calib)
    C, P = opt.n_cameras, opt.n_points
    if cam_calib:
        x0 = np.hstack((opt.camera_params.ravel(), opt.points_3d.ravel(),
                        opt.K[0, 0], opt.K[0, 2], opt.K[1, 2], opt.distCoeffs))
    else:
        x0 = np.hstack((opt.camera_params.ravel(), opt.points_3d.ravel()))
    args = (C, P, opt.by_camera_point_indices, opt.by_camera_points_2d)
    with quiet():
        f0 = opt.fun(x0, *args).copy()
        A = opt.bundle_adjustment_sparsity(C, P, opt.camera_indices, opt.point_indices)
        groups_cols = group_columns(A)
        J2 = approx_derivative(opt.fun, x0, method='2-point', sparsity=(A, groups_cols),
                               args=args).tocsr()
        J3 = approx_derivative(opt.fun, x0, method='3-point', sparsity=(A, groups_cols),
                               args=args).tocsr()
    rt = [opt.nedquat2rvectvec(c[:3], c[3:7]) for c in opt.camera_params.reshape(C, 7)]
    out = dict(
        K=np.asarray(opt.K, np.float64), dist=np.asarray(opt.distCoeffs, np.float64),
        cam_calib=np.int32(cam_calib), n_cameras=np.int32(C), n_points=np.int32(P),
        x0=x0, f0=f0,
        camera_indices=opt.camera_indices.astype(np.int32),
        point_indices=opt.point_indices.astype(np.int32),
        points_2d=np.concatenate([a.reshape(-1, 2) for a in opt.by_camera_points_2d if len(a)]),
        by_camera_counts=np.array([len(a) for a in opt.by_camera_point_indices], np.int32),
        camera_map_fwd=np.array([opt.camera_map_fwd[i] for i in range(C)], np.int32),
        feat_map_rev=np.array([opt.feat_map_rev[i] for i in range(P)], np.int32),
        rvecs=np.array([np.asarray(r).ravel() for r, t in rt]),
        tvecs=np.array([np.asarray(t).ravel() for r, t in rt]),
        J_sparsity_nnz=np.int64(A.nnz),
        J2_data=J2.data, J2_indices=J2.indices.astype(np.int32), J2_indptr=J2.indptr.astype(np.int32),
        J3_data=J3.data, J3_indices=J3.indices.astype(np.int32), J3_indptr=J3.indptr.astype(np.int32),
    )
    msg = 'G2/3/5 %-12s C=%d P=%d O=%d mre0=%.3f groups=%d' % (
        name, C, P, opt.camera_indices.size, np.mean(np.abs(f0)), int(groups_cols.max()) + 1)
    if solve:
        with quiet():
            (cams, feats, cmap, fmap, fx, fy, cu, cv, dc) = opt.run()
        res_x = np.hstack((cams.ravel(), feats.ravel()))
        if cam_calib:
            res_x = np.hstack((res_x, fx, cu, cv, dc))
        with quiet():
            opt.last_mre = None
            f_fin = opt.fun(res_x, *args).copy()
        out.update(x_final=res_x, f_final=f_fin,
                   cost_final=np.float64(0.5 * f_fin @ f_fin),
                   ret_fx=np.float64(fx), ret_fy=np.float64(fy), ret_cu=np.float64(cu),
                   ret_cv=np.float64(cv), ret_dist=np.asarray(dc, np.float64))
        msg += ' mre*=%.4f' % np.mean(np.abs(f_fin))
        # G6: pose write-back and similarity refit
        with quiet():
            opt.update_camera_poses(proj)
            poses_opt = [im.get_camera_pose(opt=True) for im in proj.image_list]
            valid = [bool(im.node.getChild('camera_pose_opt', True).getBool('valid'))
                     for im in proj.image_list]
            opt.refit(proj, matches, groups, 0)
            poses_refit = [im.get_camera_pose(opt=True) for im in proj.image_list]
        with open(os.path.join(GOLD, 'ba_%s_refit.pkl' % name), 'wb') as f:
            pickle.dump(dict(poses_opt=poses_opt, valid=valid, poses_refit=poses_refit,
                             matches_points=[m[0] for m in matches]), f, protocol=4)
    with open(os.path.join(GOLD, 'ba_%s_in.pkl' % name), 'wb') as f:
        pickle.dump(dict(names=[im.name for im in proj.image_list], groups=groups,
                         matches=matches_in, poses=poses_in,
                         K=K_FC6310S, dist=list(map(float, dist)),
                         width=W_PX, height=H_PX), f, protocol=4)
    np.savez_compressed(os.path.join(GOLD, 'ba_%s.npz' % name), **out)
    print(msg)


# ---------------------------------------------------------------------------
# G7: match consolidation + initial triangulation through the
# reference's own lib/match_cleanup.py (merge_duplicates, check_for_*_dups,
# make_match_structure, link_matches, triangulate_smart)
# ---------------------------------------------------------------------------
def run_cleanup_case(name, seed, n_img, n_kp, n_tracks, gap=None):
    from lib import match_cleanup, project as ref_project, smart as ref_smart
    rng = np.random.default_rng(seed)
    tmp = '/tmp/iamx_golden_%s' % name
    os.makedirs(os.path.join(tmp, 'meta'), exist_ok=True)
    names = ['C%03d' % i for i in range(n_img)]

    class Proj(FakeProj):
        compute_kp_usage = ref_project.ProjectMgr.compute_kp_usage      # the reference's own

    proj = Proj(names, tmp)
    # keypoints: random positions; a few share the exact same pixel (different "scales")
    xy = []
    for i in range(n_img):
        p = np.stack([rng.uniform(0, W_PX - 1, n_kp), rng.uniform(0, H_PX - 1, n_kp)], 1).astype(np.float32)
        dup = rng.permutation(n_kp)[:n_kp // 10]
        p[dup] = p[(dup + 7) % n_kp]
        xy.append(p)
    # ground-truth tracks: a feature seen by a run of consecutive images, through random kps
    pair_lists = {}
    for _ in range(n_tracks):
        length = int(rng.integers(2, min(6, n_img) + 1))
        start = int(rng.integers(0, n_img - length + 1))
        if gap is not None and start < gap <= start + length - 1:
            continue                                          # two disconnected blocks of images
        kps = rng.integers(0, n_kp, length)
        for a in range(length):
            for b in range(a + 1, length):
                if b - a <= 2 and rng.random() < 0.8:
                    pair_lists.setdefault((start + a, start + b), []).append([int(kps[a]), int(kps[b])])
    for (i, j), lst in sorted(pair_lists.items()):
        order = rng.permutation(len(lst))
        lst = [lst[k] for k in order]
        if rng.random() < 0.3 and len(lst) > 3:
            lst.append(list(lst[1]))                          # an exact duplicate pair
        proj.image_list[i].match_list[names[j]] = [list(p) for p in lst]
        proj.image_list[j].match_list[names[i]] = [[p[1], p[0]] for p in lst]
    proj.image_list[0].match_list['NOT_IN_PROJECT'] = [[1, 2], [3, 4]]
    poses = []
    for i, im in enumerate(proj.image_list):
        im.kp_list = [cv2.KeyPoint(float(x), float(y), 3.0) for x, y in xy[i]]
        ned = [30.0 * (i // 4) + rng.normal(0, 0.3), 25.0 * (i % 4) + rng.normal(0, 0.3),
               -100.0 + rng.normal(0, 0.5)]
        ypr = [rng.normal(0, 20.0) + (180.0 if (i // 4) % 2 else 0.0), -90.0 + rng.normal(0, 3.0),
               rng.normal(0, 3.0)]
        im.set_camera_pose(ned, *ypr)
        poses.append(dict(ned=ned, ypr=ypr))
    camera.set_K(*[K_FC6310S[k] for k in (0, 4, 2, 5)])
    inputs = dict(names=names, xy=[p.copy() for p in xy], poses=poses,
                  match_lists=[{k: [list(p) for p in v] for k, v in im.match_list.items()}
                               for im in proj.image_list],
                  K=K_FC6310S, width=W_PX, height=H_PX)
    with quiet():
        match_cleanup.merge_duplicates(proj)
        match_cleanup.check_for_pair_dups(proj)
        match_cleanup.check_for_1vn_dups(proj)
    after = [{k: [list(map(int, p)) for p in v] for k, v in im.match_list.items()}
             for im in proj.image_list]
    kp_used = [im.kp_used.copy() for im in proj.image_list]
    with quiet():
        direct = match_cleanup.make_match_structure(proj)
        direct_copy = pickle.loads(pickle.dumps(direct))
        grouped = match_cleanup.link_matches(proj, direct)
    # triangulate_smart: per-image surface estimate from the smart node (no SRTM tiles here)
    base = {}
    for i, im in enumerate(proj.image_list):
        base[im.name] = float(rng.uniform(-3.0, 12.0))
        ref_smart.smart_node.getChild(im.name, True).setFloat('tri_surface_m', base[im.name])
    ref_smart.load = lambda path: None                        # keep the values set above
    tri = pickle.loads(pickle.dumps(grouped))
    with quiet():
        match_cleanup.triangulate_smart(proj, tri)
    # lib/groups.py:25-133 compute(): connected image groups + per-feature group level
    from lib import groups as ref_groups
    grp_in = pickle.loads(pickle.dumps(tri))
    group_out = {}
    for mcl in (0, 2):
        getNode('/config/matcher', True).setInt('min_chain_len', mcl)
        work = pickle.loads(pickle.dumps(grp_in))
        with quiet():
            gl = ref_groups.compute(proj.image_list, work)
        group_out[mcl] = dict(groups=gl, levels=[m[1] for m in work])
    getNode('/config/matcher', True).setInt('min_chain_len', 0)
    with open(os.path.join(GOLD, 'cleanup_%s.pkl' % name), 'wb') as f:
        pickle.dump(dict(inputs=inputs, match_lists_after=after, kp_used=kp_used,
                         matches_direct=direct_copy, matches_grouped=grouped, base_elev=base,
                         matches_triangulated=tri, groups=group_out), f, protocol=4)
    print('cleanup_%s: %d images, %d direct pairs -> %d chains (longest %d)'
          % (name, n_img, len(direct_copy), len(grouped), len(grouped[0]) - 2))


# ---------------------------------------------------------------------------
# G8: per-pair surface estimate through the reference's own lib/smart.py
# (triangulate_features -> estimate_surface_elevation -> update_surface_estimate) with the
# cv2 stand-in's triangulatePoints (published DLT); pins layout, conventions, bookkeeping.
# ---------------------------------------------------------------------------
def run_smart_case(name, seed, n_img, n_kp):
    from lib import smart as ref_smart
    rng = np.random.default_rng(seed)
    tmp = '/tmp/iamx_golden_%s' % name
    os.makedirs(os.path.join(tmp, 'meta'), exist_ok=True)
    names = ['M%03d' % i for i in range(n_img)]
    proj = FakeProj(names, tmp)
    camera.set_K(*[K_FC6310S[k] for k in (0, 4, 2, 5)])
    K = np.array(K_FC6310S).reshape(3, 3)
    poses, xy = [], []
    ground = rng.uniform(-5.0, 20.0)                          # true surface elevation (m, up)
    pts = np.stack([rng.uniform(-40, 140, 4 * n_kp), rng.uniform(-60, 160, 4 * n_kp),
                    -ground + rng.normal(0, 1.5, 4 * n_kp)], 1)          # NED
    for i, im in enumerate(proj.image_list):
        ned = [30.0 * (i // 3) + rng.normal(0, 0.5), 35.0 * (i % 3) + rng.normal(0, 0.5),
               -110.0 + rng.normal(0, 1.0)]
        ypr = [rng.normal(0, 15.0), -90.0 + rng.normal(0, 2.0), rng.normal(0, 2.0)]
        im.set_camera_pose(ned, *ypr)
        # the aircraft's own yaw estimate is off by a few degrees: what the yaw-error estimate
        # (smart.py:138-192) is there to find
        air_yaw = ypr[0] + rng.normal(0, 4.0)
        im.set_aircraft_pose(45.0, -93.0, 300.0, air_yaw, 0.0, 0.0)
        poses.append(dict(ned=ned, ypr=ypr, air_yaw=air_yaw))
        rvec, tvec = im.get_proj()
        uvp, _ = cv2.projectPoints(pts.reshape(-1, 1, 3), rvec, np.asarray(tvec).reshape(3, 1), K,
                                   np.zeros(5))
        uvp = uvp.reshape(-1, 2)
        vis = np.nonzero((uvp[:, 0] > 0) & (uvp[:, 0] < W_PX) & (uvp[:, 1] > 0) & (uvp[:, 1] < H_PX))[0]
        vis = vis[:n_kp]
        im._pt_ids = vis
        p = (uvp[vis] + rng.normal(0, 0.4, (len(vis), 2))).astype(np.float32)
        im.kp_list = [cv2.KeyPoint(float(x), float(y), 3.0) for x, y in p]
        xy.append(p)
    out_pairs = []
    for i in range(n_img):
        for j in range(i + 1, n_img):
            a, b = proj.image_list[i], proj.image_list[j]
            common, ia, ib = np.intersect1d(a._pt_ids, b._pt_ids, return_indices=True)
            if len(common) < 8:
                continue
            lst = [[int(x), int(y)] for x, y in zip(ia, ib)]
            if (i + j) % 3 == 0:                              # a bad pair: scrambled partners
                perm = rng.permutation(len(lst))
                lst = [[lst[k][0], lst[perm[k]][1]] for k in range(len(lst))]
            a.match_list[b.name] = lst
            b.match_list[a.name] = [[q, p_] for p_, q in lst]
            with quiet():
                avg, std = ref_smart.update_surface_estimate(a, b)
                yaw_ab = ref_smart.update_yaw_error_estimate(a, b)
                yaw_ba = ref_smart.update_yaw_error_estimate(b, a)
                aff_ab = ref_smart.find_affine(a, b)
                aff_ba = ref_smart.find_affine(b, a)
            out_pairs.append(dict(i=i, j=j, matches=lst, avg=float(avg), std=float(std),
                                  yaw_ab=float(yaw_ab), yaw_ba=float(yaw_ba),
                                  affine_ab=np.asarray(aff_ab, np.float64),
                                  affine_ba=np.asarray(aff_ba, np.float64)))
    tri = {im.name: (ref_smart.smart_node.getChild(im.name, True).getFloat('tri_surface_m')
                     if ref_smart.smart_node.getChild(im.name, True).hasChild('tri_surface_m') else None)
           for im in proj.image_list}
    yaw = {}
    for im in proj.image_list:
        node = ref_smart.smart_node.getChild(im.name, True)
        yp = node.getChild('yaw_pairs', True)
        yaw[im.name] = dict(
            yaw_error=node.getFloat('yaw_error') if node.hasChild('yaw_error') else None,
            pairs={c: tuple(yp.getChild(c).getFloat(k) for k in
                            ('yaw_error', 'dist_m', 'relative_crs', 'weight'))
                   for c in yp.getChildren()})
    with open(os.path.join(GOLD, 'smart_%s.pkl' % name), 'wb') as f:
        pickle.dump(dict(names=names, poses=poses, xy=xy, K=K_FC6310S, pairs=out_pairs,
                         tri_surface_m=tri, ground=float(ground), yaw=yaw), f, protocol=4)
    print('smart_%s: %d pairs, surface truth %.1f m, estimates %s'
          % (name, len(out_pairs), ground, sorted(set(v for v in tri.values() if v is not None))))


# ---------------------------------------------------------------------------
# G9: the reference's OWN pair loop -- lib/matcher.py:852-1031 find_matches(strategy="traditional")
# -- on a two-row strip with pre-loaded features: per-pair match lists, the surface / yaw
# bookkeeping of lib/smart.py, the yaw-error FEEDBACK (after every pair the images' camera poses
# are rewritten, lib/image.py:434-457, and the next pair triangulates with them), the
# "std >= 50 and < 100 matches -> discard" rule (:1001-1005), a quiet pair resetting the estimate
# to 0, and a second call on the same project (done pairs skipped, empty ones retried).
# ---------------------------------------------------------------------------
def _tree(node):
    out = {}
    for k, v in node.__dict__.items():
        out[k] = _tree(v) if hasattr(v, 'getChild') else (list(v) if isinstance(v, list) else v)
    return out


def _pose_record(im):
    ac = im.node.getChild('aircraft_pose', True)
    cp = im.node.getChild('camera_pose', True)
    return dict(yaw_error_deg=ac.getFloat('yaw_error_deg') if ac.hasChild('yaw_error_deg') else None,
                aircraft_quat=[ac.getFloatEnum('quat', k) for k in range(4)],
                camera_ypr=[cp.getFloat('yaw_deg'), cp.getFloat('pitch_deg'), cp.getFloat('roll_deg')],
                camera_quat=[cp.getFloatEnum('quat', k) for k in range(4)])


def run_find_matches_case(name, seed, n_img=14, n_kp=420, sorts=(True, False)):
    from lib import smart as ref_smart
    rng = np.random.default_rng(seed)
    per_row = n_img // 2
    names = ['F%03d' % i for i in range(n_img)]
    camera.set_K(*[K_FC6310S[k] for k in (0, 4, 2, 5)])
    camera.set_image_params(W_PX, H_PX)
    camera.set_mount_params(0.0, -90.0, 0.0)
    K = np.array(K_FC6310S).reshape(3, 3)
    getNode('/config/detector', True).setString('detector', 'SIFT')
    getNode('/config/detector', True).setFloat('scale', 0.4)
    mnode = getNode('/config/matcher', True)
    mnode.setFloat('match_ratio', 0.75)
    mnode.setInt('min_pairs', 25)
    matcher.configure()
    # ---- the scene: ground points with a descriptor each; two flight lines, serpentine
    ground = float(rng.uniform(2.0, 15.0))
    n_pts = 2600
    pts = np.stack([rng.uniform(-90, 30.0 * per_row + 60, n_pts), rng.uniform(-120, 160, n_pts),
                    -ground + rng.normal(0, 1.2, n_pts)], 1)
    base_des = sift_like(rng, n_pts)
    truth, reported = [], []
    for i in range(n_img):
        row, col = divmod(i, per_row)
        north = 30.0 * (col if row == 0 else per_row - 1 - col)
        ned = [north + rng.normal(0, 0.4), 38.0 * row + rng.normal(0, 0.4), -112.0 + rng.normal(0, 0.8)]
        heading = (0.0 if row == 0 else 180.0) + rng.normal(0, 3.0)
        pitch, roll = rng.normal(0, 1.5), rng.normal(0, 1.5)
        yaw_bias = rng.normal(0, 4.0) + (6.0 if i % 5 == 2 else 0.0)     # what the EKF got wrong
        truth.append(dict(ned=ned, ypr=[heading, pitch, roll]))
        reported.append(dict(ned=ned, ypr=[heading + yaw_bias, pitch, roll]))

    def fresh_project(tag):
        """images with the REPORTED aircraft pose, camera pose = aircraft pose + mount
        (lib/pose.py:125-152), keypoints = projections under the TRUE pose"""
        for n_ in list(getNode('/images', True).__dict__):
            del getNode('/images', True).__dict__[n_]
        ref_smart.smart_node.__dict__.clear()
        tmp = '/tmp/iamx_golden_fm_%s_%s' % (name, tag)
        os.makedirs(os.path.join(tmp, 'meta'), exist_ok=True)
        for f_ in os.listdir(os.path.join(tmp, 'meta')):
            os.remove(os.path.join(tmp, 'meta', f_))
        proj = FakeProj(names, tmp)
        body2cam = camera.get_body2cam()
        r2d = 180.0 / math.pi
        for i, im in enumerate(proj.image_list):
            for kind in ('truth', 'reported'):
                src = truth[i] if kind == 'truth' else reported[i]
                im.set_aircraft_pose(45.0, -93.0, 300.0, *src['ypr'])
                ned2body = [im.node.getChild('aircraft_pose').getFloatEnum('quat', k) for k in range(4)]
                ned2cam = _tf.quaternion_multiply(ned2body, body2cam)
                y, p, r = _tf.euler_from_quaternion(ned2cam, 'rzyx')
                im.set_camera_pose(src['ned'], y * r2d, p * r2d, r * r2d)
                if kind == 'truth':
                    rvec, tvec = im.get_proj()
                    uvp, _ = cv2.projectPoints(pts.reshape(-1, 1, 3), rvec,
                                               np.asarray(tvec).reshape(3, 1), K, np.zeros(5))
                    im._uvp = uvp.reshape(-1, 2)
        return proj

    proj = fresh_project('probe')
    rng_kp = np.random.default_rng(seed + 1)
    des, xy = [], []
    for i, im in enumerate(proj.image_list):
        uvp = im._uvp
        vis = np.nonzero((uvp[:, 0] > 2) & (uvp[:, 0] < W_PX - 2) & (uvp[:, 1] > 2) & (uvp[:, 1] < H_PX - 2))[0]
        vis = rng_kp.permutation(vis)[:n_kp]
        p = uvp[vis] + rng_kp.normal(0, 0.4, (len(vis), 2))
        d = np.clip(base_des[vis].astype(np.int64) + rng_kp.integers(-5, 6, (len(vis), 128)), 0, 255)
        # clutter: features of nothing on the ground
        n_cl = 60
        p = np.concatenate([p, np.stack([rng_kp.uniform(0, W_PX - 1, n_cl), rng_kp.uniform(0, H_PX - 1, n_cl)], 1)])
        d = np.concatenate([d, sift_like(rng_kp, n_cl)])
        des.append(d.astype(np.uint8))
        xy.append(np.clip(p, 0, [W_PX - 1, H_PX - 1]).astype(np.float32))
    # a false match block between images 1 and 5 (4 apart: no common ground): the same 70 "features"
    # at nearly the same pixels of both -> GMS-consistent, triangulates to nonsense (std >= 50)
    a_, b_ = 1, 5
    n_f = 70
    blk = sift_like(rng_kp, n_f)
    pos = np.stack([rng_kp.uniform(900, 2400, n_f), rng_kp.uniform(700, 1900, n_f)], 1)
    for k_, shift in ((a_, (0.0, 0.0)), (b_, (35.0, -20.0))):
        dd = np.clip(blk.astype(np.int64) + rng_kp.integers(-4, 5, (n_f, 128)), 0, 255).astype(np.uint8)
        pp = (pos + shift + rng_kp.normal(0, 2.5, (n_f, 2))).astype(np.float32)
        des[k_] = np.concatenate([des[k_], dd])
        xy[k_] = np.concatenate([xy[k_], pp])

    runs = {}
    for sort in sorts:
        proj = fresh_project('sort%d' % int(sort))
        for i, im in enumerate(proj.image_list):
            im.des_list = des[i].astype(np.float32)
            im.kp_list = [cv2.KeyPoint(float(x), float(y), 3.0) for x, y in xy[i]]
        initial = [_pose_record(im) for im in proj.image_list]
        calls = []
        for call in range(2):
            with quiet():
                matcher.find_matches(proj, K, strategy='traditional', transform='gms', sort=sort)
            calls.append(dict(
                match_lists=[{k: [list(map(int, p_)) for p_ in v] for k, v in im.match_list.items()}
                             for im in proj.image_list],
                smart=pickle.loads(pickle.dumps(_tree(ref_smart.smart_node))),
                poses=[_pose_record(im) for im in proj.image_list]))
        runs[bool(sort)] = dict(initial=initial, calls=calls)
        ml = calls[0]['match_lists']
        n_hit = sum(1 for m in ml for v in m.values() if len(v)) // 2
        n_all = sum(len(m) for m in ml) // 2
        disc = [(names[i], o) for i, m in enumerate(ml) for o, v in m.items()
                if not v and o in calls[0]['smart'].get(names[i], {}).get('tri_surface_pairs', {})]
        print('find_matches_%s sort=%s: %d pairs, %d with matches, discarded %s, yaw_error_deg %s'
              % (name, sort, n_all, n_hit, sorted(disc)[:4],
                 ['%.2f' % (p_['yaw_error_deg'] or 0) for p_ in calls[0]['poses']]))
    with open(os.path.join(GOLD, 'find_matches_%s.pkl' % name), 'wb') as f:
        pickle.dump(dict(names=names, des=des, xy=xy, K=K_FC6310S, width=W_PX, height=H_PX,
                         mount=[0.0, -90.0, 0.0], reported=reported, truth=truth,
                         aircraft_lla=[45.0, -93.0, 300.0], match_ratio=0.75, min_pairs=25,
                         ground=ground, runs=runs), f, protocol=4)


def main():
    os.makedirs(GOLD, exist_ok=True)
    # G1 ------------------------------------------------------------------
    run_match_case('basic', seed=11, n1=768, n2=640, n_true=300)
    run_match_case('dups', seed=12, n1=900, n2=900, n_true=400, dup_uv=60, tie_rows=40)
    run_match_case('fewpairs', seed=13, n1=300, n2=280, n_true=12)
    run_match_case('clip2000', seed=14, n1=2600, n2=2500, n_true=2300, noise=3)
    run_match_case('ratio06', seed=15, n1=512, n2=512, n_true=200, match_ratio=0.6,
                   min_pairs=10, noise=12)
    # G2-G6 ---------------------------------------------------------------
    run_ba_case('nodist', seed=21, rows=2, cols=3, n_pts=60, dist=(0, 0, 0, 0, 0))
    run_ba_case('dist', seed=22, rows=3, cols=4, n_pts=160,
                dist=(-0.12, 0.083, -0.0016, -0.00096, -0.012))
    run_ba_case('calib', seed=23, rows=3, cols=3, n_pts=120,
                dist=(-0.05, 0.02, 0.001, -0.0005, 0.0), cam_calib=True)
    run_ba_case('mid', seed=24, rows=5, cols=6, n_pts=700, dist=(0, 0, 0, 0, 0))
    # G7 ------------------------------------------------------------------
    run_cleanup_case('small', seed=31, n_img=6, n_kp=120, n_tracks=150)
    run_cleanup_case('strip', seed=32, n_img=16, n_kp=600, n_tracks=1500)
    run_cleanup_case('twoblocks', seed=33, n_img=22, n_kp=1500, n_tracks=2600, gap=12)
    # G8 ------------------------------------------------------------------
    run_smart_case('grid', seed=41, n_img=6, n_kp=400)
    # G9 ------------------------------------------------------------------
    run_find_matches_case('strip', seed=51)


if __name__ == '__main__':
    main()
