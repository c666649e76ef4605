"""GPU: the whole drop-in chain on rendered imagery -- what scripts/process.py does from step 3
to step 4 with the MI355X modules in place of lib.{image,matcher,match_cleanup,groups,optimizer}:

    JPEGs -> Image.detect_features -> matcher.find_matches -> merge_duplicates / check_* ->
    make_match_structure -> link_matches -> triangulate_smart -> groups.compute ->
    Optimizer.setup / run -> update_camera_poses

Scene: a textured ground plane photographed by 8 nadir cameras on a 2 x 4 lawn-mower grid
(opposite headings per row, a few degrees of tilt), rendered by casting every pixel's ray onto
the plane with the same camera model the pipeline uses.  The project is handed poses that are
off by ~1 m / ~1 deg; after bundle adjustment the reprojection error must be sub-pixel and the
camera positions close to the truth."""
import os
import pickle

import numpy as np
import pytest

from test_sift_gpu import texture

pytestmark = pytest.mark.gpu

W, H, F = 800, 600, 600.0
ALT = 100.0
GSD = 0.125                        # metres per texel of the ground texture


def _render(tex, M, ned):
    """image[v, u] = ground texture under the ray of pixel (u, v); ground plane z = 0 (NED)."""
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    ray = np.einsum('ij,jvu->ivu', M, np.stack([u, v, np.ones_like(u)]))
    t = -ned[2] / ray[2]
    n, e = ned[0] + ray[0] * t, ned[1] + ray[1] * t
    r, c = (n + 60.0) / GSD, (e + 60.0) / GSD
    r0, c0 = np.floor(r).astype(int), np.floor(c).astype(int)
    fr, fc = (r - r0)[..., None], (c - c0)[..., None]
    r0 = np.clip(r0, 0, tex.shape[0] - 2)
    c0 = np.clip(c0, 0, tex.shape[1] - 2)
    t00, t01 = tex[r0, c0].astype(float), tex[r0, c0 + 1].astype(float)
    t10, t11 = tex[r0 + 1, c0].astype(float), tex[r0 + 1, c0 + 1].astype(float)
    img = (t00 * (1 - fc) + t01 * fc) * (1 - fr) + (t10 * (1 - fc) + t11 * fc) * fr
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def test_detect_match_consolidate_triangulate_group_optimize(tmp_path):
    from PIL import Image as PILImage
    from imageanalysis_amd import groups, image as iimg, match_cleanup, matcher, optimizer
    from imageanalysis_amd._deps import getNode
    from imageanalysis_amd.hostlib import camera
    from imageanalysis_amd.hostlib import transforms as tf
    rng = np.random.default_rng(2024)
    proj_dir = tmp_path / 'proj'
    (proj_dir / 'images').mkdir(parents=True)
    an = proj_dir / 'ImageAnalysis'
    (an / 'cache').mkdir(parents=True)
    (an / 'meta').mkdir()
    tex = texture(int(220 / GSD), int(300 / GSD), 5)             # 220 m x 300 m of ground (BGR)
    K = np.array([[F, 0, W / 2.0], [0, F, H / 2.0], [0, 0, 1.0]])
    IK = np.linalg.inv(K)
    names, truth = [], []
    for row in range(2):
        for col in range(4):
            k = col if row == 0 else 3 - col
            ned = np.array([20.0 + 45.0 * row + rng.normal(0, 0.5), 25.0 + 40.0 * k + rng.normal(0, 0.5),
                            -ALT + rng.normal(0, 0.5)])
            ypr = np.array([(0.0 if row == 0 else 180.0) + rng.normal(0, 2.0), -90.0 + rng.normal(0, 1.5),
                            rng.normal(0, 1.5)])
            names.append('P%02d' % len(names))
            truth.append((ned, ypr))
    getNode('/config/directories', True).setString('project_dir', str(proj_dir))
    matcher.detector_node.setString('detector', 'SIFT')
    matcher.detector_node.setFloat('scale', 1.0)
    matcher.matcher_node.setFloat('match_ratio', 0.75)
    matcher.matcher_node.setInt('min_pairs', 25)
    matcher.matcher_node.setString('schedule', 'all-pairs')
    matcher.matcher_node.setInt('min_chain_len', 0)
    node = getNode('/config/camera', True)
    node.__dict__.pop('K_opt', None)
    node.__dict__.pop('dist_coeffs_opt', None)
    camera.set_K(F, F, W / 2.0, H / 2.0)
    camera.set_dist_coeffs([0.0] * 5)
    camera.set_image_params(W, H)
    camera.set_mount_params(0.0, -90.0, 0.0)

    class Proj(object):
        analysis_dir = str(an)

        def findIndexByName(self, name):
            return names.index(name) if name in names else None

        def findImageByName(self, name):
            return self.image_list[names.index(name)] if name in names else None

        def save_images_info(self):
            pass

    proj = Proj()
    proj.image_list = []
    d2r = np.pi / 180.0
    for name, (ned, ypr) in zip(names, truth):
        q = tf.quaternion_from_euler(ypr[0] * d2r, ypr[1] * d2r, ypr[2] * d2r, 'rzyx')
        M = tf.quaternion_matrix(q)[:3, :3].dot(match_cleanup.CAM2BODY).dot(IK)
        bgr = _render(tex, M, ned)
        PILImage.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(
            str(proj_dir / 'images' / (name + '.JPG')), quality=95)
        im = iimg.Image(str(an), name)
        # what the flight log would say: off by ~1 m and ~1 deg
        im.set_pose_from_camera((ned + rng.normal(0, 0.8, 3)).tolist(), *(ypr + rng.normal(0, 0.7, 3)).tolist())
        getNode('/smart', True).getChild(name, True).setFloat('tri_surface_m', 0.0)
        proj.image_list.append(im)

    try:
        matcher.configure()
        matcher.find_matches(proj, K, strategy='traditional', transform='homography', sort=True)
    finally:
        matcher.matcher_node.__dict__.pop('schedule', None)
    n_pairs = sum(len(v) > 0 for im in proj.image_list for v in im.match_list.values()) // 2
    assert n_pairs >= 12                                   # neighbours overlap, far pairs do not
    assert all(len(im.kp_list) > 1000 for im in proj.image_list)
    assert os.path.exists(os.path.join(str(an), 'meta', 'P00.match'))
    # the surface statistics find_matches recorded: every pair was triangulated with the camera
    # poses the reference's loop holds when it reaches the pair (tests/test_find_matches_loop.py
    # pins that against the reference's own loop) -- on these frames, rendered from the plane
    # z = 0 with ~1 m / ~1 deg of pose error, that is a surface within a few metres of 0 for every pair
    # (40 m baselines seen from 100 m: a degree of attitude error moves a pair's surface by metres)
    from imageanalysis_amd import smart
    n_checked = 0
    for a in proj.image_list:
        for b in proj.image_list:
            if a is not b and len(a.match_list.get(b.name, [])):
                rec = smart.smart_node.getChild(a.name).getChild('tri_surface_pairs').getChild(b.name)
                assert abs(rec.getFloat('surface_m')) < 10.0 and rec.getFloat('stddev') < 8.0
                n_checked += 1
    assert n_checked >= 24

    match_cleanup.merge_duplicates(proj)
    match_cleanup.check_for_pair_dups(proj)
    match_cleanup.check_for_1vn_dups(proj)
    direct = match_cleanup.make_match_structure(proj)
    grouped = match_cleanup.link_matches(proj, direct)
    assert len(grouped) > 500 and len(grouped[0]) - 2 >= 4  # chains through >= 4 images exist
    match_cleanup.triangulate_smart(proj, grouped)
    # the triangulated features lie on the per-image surface estimates find_matches accumulated
    # (smart: pairwise DLT triangulation with the ~1 m / ~1 deg pose errors => within ~2.5 m of
    # the plane the images were rendered from)
    pts = np.array([m[0] for m in grouped])
    assert np.abs(pts[:, 2]).max() < 2.5 and pts[:, 0].min() > -60 and pts[:, 1].max() < 240
    surf = [getNode('/smart', True).getChild(n, True).getFloat('tri_surface_m') for n in names]
    assert max(abs(v) for v in surf) < 2.5 and os.path.exists(os.path.join(str(an), 'smart.json'))
    pickle.loads(pickle.dumps(grouped))
    group_list = groups.compute(proj.image_list, grouped)
    assert len(group_list) == 1 and sorted(group_list[0]) == names

    opt = optimizer.Optimizer(str(an))
    opt.setup(proj, group_list, 0, grouped, optimized=False, cam_calib=False)
    x0 = opt._x0()
    args = (opt.n_cameras, opt.n_points, opt.by_camera_point_indices, opt.by_camera_points_2d)
    mre0 = np.mean(np.abs(opt.fun(x0, *args)))
    cameras, features, cam_index_map, feat_index_map, fx, fy, cu, cv, dist = opt.run()
    mre1 = np.mean(np.abs(opt.result.fun))
    assert mre0 > 3.0 and mre1 < 0.6, (mre0, mre1)         # pixels
    opt.update_camera_poses(proj)
    # relative geometry recovered: the camera-to-camera baselines agree with the truth up to the
    # one global scale BA cannot observe (absolute position / heading / scale of the block are
    # gauge freedoms, only bounded by the +-3 m / +-9 m box around the logged poses)
    est = np.array([proj.image_list[i].get_camera_pose(opt=True)[0] for i in range(len(names))])
    tru = np.array([t[0] for t in truth])
    db_est = np.linalg.norm(est[:, None, :] - est[None, :, :], axis=2)
    db_tru = np.linalg.norm(tru[:, None, :] - tru[None, :, :], axis=2)
    scale = (db_est * db_tru).sum() / (db_tru * db_tru).sum()
    assert abs(scale - 1.0) < 0.03
    assert np.abs(db_est - scale * db_tru).max() < 0.15, np.abs(db_est - scale * db_tru).max()
