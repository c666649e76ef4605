"""ORACLE / TEST INFRASTRUCTURE ONLY -- never imported by the product path.

One scripted pass over every HOST-ONLY entry point of the drop-in (no device work), written
against an "environment" so that it can run twice:

  * oracle/check_dropin.py runs it INSIDE the reference environment -- `props` importable
    (oracle/shims), the reference's own lib/{project,image,camera,logger}.py imported from
    /root/reference, INTEGRATION.md's shim files in place of lib/{matcher,optimizer,smart,
    match_cleanup,groups}.py, `imageanalysis_amd._deps.HAVE_PROPS = True` -- and records what comes
    out as tests/golden/dropin_env.pkl;
  * tests/test_dropin.py replays it with the package's own stand-ins (hostlib/) and compares.

Equal outputs = the two branches of imageanalysis_amd/_deps.py behave the same: the branch a real
integration takes (never executed before round 5) and the branch every other test runs.

Inputs are the committed goldens (tests/golden/smart_grid.pkl, cleanup_small.pkl, ba_dist*.pkl),
which oracle/gen_golden.py produced by running the reference's own modules."""
import gzip
import hashlib
import json
import os
import pickle

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def _sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def _tree(node):
    """a property (sub)tree as plain python, through the public accessors only"""
    out = {}
    for name in node.getChildren():
        child = node.getChild(name)
        if child is not None and not (hasattr(node, 'isLeaf') and node.isLeaf(name)):
            out[name] = _tree(child)
        else:
            n = node.getLen(name)
            if n:
                out[name] = [node.getFloatEnum(name, i) for i in range(n)]
            else:
                s = node.getString(name)
                try:
                    out[name] = float(s)
                except ValueError:
                    out[name] = s
    return out


class Env(object):
    """what the scenario needs from its surroundings (see the two constructors of the callers)"""
    def __init__(self, make_project, camera, getNode, matcher, smart, match_cleanup, groups,
                 optimizer, make_keypoints, wait_writes):
        self.make_project = make_project          # (names, directory) -> project with .image_list
        self.camera = camera
        self.getNode = getNode
        self.matcher, self.smart = matcher, smart
        self.match_cleanup, self.groups, self.optimizer = match_cleanup, groups, optimizer
        self.make_keypoints = make_keypoints      # (xy [n,2]) -> kp_list
        self.wait_writes = wait_writes            # (path) -> None: background cache writers


def _reset_images(env, names):
    images = env.getNode('/images', True)
    smart_node = env.getNode('/smart', True)
    for n in names:
        for node in (images, smart_node):
            if node.hasChild(n):
                # (a fresh child: both tree implementations keep children in __dict__)
                node.__dict__.pop(n, None)


def run(env, workdir):
    """-> dict of plain data (numbers, strings, lists, sha256 digests)"""
    out = {}
    with open(os.path.join(GOLD, 'smart_grid.pkl'), 'rb') as f:
        g = pickle.load(f)
    names = g['names']
    _reset_images(env, names)
    K = g['K']
    project_dir = os.path.join(workdir, 'project')
    os.makedirs(os.path.join(project_dir, 'ImageAnalysis', 'meta'), exist_ok=True)
    os.makedirs(os.path.join(project_dir, 'ImageAnalysis', 'cache'), exist_ok=True)
    env.getNode('/config/directories', True).setString('project_dir', project_dir)
    proj = env.make_project(names, project_dir)
    analysis_dir = proj.analysis_dir
    # (after the project: a new ProjectMgr resets /config/camera to its defaults, project.py:118)
    env.camera.set_K(K[0], K[4], K[2], K[5])
    env.camera.set_image_params(5472, 3648)
    rng = np.random.default_rng(7)
    for im, pose, xy in zip(proj.image_list, g['poses'], g['xy']):
        im.set_camera_pose(pose['ned'], *pose['ypr'])
        im.set_aircraft_pose(45.0, -93.0, 300.0, pose['air_yaw'], 0.0, 0.0)
        im.kp_list = env.make_keypoints(np.asarray(xy, np.float32))
        im.des_list = rng.integers(0, 256, (len(xy), 128)).astype(np.float32)

    # ---- 1. feature cache: save_features / save_descriptors / load_* (lib/image.py:140-217)
    feat = {}
    for im in proj.image_list:
        im.save_features()
        im.save_descriptors()
    for im in proj.image_list:
        env.wait_writes(im.features_file)
        env.wait_writes(im.desc_file)
        with gzip.open(im.features_file, 'rb') as fp:
            payload = fp.read()
        with gzip.open(im.desc_file, 'rb') as fp:
            dpayload = fp.read()
        feat[im.name] = dict(feat_sha=_sha(payload), desc_sha=_sha(dpayload), n=len(im.kp_list))
        des = im.des_list
        im.kp_list, im.des_list = [], None
        assert im.load_features() and im.load_descriptors()
        assert len(im.kp_list) == feat[im.name]['n']
        assert np.array_equal(np.asarray(im.des_list, np.float32), des)
        pts = np.array([kp.pt for kp in im.kp_list], np.float32)
        feat[im.name]['pt_sha'] = _sha(pts.tobytes())
    out['feature_cache'] = feat

    # ---- 2. match lists: saveMatches + load_matches (lib/matcher.py:1033-1041, image.py:182-228)
    from imageanalysis_amd.matchpairs import MatchPairs
    for rec in g['pairs']:
        a, b = proj.image_list[rec['i']], proj.image_list[rec['j']]
        m = np.asarray(rec['matches'], np.int32).reshape(-1, 2)
        a.match_list[b.name] = MatchPairs(m)                       # what find_matches leaves
        b.match_list[a.name] = [[int(q), int(p)] for p, q in m]    # what other code leaves
        a.matches_clean = b.matches_clean = False
    env.matcher.saveMatches(proj.image_list)
    match = {}
    for im in proj.image_list:
        with open(im.match_file, 'rb') as fp:
            raw = fp.read()
        loaded = pickle.loads(raw)
        assert all(type(v) is list and all(type(p) is list for p in v) for v in loaded.values())
        match[im.name] = dict(sha=_sha(raw), keys=sorted(loaded),
                              n=[len(loaded[k]) for k in sorted(loaded)])
        want = {k: [list(map(int, p)) for p in v] for k, v in im.match_list.items()}
        im.match_list = {}
        im.load_matches()
        assert {k: [list(p) for p in v] for k, v in im.match_list.items()} == want and im.matches_clean
    out['match_files'] = match

    # ---- 3. /smart bookkeeping as find_matches drives it (lib/matcher.py:987-993,1010,1030 ->
    #         lib/smart.py:196-283, 319-339), device outputs replaced by the golden's values
    smart = env.smart
    smart.load(analysis_dir)                       # (no file yet: clears the module's caches)
    smart.freeze_poses(True)
    smart.begin_batch()
    rounds = []
    for rec in g['pairs']:
        a, b = proj.image_list[rec['i']], proj.image_list[rec['j']]
        na, nb = np.array(a.get_camera_pose()[0]), np.array(b.get_camera_pose()[0])
        dist_m = float(np.linalg.norm(nb - na))
        yaws = []
        for (p, q, aff) in ((a, b, rec['affine_ab']), (b, a, rec['affine_ba'])):
            np_, nq_ = np.array(p.get_camera_pose()[0]), np.array(q.get_camera_pose()[0])
            yv = smart.yaw_errors_from_affines(np_[None], np.array([p.get_aircraft_pose()[1][0]]),
                                               nq_[None], np.asarray(aff, float).reshape(1, 6))
            yaws.append(tuple(float(v[0]) for v in yv))
        rounds.append((a, b, rec['avg'], rec['std'], dist_m, yaws[0], yaws[1]))
    half = len(rounds) // 2
    smart.record_round(rounds[:half])
    yaw_avg = dict(smart.flush_aggregates())
    smart.record_round(rounds[half:])
    yaw_avg.update(smart.flush_aggregates())
    smart.freeze_poses(False)
    out['yaw_averages'] = {k: round(float(v), 9) for k, v in sorted(yaw_avg.items())}
    # the golden's own tree (made by the reference's lib/smart.py, pair by pair) is the yardstick
    for im in proj.image_list:
        node = smart.smart_node.getChild(im.name, True)
        want = g['tri_surface_m'][im.name]
        got = node.getFloat('tri_surface_m') if node.hasChild('tri_surface_m') else None
        assert got == want, (im.name, got, want)
        wy = g['yaw'][im.name]
        got_y = node.getFloat('yaw_error') if node.hasChild('yaw_error') else None
        assert got_y == wy['yaw_error'], (im.name, got_y, wy['yaw_error'])
        yp = node.getChild('yaw_pairs', True)
        assert sorted(yp.getChildren()) == sorted(wy['pairs'])
        for c, vals in wy['pairs'].items():
            got = tuple(yp.getChild(c).getFloat(k) for k in ('yaw_error', 'dist_m', 'relative_crs', 'weight'))
            assert got == tuple(vals), (im.name, c, got, vals)
    assert smart.get_surface_estimate(proj.image_list[0], proj.image_list[1]) == \
        (g['tri_surface_m'][names[0]] + g['tri_surface_m'][names[1]]) / 2
    smart.update_srtm_elevations(proj, ned_interp=lambda p: [250.0 + 0.01 * p[0] - 0.02 * p[1]])
    smart.save(analysis_dir)
    with open(os.path.join(analysis_dir, 'smart.json')) as f:
        on_disk = json.load(f)
    out['smart_json'] = {k: on_disk[k] for k in names}
    # a reader that starts from the file (process.py:239-240)
    for n in names:
        smart.smart_node.__dict__.pop(n, None)
    smart.load(analysis_dir)
    out['smart_tree_reloaded'] = {k: _tree(smart.smart_node.getChild(k, True)) for k in names}
    smart.set_yaw_error_estimates(proj)
    out['yaw_error_deg'] = {im.name: im.node.getChild('aircraft_pose', True).getFloat('yaw_error_deg')
                            for im in proj.image_list}
    out['yaw_error_estimate'] = {im.name: smart.get_yaw_error_estimate(im) for im in proj.image_list}

    # ---- 4. consolidation + grouping (lib/match_cleanup.py:14-301, lib/groups.py:25-133)
    with open(os.path.join(GOLD, 'cleanup_small.pkl'), 'rb') as f:
        c = pickle.load(f)
    inp = c['inputs']
    _reset_images(env, inp['names'])
    cproj = env.make_project(inp['names'], os.path.join(workdir, 'cleanup'))
    for im, xy, pose, ml in zip(cproj.image_list, inp['xy'], inp['poses'], inp['match_lists']):
        im.kp_list = env.make_keypoints(np.asarray(xy, np.float32))
        im.set_camera_pose(pose['ned'], *pose['ypr'])
        im.match_list = {k: [list(p) for p in v] for k, v in ml.items()}
    mc = env.match_cleanup
    mc.merge_duplicates(cproj)
    mc.check_for_pair_dups(cproj)
    mc.check_for_1vn_dups(cproj)
    after = [{k: [list(map(int, p)) for p in v] for k, v in im.match_list.items()}
             for im in cproj.image_list]
    assert after == c['match_lists_after']
    direct = mc.make_match_structure(cproj)
    assert pickle.loads(pickle.dumps(direct)) == c['matches_direct']
    grouped = mc.link_matches(cproj, direct)
    grouped_l = pickle.loads(pickle.dumps(grouped))
    assert grouped_l == c['matches_grouped']
    tri = pickle.loads(pickle.dumps(c['matches_triangulated']))
    env.getNode('/config/matcher', True).setInt('min_chain_len', 0)
    gl = env.groups.compute(cproj.image_list, tri)
    assert gl == c['groups'][0]['groups'] and [m[1] for m in tri] == c['groups'][0]['levels']
    env.groups.save(cproj.analysis_dir, gl)
    assert env.groups.load(cproj.analysis_dir) == gl
    out['cleanup'] = dict(n_direct=len(c['matches_direct']), n_chains=len(grouped_l),
                          groups=[len(x) for x in gl], chains_sha=_sha(pickle.dumps(grouped_l, 2)))

    # ---- 5. optimizer: setup -> pose write-back -> refit (lib/optimizer.py:283-405,543-683)
    with open(os.path.join(GOLD, 'ba_dist_in.pkl'), 'rb') as f:
        b = pickle.load(f)
    gz = np.load(os.path.join(GOLD, 'ba_dist.npz'))
    with open(os.path.join(GOLD, 'ba_dist_refit.pkl'), 'rb') as f:
        want = pickle.load(f)
    _reset_images(env, b['names'])
    bproj = env.make_project(b['names'], os.path.join(workdir, 'ba'))
    for im, (ned, ypr, _quat) in zip(bproj.image_list, b['poses']):
        im.set_camera_pose(ned, ypr[0], ypr[1], ypr[2])
    cam_node = env.getNode('/config/camera', True)
    for key in ('K_opt', 'dist_coeffs_opt'):
        cam_node.__dict__.pop(key, None)
    env.camera.set_K(b['K'][0], b['K'][4], b['K'][2], b['K'][5])
    env.camera.set_dist_coeffs(b['dist'])
    env.camera.set_image_params(b['width'], b['height'])
    opt = env.optimizer.Optimizer(bproj.analysis_dir)
    matches = pickle.loads(pickle.dumps(b['matches']))
    opt.setup(bproj, b['groups'], 0, matches, optimized=False, cam_calib=bool(gz['cam_calib']))
    assert np.array_equal(opt.camera_indices, gz['camera_indices'])
    assert np.array_equal(opt.point_indices, gz['point_indices'])
    C, P = opt.n_cameras, opt.n_points
    xf = np.asarray(gz['x_final'], float)
    opt.camera_params = xf[:C * 7].reshape(C, 7)
    opt.points_3d = xf[C * 7:C * 7 + P * 3].reshape(P, 3)
    opt.update_camera_poses(bproj)
    for im, (ned, ypr, quat), valid in zip(bproj.image_list, want['poses_opt'], want['valid']):
        assert bool(im.node.getChild('camera_pose_opt', True).getBool('valid')) == valid
        if valid:
            n2, y2, q2 = im.get_camera_pose(opt=True)
            assert np.allclose(n2, ned, atol=1e-9) and np.allclose(y2, ypr, atol=1e-9)
            assert np.allclose(q2, quat, atol=1e-12)
    opt.refit(bproj, matches, b['groups'], 0)
    poses = {}
    for im, (ned, ypr, quat), valid in zip(bproj.image_list, want['poses_refit'], want['valid']):
        if valid:
            n2, y2, q2 = im.get_camera_pose(opt=True)
            assert np.allclose(n2, ned, atol=1e-8) and np.allclose(y2, ypr, atol=1e-8)
            assert np.allclose(q2, quat, atol=1e-10)
            poses[im.name] = [round(float(v), 7) for v in list(n2) + list(y2) + list(q2)]
    for m, wpt in zip(matches, want['matches_points']):
        assert np.allclose(m[0], wpt, atol=1e-8)
    out['ba_poses_refit'] = poses
    # the pose JSON a later stage reads (lib/project.py:199-210 save_images_info)
    if hasattr(bproj, 'save'):
        bproj.save()                              # config.json (lib/project.py:84-92)
    if hasattr(bproj, 'save_images_info'):
        bproj.save_images_info()
    return out
