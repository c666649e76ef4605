#!/usr/bin/env python3
"""ORACLE / TEST INFRASTRUCTURE ONLY -- build container only (needs /root/reference), never shipped,
never imported by the product path.

Executes the branch a REAL integration takes, which no test could run before round 5
(VERDICT round 4, weak #8): `props` importable (oracle/shims) so that
imageanalysis_amd._deps.HAVE_PROPS is True, the reference's OWN
scripts/lib/{project,image,camera,logger,state,...}.py imported from /root/reference, and the shim
files INTEGRATION.md section 2 tells a maintainer to add -- cut out of INTEGRATION.md itself, so
the document is what is tested -- standing in for lib/{matcher,optimizer,smart,match_cleanup,
groups}.py and appended to lib/image.py.  Nothing of the reference is copied: the overlay package
is written to a temporary directory at run time, its `lib/__init__.py` extends the package path to
/root/reference/scripts/lib, and its image.py exec()s the reference file where it lies.

What it checks (each line of the output is one check; any failure raises):
  1. bindings      _deps binds props.getNode, the reference's lib.camera / lib.logger -- and the
                   mirror's smart (NOT lib/smart.py: the batched device triangulation / similarity
                   kernels would be bypassed); `lib.smart` etc. resolve to the shim files
  2. API surface   every top-level function of the reference's lib/smart.py, match_cleanup.py,
                   groups.py exists in the mirror; the live entry points of matcher / optimizer
  3. files         cache and .match files written by the installed methods are read by the
                   reference's ORIGINAL loaders (and the other way round)
  4. scenario      oracle/dropin_scenario.run(): feature cache, saveMatches, /smart bookkeeping on
                   the real tree + props_json, srtm / yaw hooks of process.py:218-240,
                   consolidation + groups, Optimizer.setup / update_camera_poses / refit on the
                   reference's Image objects and ProjectMgr -- against the goldens
  6. find_matches  the pair loop in the reference environment (real ProjectMgr, Image, props
                   tree, lib.camera; the two device steps replaced by the oracle stand-ins of
                   tests/test_find_matches_loop.py) == the reference's ORIGINAL find_matches (G9)
  5. reference smart   the reference's ORIGINAL lib/smart.py (imported under another name) gives
                   the same update_srtm_elevations / set_yaw_error_estimates results, and its
                   triangulate_features() output is recorded for the GPU test of the mirror's

Writes tests/golden/dropin_env.pkl (data only) for tests/test_dropin.py to replay with the
package's stand-ins.

    python oracle/check_dropin.py
"""
import ast
import contextlib
import importlib
import importlib.util
import io
import os
import pickle
import re
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = '/root/reference/scripts'
GOLD = os.path.join(REPO, 'tests', 'golden')


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def shim_files():
    """{file name: text} of the blocks of INTEGRATION.md that start with `# scripts/lib/<name>.py`"""
    text = open(os.path.join(REPO, 'INTEGRATION.md')).read()
    out = {}
    for block in re.findall(r'```python\n(.*?)```', text, re.S):
        cur = None
        for line in block.splitlines(True):
            m = re.match(r'# scripts/lib/(\w+)\.py', line)
            if m:
                cur = m.group(1)
                out.setdefault(cur, '')
            if cur is not None:
                out[cur] += line
    return out


def build_overlay(root):
    lib = os.path.join(root, 'lib')
    os.makedirs(lib)
    with open(os.path.join(lib, '__init__.py'), 'w') as f:
        f.write("__path__.append(%r)\n" % os.path.join(REF, 'lib'))
    shims = shim_files()
    need = {'matcher', 'optimizer', 'smart', 'match_cleanup', 'groups', 'image'}
    assert need <= set(shims), "INTEGRATION.md lacks the shim file of: %s" % sorted(need - set(shims))
    for name in need - {'image'}:
        with open(os.path.join(lib, name + '.py'), 'w') as f:
            f.write(shims[name])
    with open(os.path.join(lib, 'image.py'), 'w') as f:
        # "scripts/lib/image.py, last lines": the reference's file, then the block
        f.write("_src = %r\nexec(compile(open(_src).read(), _src, 'exec'))\n" % os.path.join(REF, 'lib', 'image.py'))
        f.write(shims['image'])
    return shims


def top_level_defs(path):
    tree = ast.parse(open(path).read())
    return [n.name for n in tree.body if isinstance(n, ast.FunctionDef)]


def load_original(name):
    """the reference's lib/<name>.py as module lib._orig_<name> (relative imports resolve in lib)"""
    spec = importlib.util.spec_from_file_location('lib._orig_' + name, os.path.join(REF, 'lib', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    overlay = tempfile.mkdtemp(prefix='iamx_dropin_')
    work = tempfile.mkdtemp(prefix='iamx_dropin_work_')
    try:
        build_overlay(overlay)
        sys.path[:0] = [os.path.join(HERE, 'shims'), overlay, REF, REPO]
        sys.path.append(os.path.join(REF, 'lib', 'archive'))          # -> transformations
        import numpy as np
        import transformations as _tf

        class _Numpy1Compat(object):
            """the archived transformations.py means numpy 1.x's array(copy=False)"""
            def __getattr__(self, k):
                return getattr(np, k)

            @staticmethod
            def array(obj, *a, **k):
                if k.get('copy', True) is False:
                    k.pop('copy')
                    return np.asarray(obj, *a, **k)
                return np.array(obj, *a, **k)
        _tf.numpy = _Numpy1Compat()

        import props
        import props_json   # noqa: F401
        import cv2          # the shim
        with quiet():
            from lib import camera, logger, project, image as lib_image      # reference modules
            from lib import matcher, optimizer, smart, match_cleanup, groups  # shim files
        import imageanalysis_amd
        from imageanalysis_amd import _deps, cacheio, keypoints
        ok = lambda msg: print('ok   ' + msg)

        # ---- 1. bindings ---------------------------------------------------------------
        assert _deps.HAVE_PROPS and _deps.getNode is props.getNode
        assert _deps.camera() is camera and camera.__file__.startswith(REF)
        assert _deps.logger() is logger and logger.__file__.startswith(REF)
        assert project.__file__.startswith(REF)
        assert lib_image.__file__.startswith(overlay)
        assert _deps.smart() is imageanalysis_amd.smart and smart.smart_node is props.getNode('/smart', True)
        for name, mod in (('matcher', matcher), ('optimizer', optimizer), ('smart', smart),
                          ('match_cleanup', match_cleanup), ('groups', groups)):
            assert name in _deps.REPLACED and mod.__file__.startswith(overlay), name
        assert smart.update_surface_estimate is imageanalysis_amd.smart.update_surface_estimate
        assert matcher.find_matches is imageanalysis_amd.matcher.find_matches
        assert optimizer.Optimizer is imageanalysis_amd.optimizer.Optimizer
        for name in _deps.REPLACED:
            assert 'lib._orig_' + name not in sys.modules
            for m in list(sys.modules.values()):
                f = getattr(m, '__file__', None) or ''
                assert not (f == os.path.join(REF, 'lib', name + '.py')), \
                    "the reference's lib/%s.py was imported by the drop-in" % name
        from imageanalysis_amd import image as mirror_image
        for meth in ('load_features', 'load_descriptors', 'save_features', 'save_descriptors',
                     'load_matches', 'save_matches', 'detect_features', 'load_rgb'):
            assert getattr(lib_image.Image, meth) is getattr(mirror_image.Image, meth), meth
        ok('bindings: props tree, lib.camera, lib.logger are the reference\'s; lib.{%s} are the shim '
           'files; lib.image.Image carries the installed methods' % ','.join(_deps.REPLACED))

        # ---- 2. API surface --------------------------------------------------------------
        for name, mod in (('smart', smart), ('match_cleanup', match_cleanup), ('groups', groups)):
            want = top_level_defs(os.path.join(REF, 'lib', name + '.py'))
            missing = [f for f in want if not hasattr(mod, f)]
            assert not missing, "lib/%s.py functions the mirror lacks: %s" % (name, missing)
            ok('API: all %d top-level functions of lib/%s.py exist in the mirror' % (len(want), name))
        live = dict(matcher=['configure', 'find_matches', 'saveMatches', 'raw_matches', 'basic_pair_matches',
                             'bidirectional_pair_matches', 'filter_duplicates', 'filter_cross_check'],
                    optimizer=['Optimizer'])
        for name, mod in (('matcher', matcher), ('optimizer', optimizer)):
            assert all(hasattr(mod, f) for f in live[name])
            rest = [f for f in top_level_defs(os.path.join(REF, 'lib', name + '.py')) if not hasattr(mod, f)]
            ok('API: live entry points of lib/%s.py present (not mirrored, off the live path: %s)'
               % (name, ', '.join(rest) or '-'))

        # ---- 3. files against the reference's ORIGINAL loaders -----------------------------
        with quiet():
            orig_image = load_original('image')
        proj_dir = os.path.join(work, 'files')
        with quiet():
            proj = project.ProjectMgr(proj_dir, create=True)
        rng = np.random.default_rng(3)
        n = 1500
        ours = lib_image.Image(proj.analysis_dir, 'A001')
        ours.kp_list = keypoints.KeyPointList(rng.uniform(0, 5000, n), rng.uniform(0, 3000, n),
                                              rng.uniform(2, 40, n), rng.uniform(0, 360, n),
                                              rng.uniform(0.01, 0.1, n), rng.integers(0, 1 << 24, n))
        ours.des_list = rng.integers(0, 256, (n, 128)).astype(np.float32)
        from imageanalysis_amd.matchpairs import MatchPairs
        ours.match_list = {'B002': MatchPairs(rng.integers(0, n, (300, 2)).astype(np.int32)), 'C003': []}
        ours.save_features(); ours.save_descriptors(); ours.save_matches()
        cacheio.wait()
        theirs = orig_image.Image(proj.analysis_dir, 'A001')
        assert theirs.load_features() and theirs.load_descriptors()
        theirs.load_matches()
        assert len(theirs.kp_list) == n and type(theirs.kp_list) is list
        for k in (0, 1, n - 1):
            a, b = ours.kp_list[k], theirs.kp_list[k]
            assert (tuple(a.pt), a.size, a.angle, a.response, a.octave, a.class_id) == \
                (tuple(b.pt), b.size, b.angle, b.response, b.octave, b.class_id)
        assert np.array_equal(theirs.des_list, ours.des_list) and theirs.des_list.dtype == np.float32
        assert theirs.match_list == {'B002': [list(map(int, p)) for p in ours.match_list['B002']], 'C003': []}
        assert type(theirs.match_list['B002']) is list and type(theirs.match_list['B002'][0]) is list
        ok('files: .feat / .desc / .match written by the installed methods load with the reference\'s '
           'original Image.load_* (%d keypoints, 300 matches)' % n)
        theirs2 = orig_image.Image(proj.analysis_dir, 'D004')
        theirs2.kp_list = [cv2.KeyPoint(float(x), float(y), 3.5, 12.0, 0.05, 258, -1)
                           for x, y in rng.uniform(0, 3000, (200, 2)).astype(np.float32)]
        theirs2.des_list = rng.integers(0, 256, (200, 128)).astype(np.float32)
        theirs2.match_list = {'A001': [[1, 2], [3, 4]]}
        theirs2.save_features(); theirs2.save_descriptors(); theirs2.save_matches()
        back = lib_image.Image(proj.analysis_dir, 'D004')
        assert back.load_features() and back.load_descriptors()
        back.load_matches()
        assert [tuple(k.pt) for k in back.kp_list] == [tuple(k.pt) for k in theirs2.kp_list]
        assert back.kp_list[5].octave == 258 and np.array_equal(back.des_list, theirs2.des_list)
        assert back.match_list == theirs2.match_list
        ok('files: caches written by the reference\'s original Image.save_* load with the installed methods')

        # ---- 4. the scenario inside the reference environment ---------------------------------
        from oracle import dropin_scenario

        def make_project(names, directory):
            with quiet():
                p = project.ProjectMgr(directory, create=True)
                p.image_list = [lib_image.Image(p.analysis_dir, nm) for nm in names]
            return p

        def make_keypoints(xy):
            return [cv2.KeyPoint(float(x), float(y), 3.0) for x, y in xy]

        env = dropin_scenario.Env(make_project, camera, props.getNode, matcher, smart, match_cleanup,
                                  groups, optimizer, make_keypoints, cacheio.wait)
        with quiet():
            result = dropin_scenario.run(env, os.path.join(work, 'scenario'))
        ok('scenario: feature cache, saveMatches, /smart bookkeeping (== the reference\'s lib/smart.py '
           'tree of the golden), srtm / yaw hooks, consolidation + groups, Optimizer.setup / '
           'update_camera_poses / refit on lib.project.ProjectMgr + lib.image.Image')
        meta = os.path.join(work, 'scenario', 'ba', 'ImageAnalysis', 'meta')
        n_json = len([f for f in os.listdir(meta) if f.endswith('.json')])
        assert n_json > 0
        with quiet():
            p2 = project.ProjectMgr(os.path.join(work, 'scenario', 'ba'), create=False)
            p2.load_images_info()
        assert len(p2.image_list) == n_json
        ok('pose JSON: %d meta/*.json written through ProjectMgr.save_images_info and read back by '
           'load_images_info' % n_json)

        # ---- 5. the reference's original lib/smart.py beside the mirror ----------------------
        with quiet():
            orig_smart = load_original('smart')
        assert orig_smart.smart_node is smart.smart_node
        with open(os.path.join(GOLD, 'smart_grid.pkl'), 'rb') as f:
            g = pickle.load(f)
        sproj = make_project(g['names'], os.path.join(work, 'scenario', 'project'))
        for im, pose, xy in zip(sproj.image_list, g['poses'], g['xy']):
            im.kp_list = make_keypoints(np.asarray(xy, np.float32))
        interp = lambda p: [250.0 + 0.01 * p[0] - 0.02 * p[1]]

        class _Srtm(object):
            ned_interp = staticmethod(lambda p: np.array(interp(p)))
        snap = lambda: {n_: dropin_scenario._tree(smart.smart_node.getChild(n_, True)) for n_ in g['names']}
        mine = snap()
        for n_ in g['names']:
            smart.smart_node.getChild(n_, True).__dict__.pop('srtm_surface_m', None)
        orig_smart.srtm = _Srtm
        with quiet():
            orig_smart.update_srtm_elevations(sproj)
        assert snap() == mine
        with quiet():
            orig_smart.set_yaw_error_estimates(sproj)
        theirs_yaw = {im.name: im.node.getChild('aircraft_pose', True).getFloat('yaw_error_deg')
                      for im in sproj.image_list}
        assert theirs_yaw == result['yaw_error_deg']
        ok('lib/smart.py (original) update_srtm_elevations / set_yaw_error_estimates leave the tree and '
           'the images as the mirror does')
        tri = []
        # (set_aircraft_yaw_error_estimate above re-derived every camera pose from the aircraft
        #  pose and the mount, lib/image.py:434-460: back to the golden's poses first)
        for im, pose in zip(sproj.image_list, g['poses']):
            im.set_camera_pose(pose['ned'], *pose['ypr'])
        for rec in g['pairs'][:4]:
            a, b = sproj.image_list[rec['i']], sproj.image_list[rec['j']]
            a.match_list[b.name] = [list(p) for p in rec['matches']]
            with quiet():
                pts = orig_smart.triangulate_features(a, b)
            pts = np.asarray(pts, np.float64)
            # the same call chain produced the golden's surface estimate: -mean / std of "down"
            assert abs(-np.average(pts[2]) - rec['avg']) < 1e-9 * max(1.0, abs(rec['avg']))
            assert abs(np.std(pts[2]) - rec['std']) < 1e-9 * max(1.0, rec['std'])
            tri.append(dict(i=rec['i'], j=rec['j'], points=pts))
        result['triangulate_features'] = tri
        ok('lib/smart.py (original) triangulate_features: %d pairs recorded for tests/test_dropin.py'
           % len(tri))

        # ---- 6. find_matches inside the reference environment == the reference's own loop (G9) ----
        # lib.matcher.find_matches here is imageanalysis_amd.matcher.find_matches (the shim file),
        # driven on the reference's ProjectMgr / lib.image.Image / props tree / lib.camera; the two
        # DEVICE steps are replaced by the oracle stand-ins of tests/test_find_matches_loop.py
        # (there is no GPU in this container) -- the schedule, the pose feedback through the REAL
        # Image.set_aircraft_yaw_error_estimate / get_aircraft_pose / get_body2cam, the bookkeeping
        # on the REAL /smart tree, .match files and smart.json through the reference's writers are
        # the product's.  Compared with what the reference's ORIGINAL lib/matcher.py did on the same
        # project (tests/golden/find_matches_strip.pkl, oracle/gen_golden.py G9).
        sys.path.insert(0, os.path.join(REPO, 'tests'))
        import test_find_matches_loop as fml
        import imageanalysis_amd.matcher as amx_matcher
        g9 = fml._golden()
        r2d = 180.0 / np.pi
        saved = {n_: getattr(amx_matcher, n_) for n_ in ('_launch_batch', '_finish_batch', '_surface_device',
                                                        'the_matcher', 'PAIRS_PER_BATCH', 'max_distance',
                                                        'min_pairs')}
        try:
            for sort in (True, False):
                for n_ in list(props.getNode('/images', True).__dict__):
                    del props.getNode('/images', True).__dict__[n_]
                smart.smart_node.__dict__.clear()
                smart.load(None)
                fproj = make_project(g9['names'], os.path.join(work, 'g9_sort%d' % int(sort)))
                Kg = g9['K']
                camera.set_K(Kg[0], Kg[4], Kg[2], Kg[5])
                camera.set_image_params(g9['width'], g9['height'])
                camera.set_mount_params(*g9['mount'])
                # (the nodes the module bound when it was imported, like the reference's own
                #  `matcher_node = getNode('/config/matcher', True)` at lib/matcher.py:31)
                amx_matcher.detector_node.setString('detector', 'SIFT')
                amx_matcher.detector_node.setFloat('scale', 0.4)
                amx_matcher.matcher_node.setFloat('match_ratio', g9['match_ratio'])
                amx_matcher.matcher_node.setInt('min_pairs', g9['min_pairs'])
                for key in ('schedule', 'min_dist', 'max_dist'):
                    amx_matcher.matcher_node.__dict__.pop(key, None)
                body2cam = camera.get_body2cam()
                for i, im in enumerate(fproj.image_list):
                    rep = g9['reported'][i]
                    im.set_aircraft_pose(*g9['aircraft_lla'], *rep['ypr'])
                    ned2body = [im.node.getChild('aircraft_pose').getFloatEnum('quat', k) for k in range(4)]
                    y, p_, r_ = _tf.euler_from_quaternion(_tf.quaternion_multiply(ned2body, body2cam), 'rzyx')
                    im.set_camera_pose(rep['ned'], y * r2d, p_ * r2d, r_ * r2d)
                    im.des_list = g9['des'][i].astype(np.float32)
                    im.kp_list = make_keypoints(g9['xy'][i])
                assert [fml._pose_record(im) for im in fproj.image_list] == g9['runs'][sort]['initial']
                amx_matcher.max_distance, amx_matcher.min_pairs = 270.0, float(g9['min_pairs'])
                amx_matcher.the_matcher = object()
                amx_matcher._launch_batch = fml._oracle_launch
                amx_matcher._finish_batch = lambda handle: handle
                amx_matcher._surface_device = fml._oracle_surface
                amx_matcher.PAIRS_PER_BATCH = 6
                for call in range(2):
                    with quiet():
                        matcher.find_matches(fproj, None, strategy='traditional', transform='gms', sort=sort)
                    fml.check_against(g9, g9['runs'][sort], call, fproj)
                    # ... and the files: .match through the reference's ORIGINAL reader, smart.json
                    for i, im in enumerate(fproj.image_list):
                        back = orig_image.Image(fproj.analysis_dir, im.name)
                        with quiet():
                            back.load_matches()
                        assert back.match_list == g9['runs'][sort]['calls'][call]['match_lists'][i], im.name
                    import json
                    with open(os.path.join(fproj.analysis_dir, 'smart.json')) as fp:
                        assert json.load(fp) == g9['runs'][sort]['calls'][call]['smart']
        finally:
            for n_, v in saved.items():
                setattr(amx_matcher, n_, v)
        ok('find_matches on lib.project.ProjectMgr / lib.image.Image / the props tree == the reference\'s '
           'ORIGINAL lib.matcher.find_matches (G9): match lists, /smart, poses, .match files, smart.json; '
           'sort=True and False, first and second call')

        with open(os.path.join(GOLD, 'dropin_env.pkl'), 'wb') as f:
            pickle.dump(result, f, protocol=4)
        print('wrote tests/golden/dropin_env.pkl')
    finally:
        shutil.rmtree(overlay, ignore_errors=True)
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    try:
        main()
    except SystemExit:
        # (the reference's modules call quit() where they give up: that is a failed check, not rc 0)
        print('FAILED: a module called quit() / sys.exit() inside the check')
        os._exit(1)
