"""The drop-in's two environments (imageanalysis_amd/_deps.py) behave the same.

tests/golden/dropin_env.pkl was written by oracle/check_dropin.py INSIDE the reference environment
(props importable, the reference's own lib/{project,image,camera,logger}.py, INTEGRATION.md's shim
files in place of lib/{matcher,optimizer,smart,match_cleanup,groups}.py); here the same scripted
pass over every host-only entry point (oracle/dropin_scenario.py) runs with the package's
stand-ins and must produce the same files, trees and poses."""
import os
import pickle
import sys

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _fixture():
    with open(os.path.join(GOLDEN, 'dropin_env.pkl'), 'rb') as f:
        return pickle.load(f)


def test_deps_never_binds_a_replaced_reference_module():
    """the mirror replaces lib/{matcher,optimizer,smart,match_cleanup,groups}.py: _deps may bind the
    reference's camera / logger / srtm, never one of those (round 4 bound lib.smart, which sent
    every matching pair of find_matches through the reference's per-pair CPU path)"""
    from imageanalysis_amd import _deps
    mirror = lambda m: m.__name__ == 'imageanalysis_amd.smart' and hasattr(m, 'record_round')
    assert set(_deps.REPLACED) == {'matcher', 'optimizer', 'smart', 'match_cleanup', 'groups'}
    assert mirror(_deps.smart())
    for name in _deps.REPLACED:
        with pytest.raises(AssertionError):
            _deps._ref_or_host(name)
    # ... also with the reference environment simulated: a `lib` package whose smart module would
    # be importable must not be picked up
    import types
    fake_lib = types.ModuleType('lib')
    fake_lib.__path__ = []
    fake_smart = types.ModuleType('lib.smart')
    saved = {k: sys.modules.get(k) for k in ('lib', 'lib.smart')}
    was = _deps.HAVE_PROPS
    try:
        sys.modules['lib'], sys.modules['lib.smart'] = fake_lib, fake_smart
        _deps.HAVE_PROPS = True
        assert mirror(_deps.smart())
    finally:
        _deps.HAVE_PROPS = was
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    src = open(_deps.__file__).read()
    assert "import_module('lib.smart')" not in src


def test_smart_mirror_has_the_reference_module_surface():
    """every function process.py / lib.matcher / lib.match_cleanup call on lib.smart
    (scripts/lib/smart.py:26-339) exists in the mirror"""
    from imageanalysis_amd import smart
    for name in ('triangulate_features', 'find_affine', 'decompose_affine', 'estimate_surface_elevation',
                 'estimate_yaw_error', 'update_surface_estimate', 'update_yaw_error_estimate',
                 'get_yaw_error_estimate', 'get_surface_estimate', 'update_srtm_elevations',
                 'set_yaw_error_estimates', 'load', 'save', 'smart_node'):
        assert hasattr(smart, name), name


def test_scenario_equals_the_reference_environment(tmp_path):
    from imageanalysis_amd import _deps, cacheio, groups, match_cleanup, matcher, optimizer, smart
    from imageanalysis_amd import image as iimg
    from imageanalysis_amd.hostlib import camera
    from oracle import dropin_scenario
    assert not _deps.HAVE_PROPS          # (the other branch is oracle/check_dropin.py's)
    want = _fixture()

    class Project(object):
        def __init__(self, names, directory):
            self.analysis_dir = os.path.join(directory, 'ImageAnalysis')
            for sub in ('meta', 'cache'):
                os.makedirs(os.path.join(self.analysis_dir, sub), exist_ok=True)
            self.image_list = [iimg.Image(self.analysis_dir, n) for n in names]

        def findIndexByName(self, name):
            for i, im in enumerate(self.image_list):
                if im.name == name:
                    return i
            return None

        def findImageByName(self, name):
            i = self.findIndexByName(name)
            return None if i is None else self.image_list[i]

        def save_images_info(self):
            pass

    def make_keypoints(xy):
        return [iimg.make_keypoint(float(x), float(y), 3.0, -1.0, 0.0, 0) for x, y in xy]

    env = dropin_scenario.Env(Project, camera, _deps.getNode, matcher, smart, match_cleanup, groups,
                              optimizer, make_keypoints, cacheio.wait)
    got = dropin_scenario.run(env, str(tmp_path))
    skip = {'triangulate_features'}
    assert set(got) == set(want) - skip
    for key in sorted(got):
        assert got[key] == want[key], key


@pytest.mark.gpu
def test_triangulate_features_equals_reference():
    """smart.triangulate_features (iamx_triangulate_pairs_xyz) against the [4, N] arrays the
    reference's own lib/smart.py:26-63 returned in the reference environment
    (oracle/check_dropin.py; its cv2.triangulatePoints is the published DLT of oracle/shims/cv2.py,
    float64 with another factorisation -> 1e-6 relative)"""
    from imageanalysis_amd import smart
    from imageanalysis_amd import image as iimg
    from imageanalysis_amd.hostlib import camera
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    want = _fixture()['triangulate_features']
    with open(os.path.join(GOLDEN, 'smart_grid.pkl'), 'rb') as f:
        g = pickle.load(f)
    proj = PoseProject(g['names'])
    K = g['K']
    camera.set_K(K[0], K[4], K[2], K[5])
    for im, pose, xy in zip(proj.image_list, g['poses'], g['xy']):
        im.set_camera_pose(pose['ned'], *pose['ypr'])
        im.kp_list = [iimg.make_keypoint(float(x), float(y), 3.0, -1.0, 0.0, 0) for x, y in xy]
    assert len(want) >= 3
    recs = {(r['i'], r['j']): r for r in g['pairs']}
    for w in want:
        a, b = proj.image_list[w['i']], proj.image_list[w['j']]
        a.match_list[b.name] = recs[(w['i'], w['j'])]['matches']
        pts = smart.triangulate_features(a, b)
        assert pts.shape == w['points'].shape and pts.shape[0] == 4
        assert np.array_equal(pts[3], np.ones(pts.shape[1]))
        scale = np.abs(w['points'][:3]).max()
        assert np.abs(pts - w['points']).max() <= 1e-6 * scale
    assert smart.triangulate_features(a, a) is None
    b.match_list.pop(a.name, None)
    assert smart.triangulate_features(b, a) is None
