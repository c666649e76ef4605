"""Match consolidation against the golden vectors produced by the reference's
own scripts/lib/match_cleanup.py (oracle/gen_golden.py G7).  The chain linking is native HOST code
in libiamx.so (no device work), so these run without a GPU; the triangulation test is the GPU one."""
import copy
import glob
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN

CASES = sorted(glob.glob(os.path.join(GOLDEN, 'cleanup_*.pkl')))


class KP(object):
    def __init__(self, x, y):
        self.pt = (float(x), float(y))


def _project(g):
    from imageanalysis_amd.hostlib import camera
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    inp = g['inputs']
    proj = PoseProject(inp['names'])
    for i, im in enumerate(proj.image_list):
        im.kp_list = [KP(x, y) for x, y in inp['xy'][i]]
        im.match_list = copy.deepcopy(inp['match_lists'][i])
        im.set_camera_pose(inp['poses'][i]['ned'], *inp['poses'][i]['ypr'])
    K = inp['K']
    camera.set_K(K[0], K[4], K[2], K[5])
    return proj


@pytest.mark.parametrize('path', CASES, ids=os.path.basename)
def test_consolidation_equals_reference(path):
    from imageanalysis_amd import match_cleanup
    with open(path, 'rb') as f:
        g = pickle.load(f)
    proj = _project(g)
    match_cleanup.merge_duplicates(proj)
    match_cleanup.check_for_pair_dups(proj)
    match_cleanup.check_for_1vn_dups(proj)
    for im, want, used in zip(proj.image_list, g['match_lists_after'], g['kp_used']):
        assert list(im.match_list.keys()) == list(want.keys())
        for k in want:
            assert im.match_list[k] == want[k], (im.name, k)
        assert np.array_equal(im.kp_used, used)
    direct = match_cleanup.make_match_structure(proj)
    assert direct == g['matches_direct']
    grouped = match_cleanup.link_matches(proj, direct)
    assert grouped == g['matches_grouped']
    assert all(type(p[0]) is int and type(p[1][0]) is float for m in grouped for p in m[2:])
    pickle.loads(pickle.dumps(grouped))                   # plain python: the matches_grouped pickle


def test_link_matches_scale_and_determinism():
    """a larger random survey: native linking == a literal python transcription of the rules"""
    from imageanalysis_amd import match_cleanup
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    rng = np.random.default_rng(5)
    n_img, n_kp = 40, 3000
    proj = PoseProject(['L%03d' % i for i in range(n_img)])
    for im in proj.image_list:
        xy = np.stack([rng.uniform(0, 5000, n_kp), rng.uniform(0, 3000, n_kp)], 1).astype(np.float32)
        im.kp_list = [KP(x, y) for x, y in xy]
    direct = []
    for i in range(n_img):
        for j in range(i + 1, min(i + 4, n_img)):
            a = rng.choice(n_kp, 800, replace=False)
            b = rng.choice(n_kp, 800, replace=False)
            direct += [[None, -1, [i, int(x)], [j, int(y)]] for x, y in zip(a, b)]

    def python_rules(matches):                           # match_cleanup.py:246-286, verbatim logic
        matches = [list(m) for m in matches]
        while True:
            new, lookup = [], {}
            for match in matches:
                index = -1
                for p in match[2:]:
                    if (p[0], p[1]) in lookup:
                        index = lookup[(p[0], p[1])]
                        break
                if index < 0:
                    for p in match[2:]:
                        lookup[(p[0], p[1])] = len(new)
                    new.append(list(match))
                else:
                    existing = new[index]
                    for p in match[2:]:
                        if not any(p[0] == e[0] for e in existing[2:]):
                            existing.append(list(p))
                            lookup[(p[0], p[1])] = index
            if len(new) == len(matches):
                return new
            matches = new

    want = python_rules(direct)
    want.sort(key=len, reverse=True)
    got = match_cleanup.link_matches(proj, direct)
    assert len(got) == len(want) and len(got) < len(direct)
    for a, b in zip(got, want):
        assert [p[0] for p in a[2:]] == [p[0] for p in b[2:]]
        for p, q in zip(a[2:], b[2:]):
            assert p[1] == list(proj.image_list[q[0]].kp_list[q[1]].pt)


@pytest.mark.parametrize('min_chain_len', [0, 2])
@pytest.mark.parametrize('path', CASES, ids=os.path.basename)
def test_groups_compute_equals_reference(path, min_chain_len):
    from imageanalysis_amd import groups
    from imageanalysis_amd._deps import getNode
    with open(path, 'rb') as f:
        g = pickle.load(f)
    proj = _project(g)
    matches = copy.deepcopy(g['matches_triangulated'])
    getNode('/config/matcher', True).setInt('min_chain_len', min_chain_len)
    try:
        got = groups.compute(proj.image_list, matches)
    finally:
        getNode('/config/matcher', True).setInt('min_chain_len', 0)
    want = g['groups'][min_chain_len]
    assert got == want['groups']                          # same groups, same order of names
    assert [m[1] for m in matches] == want['levels']


@pytest.mark.parametrize('path', CASES, ids=os.path.basename)
def test_chains_array_form_equals_the_lists(path):
    """link_matches() returns match_cleanup.Chains: arrays behind the reference's list of lists.
    Its own consumers read / write the arrays (groups.compute here; triangulate_smart and
    Optimizer.setup / refit in the GPU tests), pickling does not turn it into lists, indexing does
    -- and from then on the lists are the truth."""
    from imageanalysis_amd import groups, match_cleanup
    with open(path, 'rb') as f:
        g = pickle.load(f)
    proj = _project(g)
    match_cleanup.merge_duplicates(proj)
    match_cleanup.check_for_pair_dups(proj)
    chains = match_cleanup.link_matches(proj, match_cleanup.make_match_structure(proj))
    assert isinstance(chains, match_cleanup.Chains) and chains.untouched()
    assert len(chains) == len(g['matches_grouped'])
    assert pickle.loads(pickle.dumps(chains)) == g['matches_grouped'] and chains.untouched()
    # grouping on the arrays == grouping on the lists
    tri = match_cleanup.Chains.from_lists(g['matches_triangulated'])
    assert pickle.loads(pickle.dumps(tri)) == g['matches_triangulated']
    got = groups.compute(proj.image_list, tri)
    assert tri.untouched()
    want = g['groups'][0]
    assert got == want['groups'] and tri.group.tolist() == want['levels']
    assert [m[1] for m in pickle.loads(pickle.dumps(tri))] == want['levels']
    # a reader that indexes gets lists it may edit; the edit is seen by everybody afterwards
    first = tri[0]
    assert not tri.untouched() and first[1] == want['levels'][0]
    first[1] = 99
    assert pickle.loads(pickle.dumps(tri))[0][1] == 99 and list(tri)[0][1] == 99


def test_groups_save_load_roundtrip(tmp_path):
    from imageanalysis_amd import groups
    gl = [['b', 'a', 'c'], ['z']]
    groups.save(str(tmp_path), gl)
    assert groups.load(str(tmp_path)) == gl
    assert groups.load(str(tmp_path / 'missing')) == []


@pytest.mark.gpu
@pytest.mark.parametrize('path', CASES, ids=os.path.basename)
def test_triangulate_smart_equals_reference(path):
    from imageanalysis_amd import match_cleanup
    from imageanalysis_amd._deps import getNode
    with open(path, 'rb') as f:
        g = pickle.load(f)
    proj = _project(g)
    for name, b in g['base_elev'].items():
        getNode('/smart', True).getChild(name, True).setFloat('tri_surface_m', b)
    matches = copy.deepcopy(g['matches_grouped'])
    match_cleanup.triangulate_smart(proj, matches)
    want = g['matches_triangulated']
    assert len(matches) == len(want)
    got = np.array([m[0] for m in matches])
    ref = np.array([m[0] for m in want])
    assert got.shape == ref.shape == (len(want), 3)
    assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())     # f64, same formulas
    assert all(m[2:] == w[2:] for m, w in zip(matches, want))
    assert all(type(m[0]) is list and type(m[0][0]) is float for m in matches)
    # the array-backed form: same points, written into the arrays, no lists built
    chains = match_cleanup.Chains.from_lists(g['matches_grouped'])
    match_cleanup.triangulate_smart(proj, chains)
    assert chains.untouched() and chains.has_ned.all()
    assert np.array_equal(chains.ned, got)
    assert pickle.loads(pickle.dumps(chains)) == matches


@pytest.mark.gpu
def test_surface_estimate_equals_reference():
    """imageanalysis_amd.smart (device DLT + the property-tree bookkeeping) against the
    reference's lib/smart.py outputs (oracle/gen_golden.py G8)."""
    from imageanalysis_amd import smart
    from imageanalysis_amd.hostlib import camera
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    with open(os.path.join(GOLDEN, 'smart_grid.pkl'), 'rb') as f:
        g = pickle.load(f)
    proj = PoseProject(g['names'])
    K = g['K']
    camera.set_K(K[0], K[4], K[2], K[5])
    for im, pose, xy in zip(proj.image_list, g['poses'], g['xy']):
        im.set_camera_pose(pose['ned'], *pose['ypr'])
        im.kp_list = [KP(x, y) for x, y in xy]
        smart.smart_node.__dict__.pop(im.name, None)
    n_bad = 0
    for rec in g['pairs']:
        a, b = proj.image_list[rec['i']], proj.image_list[rec['j']]
        a.match_list[b.name] = rec['matches']
        b.match_list[a.name] = [[q, p] for p, q in rec['matches']]
        avg, std = smart.update_surface_estimate(a, b)
        scale = max(1.0, abs(rec['avg']), rec['std'])
        assert abs(avg - rec['avg']) <= 1e-6 * scale and abs(std - rec['std']) <= 1e-6 * scale, rec['i']
        n_bad += rec['std'] >= 25
    assert 0 < n_bad < len(g['pairs'])                       # scrambled pairs are in the set
    for im in proj.image_list:
        node = smart.smart_node.getChild(im.name, True)
        want = g['tri_surface_m'][im.name]
        assert (node.getFloat('tri_surface_m') if node.hasChild('tri_surface_m') else None) == want
    assert abs(g['tri_surface_m'][g['names'][0]] - g['ground']) < 0.5


@pytest.mark.gpu
def test_yaw_error_estimate_equals_reference():
    """imageanalysis_amd.smart's yaw-error estimate (device similarity fit + the reference's
    course arithmetic and property-tree weighting, scripts/lib/smart.py:66-115,138-192,251-283)
    against the outputs of the reference's own lib/smart.py (oracle/gen_golden.py G8, whose
    cv2.estimateAffinePartial2D is the documented deterministic stand-in of oracle/shims/cv2.py:
    the fit is float64 with different summation orders on the two sides -> 1e-9)."""
    from imageanalysis_amd import smart
    from imageanalysis_amd.hostlib import camera
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    with open(os.path.join(GOLDEN, 'smart_grid.pkl'), 'rb') as f:
        g = pickle.load(f)
    proj = PoseProject(g['names'])
    K = g['K']
    camera.set_K(K[0], K[4], K[2], K[5])
    camera.set_image_params(5472, 3648)
    for im, pose, xy in zip(proj.image_list, g['poses'], g['xy']):
        im.set_camera_pose(pose['ned'], *pose['ypr'])
        im.set_aircraft_pose(45.0, -93.0, 300.0, pose['air_yaw'], 0.0, 0.0)
        im.kp_list = [KP(x, y) for x, y in xy]
        smart.smart_node.__dict__.pop(im.name, None)
    n_nonzero = 0
    for rec in g['pairs']:
        a, b = proj.image_list[rec['i']], proj.image_list[rec['j']]
        a.match_list[b.name] = rec['matches']
        b.match_list[a.name] = [[q, p] for p, q in rec['matches']]
        for (x, y, key) in ((a, b, 'ab'), (b, a, 'ba')):
            aff = smart.find_affine(x, y)
            want = rec['affine_' + key]
            assert np.abs(aff - want).max() <= 1e-9 * max(1.0, np.abs(want).max()), (rec['i'], key)
            got = smart.record_yaw_error_estimate(x, y, aff)
            assert abs(got - rec['yaw_' + key]) <= 1e-6, (rec['i'], rec['j'], key)
            n_nonzero += rec['yaw_' + key] != 0
    assert n_nonzero >= 4
    for im in proj.image_list:
        node = smart.smart_node.getChild(im.name, True)
        want = g['yaw'][im.name]
        assert (node.getFloat('yaw_error') if node.hasChild('yaw_error') else None) == want['yaw_error']
        yp = node.getChild('yaw_pairs', True)
        assert sorted(yp.getChildren()) == sorted(want['pairs'])
        for c, vals in want['pairs'].items():
            got = tuple(yp.getChild(c).getFloat(k) for k in ('yaw_error', 'dist_m', 'relative_crs', 'weight'))
            assert got == vals, (im.name, c)
    # a pair without matches records nothing and resets the estimate to 0 like the reference
    a, b = proj.image_list[0], proj.image_list[1]
    a.match_list[b.name] = []
    assert smart.update_yaw_error_estimate(a, b) == 0


def test_consolidation_host_helpers_equal_numpy():
    """libiamx host helpers of the consolidation stage against the numpy expressions they
    replace: first occurrence per key (merge_duplicates, match_cleanup.py:19-104) and the stable
    longest-first order of the linked chains (match_cleanup.py:291-292)."""
    import ctypes
    from imageanalysis_amd import _lib
    L = _lib.lib()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)          # noqa: E731
    rng = np.random.default_rng(8)
    for n in (0, 1, 17, 40000):
        key = np.ascontiguousarray(rng.integers(-50, 50 + n // 3, n) * 4294967297, np.int64)
        first = np.full(n, -7, np.int64)
        _lib.check(L.iamx_first_occurrence(P(key), n, P(first)), 'iamx_first_occurrence')
        _u, idx, inv = np.unique(key, return_index=True, return_inverse=True)
        assert np.array_equal(first, idx[inv] if n else first)
    for n_chains in (0, 1, 5, 30000):
        lens = rng.integers(2, 9, n_chains)
        if n_chains > 10:
            lens[rng.integers(0, n_chains, 5)] = 40                # a few long ones
        ptr = np.zeros(n_chains + 1, np.int64)
        np.cumsum(lens, out=ptr[1:])
        total = int(ptr[-1])
        img = rng.integers(0, 500, total).astype(np.int32)
        kp = rng.integers(0, 40000, total).astype(np.int32)
        o_img, o_kp = np.full(total, -1, np.int32), np.full(total, -1, np.int32)
        o_ptr = np.full(n_chains + 1, -1, np.int64)
        for threads in (1, 3):
            _lib.check(L.iamx_chains_longest_first(P(img), P(kp), P(ptr), n_chains, P(o_img), P(o_kp),
                                                   P(o_ptr), threads), 'iamx_chains_longest_first')
            order = np.argsort(-lens, kind='stable')
            want_ptr = np.zeros(n_chains + 1, np.int64)
            np.cumsum(lens[order], out=want_ptr[1:])
            take = np.concatenate([np.arange(ptr[c], ptr[c + 1]) for c in order]) if n_chains else \
                np.zeros(0, np.int64)
            assert np.array_equal(o_ptr, want_ptr)
            assert np.array_equal(o_img, img[take]) and np.array_equal(o_kp, kp[take])


def test_link_matches_skips_only_the_pass_that_changes_nothing():
    """iamx_link_matches stops after a pass that left pairwise disjoint chains instead of running
    the reference's final no-change pass (match_cleanup.py:223-274 loops until the count stays):
    same chains, same order, same reported pass count as with IAMX_LINK_VERIFY=1, on random
    surveys whose pair matches need several passes."""
    import ctypes
    import subprocess
    import sys
    code = r'''
import ctypes, os, sys, numpy as np
sys.path.insert(0, %r)
from imageanalysis_amd import _lib
L = _lib.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
out = []
for seed in range(6):
    rng = np.random.default_rng(seed)
    n_img, n_kp, n = 12, 300, 4000 + 500 * seed
    img = np.empty(2 * n, np.int32); kp = np.empty(2 * n, np.int32)
    a = rng.integers(0, n_img, n); b = (a + 1 + rng.integers(0, n_img - 1, n)) %% n_img
    img[0::2], img[1::2] = a, b
    # keypoints tied to a small set of world points, with mismatches: chains that meet late
    world = rng.integers(0, 500, n)
    kp[0::2] = (world * 7 + a) %% n_kp
    kp[1::2] = np.where(rng.random(n) < 0.9, (world * 7 + b) %% n_kp, rng.integers(0, n_kp, n))
    ptr = np.arange(n + 1, dtype=np.int64) * 2
    o_img, o_kp = np.empty_like(img), np.empty_like(kp)
    o_ptr = np.zeros(n + 1, np.int64); passes = np.zeros(1, np.int32)
    nc = int(L.iamx_link_matches(P(img), P(kp), P(ptr), n, P(o_img), P(o_kp), P(o_ptr), P(passes)))
    tot = int(o_ptr[nc])
    out.append((nc, int(passes[0]), o_ptr[:nc + 1].tobytes(), o_img[:tot].tobytes(), o_kp[:tot].tobytes()))
import pickle
sys.stdout.buffer.write(pickle.dumps(out))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import pickle
    runs = []
    for verify in (False, True):
        env = dict(os.environ)
        env.pop('IAMX_LINK_VERIFY', None)
        if verify:
            env['IAMX_LINK_VERIFY'] = '1'
        runs.append(pickle.loads(subprocess.run([sys.executable, '-c', code], env=env, check=True,
                                                capture_output=True).stdout))
    assert runs[0] == runs[1]
    assert max(r[1] for r in runs[0]) >= 3                # several passes were needed somewhere


@pytest.mark.parametrize('path', CASES, ids=os.path.basename)
def test_consolidation_of_array_backed_lists_equals_reference(path):
    """the same goldens with the match lists as find_matches leaves them (matchpairs.MatchPairs over
    int32 arrays): compute_kp_usage, the remap of merge_duplicates and the two duplicate checks run
    as native passes over the arrays (iamx_match_lists_scan, iamx_kp_dup_remap), the chains are
    linked from the pair blocks as they lie (iamx_link_pair_blocks) and get their pixel positions
    in one native pass (iamx_chain_members_uv)"""
    from imageanalysis_amd import match_cleanup
    from imageanalysis_amd.keypoints import KeyPointList
    from imageanalysis_amd.matchpairs import MatchPairs
    with open(path, 'rb') as f:
        g = pickle.load(f)
    proj = _project(g)
    for im, xy in zip(proj.image_list, g['inputs']['xy']):
        xy = np.asarray(xy, np.float32)
        z = np.zeros(len(xy), np.float32)
        im.kp_list = KeyPointList(xy[:, 0].copy(), xy[:, 1].copy(), z + 3.0, z, z, z.astype(np.int32))
        im.match_list = {k: (MatchPairs(np.asarray(v, np.int32).reshape(-1, 2)) if len(v) else [])
                         for k, v in im.match_list.items()}
    match_cleanup.merge_duplicates(proj)
    match_cleanup.check_for_pair_dups(proj)
    match_cleanup.check_for_1vn_dups(proj)
    for im, want, used in zip(proj.image_list, g['match_lists_after'], g['kp_used']):
        assert list(im.match_list.keys()) == list(want.keys())
        for k in want:
            assert im.match_list[k] == want[k], (im.name, k)
        assert np.array_equal(im.kp_used, used)
    direct = match_cleanup.make_match_structure(proj)
    grouped = match_cleanup.link_matches(proj, direct)
    assert direct.untouched()                               # (linked from the blocks: no lists were built)
    assert grouped == g['matches_grouped']
    assert direct == g['matches_direct']


def test_link_matches_reports_a_stale_keypoint_index():
    """a match list that refers to a keypoint its image does not have (a .match file stale against
    the .feat): the reference's kp_list[m[1]] raises IndexError; the native passes do too instead of
    reading a neighbouring image's keypoint"""
    from imageanalysis_amd import match_cleanup
    from imageanalysis_amd.matchpairs import MatchPairs
    with open(CASES[0], 'rb') as f:
        g = pickle.load(f)
    proj = _project(g)
    a, b = proj.image_list[0], proj.image_list[1]
    n_b = len(b.kp_list)
    a.match_list = {b.name: MatchPairs(np.array([[0, 1], [2, n_b + 5]], np.int32))}
    b.match_list = {a.name: MatchPairs(np.array([[1, 0], [n_b + 5, 2]], np.int32))}
    for im in proj.image_list[2:]:
        im.match_list = {}
    with pytest.raises(IndexError):
        match_cleanup.merge_duplicates(proj)
    direct = match_cleanup.make_match_structure(proj)
    with pytest.raises(IndexError):
        match_cleanup.link_matches(proj, direct)


def test_consolidation_scans_stay_linear_in_the_number_of_lists():
    """(round 5 regression) the table signature of match_cleanup._scan_lists once multiplied an
    unmasked python integer per list: 133 k lists of a 4186-frame survey cost 6 s per scan.  120 k
    one-row lists must go through the four scans in a few seconds."""
    import time
    from imageanalysis_amd import match_cleanup
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    from imageanalysis_amd.keypoints import KeyPointList
    from imageanalysis_amd.matchpairs import MatchPairs
    n_img, fan = 1200, 50
    proj = PoseProject(['L%04d' % i for i in range(n_img)])
    z = np.zeros(4, np.float32)
    for i, im in enumerate(proj.image_list):
        im.kp_list = KeyPointList(z + [1, 2, 3, 4], z + [5, 6, 7, 8], z + 3, z, z, z.astype(np.int32))
        im.match_list = {}
    for i in range(n_img):
        for d in range(1, fan + 1):
            j = (i + d) % n_img
            proj.image_list[i].match_list[proj.image_list[j].name] = MatchPairs(np.array([[d % 4, (d + 1) % 4]], np.int32))
            proj.image_list[j].match_list[proj.image_list[i].name] = MatchPairs(np.array([[(d + 1) % 4, d % 4]], np.int32))
    t0 = time.perf_counter()
    match_cleanup.merge_duplicates(proj)
    match_cleanup.check_for_pair_dups(proj)
    match_cleanup.check_for_1vn_dups(proj)
    dt = time.perf_counter() - t0
    assert sum(len(im.match_list) for im in proj.image_list) == 2 * n_img * fan
    assert dt < 8.0, dt


def test_duplicate_scan_is_shared_only_while_the_lists_stand(capfd, monkeypatch):
    """check_for_1vn_dups takes the counts check_for_pair_dups' scan left (one native pass instead of
    two) -- but only while no list object, array or length has changed: a list the first check
    repaired, a list replaced or edited in between, and the remap of merge_duplicates all make it
    scan again; the messages equal those of two independent scans"""
    from imageanalysis_amd import match_cleanup
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    from imageanalysis_amd.keypoints import KeyPointList
    from imageanalysis_amd.matchpairs import MatchPairs

    def project(pairs_ab):
        proj = PoseProject(['A', 'B', 'C'])
        z = np.zeros(8, np.float32)
        for im in proj.image_list:
            im.kp_list = KeyPointList(z + np.arange(8), z + 2 * np.arange(8), z + 3, z, z, z.astype(np.int32))
            im.match_list = {}
        a, b, c = proj.image_list
        a.match_list['B'] = MatchPairs(np.array(pairs_ab, np.int32))
        b.match_list['A'] = MatchPairs(np.array(pairs_ab, np.int32)[:, ::-1].copy())
        a.match_list['C'] = MatchPairs(np.array([[0, 1], [2, 3]], np.int32))
        c.match_list['A'] = MatchPairs(np.array([[1, 0], [3, 2]], np.int32))
        return proj

    calls = []
    import imageanalysis_amd._lib as _lib
    real = _lib.lib()

    class Spy(object):
        def __getattr__(self, name):
            fn = getattr(real, name)
            if name == 'iamx_match_lists_scan':
                def wrapped(*a):
                    calls.append(a[9])                      # mode
                    return fn(*a)
                return wrapped
            return fn
    monkeypatch.setattr(_lib, 'lib', lambda: Spy())
    # (1) nothing to repair: ONE mode-4 scan serves both checks; column-0 repeats are still reported
    proj = project([[0, 1], [0, 2], [3, 4]])
    match_cleanup.check_for_pair_dups(proj)
    match_cleanup.check_for_1vn_dups(proj)
    assert calls.count(4) == 1
    # (2) a repeated pair: the first check replaces the list, the second scans the new lists
    del calls[:]
    proj = project([[0, 1], [0, 1], [3, 4]])
    match_cleanup.check_for_pair_dups(proj)
    assert proj.image_list[0].match_list['B'] == [[0, 1], [3, 4]]
    match_cleanup.check_for_1vn_dups(proj)
    assert calls.count(4) == 2
    # (3) a list replaced between the checks (same length, other object)
    del calls[:]
    proj = project([[0, 1], [2, 2], [3, 4]])
    match_cleanup.check_for_pair_dups(proj)
    proj.image_list[0].match_list['B'] = MatchPairs(np.array([[5, 1], [5, 2], [5, 4]], np.int32))
    match_cleanup.check_for_1vn_dups(proj)
    assert calls.count(4) == 2
    # (4) the whole sequence: merge_duplicates' remap in front changes nothing about the sharing
    del calls[:]
    proj = project([[0, 1], [0, 2], [3, 4]])
    match_cleanup.merge_duplicates(proj)
    match_cleanup.check_for_pair_dups(proj)
    match_cleanup.check_for_1vn_dups(proj)
    assert calls.count(4) == 1
    direct = match_cleanup.make_match_structure(proj)
    match_cleanup.link_matches(proj, direct)
    assert getattr(proj, '_iamx_scan4', None) is None       # (the stage's tables end with it)


def test_link_matches_incremental_passes_equal_the_full_walk(capfd):
    """Late passes of iamx_link_matches only walk the chains that share a point with another chain
    (round 5).  Same chains, same order, same number of passes as the full walk (IAMX_LINK_FULL=1) and
    as a literal python transcription of the reference's loop, on surveys whose conflicting matches
    (two keypoints of one image in one chain's reach) force several passes."""
    import ctypes
    from imageanalysis_amd import _lib

    def python_rules(ptr, img, kp):
        matches = [[(int(img[j]), int(kp[j])) for j in range(ptr[m], ptr[m + 1])] for m in range(len(ptr) - 1)]
        passes = 0
        while True:
            passes += 1
            new, lookup = [], {}
            for match in matches:
                index = -1
                for p in match:
                    if p in lookup:
                        index = lookup[p]
                        break
                if index < 0:
                    for p in match:
                        lookup[p] = len(new)
                    new.append(list(match))
                else:
                    existing = new[index]
                    for p in match:
                        if not any(p[0] == e[0] for e in existing):
                            existing.append(p)
                            lookup[p] = index
            if len(new) == len(matches):
                return new, passes
            matches = new

    def native(img, kp, ptr):
        n = len(ptr) - 1
        o_img, o_kp = np.empty_like(img), np.empty_like(kp)
        o_ptr = np.zeros(n + 1, np.int64)
        passes = np.zeros(1, np.int32)
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        nc = int(_lib.lib().iamx_link_matches(P(img), P(kp), P(ptr), n, P(o_img), P(o_kp), P(o_ptr), P(passes)))
        assert nc >= 0
        o_ptr = o_ptr[:nc + 1]
        return [[(int(o_img[j]), int(o_kp[j])) for j in range(o_ptr[c], o_ptr[c + 1])] for c in range(nc)], int(passes[0])

    used_incremental = 0
    for seed, n_img, n_kp, n_tracks in ((1, 30, 400, 3000), (2, 60, 150, 6000), (3, 12, 60, 900)):
        rng = np.random.default_rng(seed)
        pairs = []
        for _ in range(n_tracks):
            # a track seen by a run of images; now and then the "same" feature is matched through
            # ANOTHER keypoint of an image (a conflict the linking resolves over several passes)
            length = int(rng.integers(2, 8))
            start = int(rng.integers(0, n_img - length + 1))
            kps = rng.integers(0, n_kp, length)
            for a in range(length):
                for b in range(a + 1, min(a + 3, length)):
                    ka, kb = int(kps[a]), int(kps[b])
                    if rng.random() < 0.15:
                        kb = int(rng.integers(0, n_kp))
                    pairs.append((start + a, ka, start + b, kb))
        order = rng.permutation(len(pairs))
        img = np.array([[pairs[k][0], pairs[k][2]] for k in order], np.int32).ravel()
        kp = np.array([[pairs[k][1], pairs[k][3]] for k in order], np.int32).ravel()
        ptr = np.arange(len(pairs) + 1, dtype=np.int64) * 2
        want, want_passes = python_rules(ptr, img, kp)
        os.environ['IAMX_LINK_TIMING'] = '1'
        try:
            got, got_passes = native(img, kp, ptr)
            err = capfd.readouterr().err
            os.environ['IAMX_LINK_FULL'] = '1'
            full, full_passes = native(img, kp, ptr)
            capfd.readouterr()
        finally:
            os.environ.pop('IAMX_LINK_TIMING', None)
            os.environ.pop('IAMX_LINK_FULL', None)
        assert got == want and full == want, seed
        assert got_passes == want_passes == full_passes, (seed, got_passes, want_passes, full_passes)
        used_incremental += err.count('incremental over')
        assert want_passes >= 3
    assert used_incremental >= 2                          # the path under test did run
