"""find_matches against the reference's OWN pair loop (scripts/lib/matcher.py:852-1031), golden
G9 (oracle/gen_golden.py run_find_matches_case: the reference's lib.matcher.find_matches run with
the shims on a 14-image two-row strip).  What the golden holds and these tests compare, for
sort=True and sort=False and for a second call on the same project:
  * every image's match_list (the .match content), including the pair the "std >= 50 and < 100
    matches" rule empties (:1001-1005);
  * the whole /smart tree (per-pair surface / yaw entries, the weighted averages) -- the per-pair
    surface values depend on the camera poses the loop holds when it REACHES the pair, i.e. on the
    yaw-error feedback of every earlier pair (lib/matcher.py:990-993 -> lib/image.py:434-457);
  * every image's final aircraft quaternion, yaw_error_deg and camera pose.
CPU: the two device steps are replaced by oracle restatements INSIDE THE TEST (the schedule, the
pose feedback, the rank protocol and the bookkeeping are the product's); GPU: nothing is."""
import os
import pickle
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO

D2R = np.pi / 180.0
R2D = 180.0 / np.pi


def _golden():
    with open(os.path.join(GOLDEN, 'find_matches_strip.pkl'), 'rb') as f:
        return pickle.load(f)


def _pose_record(im):
    ac = im.node.getChild('aircraft_pose', True)
    cp = im.node.getChild('camera_pose', True)
    return dict(yaw_error_deg=ac.getFloat('yaw_error_deg') if ac.hasChild('yaw_error_deg') else None,
                aircraft_quat=[ac.getFloatEnum('quat', k) for k in range(4)],
                camera_ypr=[cp.getFloat('yaw_deg'), cp.getFloat('pitch_deg'), cp.getFloat('roll_deg')],
                camera_quat=[cp.getFloatEnum('quat', k) for k in range(4)])


def _tree(node):
    out = {}
    for k, v in node.__dict__.items():
        out[k] = _tree(v) if hasattr(v, 'getChild') else (list(v) if isinstance(v, list) else v)
    return out


def make_project(g, device):
    """the golden's project on the host stand-ins: REPORTED aircraft poses, camera pose = aircraft
    pose + mount (lib/pose.py:125-152), features attached"""
    from imageanalysis_amd import matcher, smart
    from imageanalysis_amd._deps import getNode
    from imageanalysis_amd.hostlib import camera, transforms as tf
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    from test_host_logic import _image
    for n_ in list(getNode('/images', True).__dict__):
        del getNode('/images', True).__dict__[n_]
    smart.smart_node.__dict__.clear()
    smart.load(None)                                  # (drops the module's caches of the tree)
    K = g['K']
    camera.set_K(K[0], K[4], K[2], K[5])
    camera.set_image_params(g['width'], g['height'])
    camera.set_mount_params(*g['mount'])
    matcher.detector_node.setString('detector', 'SIFT')
    matcher.detector_node.setFloat('scale', 0.4)
    matcher.matcher_node.setFloat('match_ratio', g['match_ratio'])
    matcher.matcher_node.setInt('min_pairs', g['min_pairs'])
    for key in ('schedule', 'min_dist', 'max_dist'):
        matcher.matcher_node.__dict__.pop(key, None)
    if device:
        matcher.the_matcher = None
        matcher.configure()
    else:
        matcher.max_distance, matcher.min_pairs = 270.0, float(g['min_pairs'])
        matcher.the_matcher = object()                # configure() needs the GPU
    proj = PoseProject(g['names'])
    body2cam = camera.get_body2cam()
    for i, im in enumerate(proj.image_list):
        rep = g['reported'][i]
        im.set_aircraft_pose(*g['aircraft_lla'], *rep['ypr'])
        ned2body = [im.node.getChild('aircraft_pose').getFloatEnum('quat', k) for k in range(4)]
        y, p, r = tf.euler_from_quaternion(tf.quaternion_multiply(ned2body, body2cam), 'rzyx')
        im.set_camera_pose(rep['ned'], y * R2D, p * R2D, r * R2D)
        f = _image(g['names'][i], g['des'][i], g['xy'][i])
        im.des_list, im.kp_list = f.des_list, f.kp_list
    return proj


def check_against(g, run, call, proj):
    from imageanalysis_amd import smart
    want = run['calls'][call]
    for i, im in enumerate(proj.image_list):
        got = {k: [list(map(int, p)) for p in v] for k, v in im.match_list.items()}
        assert got == want['match_lists'][i], (im.name, call)
    assert _tree(smart.smart_node) == want['smart'], call
    for i, im in enumerate(proj.image_list):
        assert _pose_record(im) == want['poses'][i], (im.name, call)


def _oracle_launch(batch, match_ratio, **kw):
    """TEST-ONLY stand-in of the device batch: oracle k=2 NN + threshold, the host filters, and
    the similarity fits of both directions (oracle/smart_oracle.py) turned into yaw values by the
    product's own arithmetic"""
    from imageanalysis_amd import matcher, smart
    from oracle import smart_oracle
    from test_dist_cpu import _oracle_match_batch
    out = []
    for (a, b), res in zip(batch, _oracle_match_batch(batch, match_ratio)):
        fwd = np.asarray(res[0], np.int64).reshape(-1, 2)
        fit = None
        if kw.get('surface'):
            xa, xb = matcher._kp_xy(a), matcher._kp_xy(b)
            fit = [float(smart._pair_distance(a, b)), None, None]
            if len(fwd):
                for side, (x, y, frm, to) in enumerate(((a, b, xb[fwd[:, 1]], xa[fwd[:, 0]]),
                                                        (b, a, xa[fwd[:, 0]], xb[fwd[:, 1]]))):
                    M = smart_oracle.fit_similarity(frm, to)
                    if M is not None:
                        fit[1 + side] = tuple(float(v) for v in smart.yaw_error_from_affine(x, y, M))
            fit = tuple(fit)
        out.append(res + (fit,))
    return out


def _oracle_surface(image_list, jobs, waiter=None):
    """TEST-ONLY stand-in of matcher._surface_device: numpy DLT with the job's per-pair matrices"""
    from imageanalysis_amd import matcher
    from imageanalysis_amd.hostlib import camera
    from oracle import smart_oracle
    K = camera.get_K()
    outs = []
    for job in jobs:
        z = np.zeros(int(job['m_off'][-1]))
        for t, (x, y) in enumerate(zip(job['pi'].tolist(), job['pj'].tolist())):
            a, b = int(job['m_off'][t]), int(job['m_off'][t + 1])
            pr = job['pairs'][a:b]
            xa, xb = matcher._kp_xy(image_list[x]), matcher._kp_xy(image_list[y])
            z[a:b] = smart_oracle.triangulate_down(job['proj'][t, 0], job['proj'][t, 1], K,
                                                   xa[pr[:, 0]], xb[pr[:, 1]])
        outs.append(z)
    return outs


def _run_cpu(rank, world, port, outdir, sort):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    if world > 1:
        from test_dist_cpu import _init
        _init(rank, world, port)
    from imageanalysis_amd import matcher
    g = _golden()
    proj = make_project(g, device=False)
    assert [_pose_record(im) for im in proj.image_list] == g['runs'][sort]['initial']
    matcher._launch_batch = _oracle_launch
    matcher._finish_batch = lambda handle: handle
    matcher._surface_device = _oracle_surface
    matcher.PAIRS_PER_BATCH = 5                       # several rounds, hits and quiet pairs mixed
    for call in range(2):
        matcher.find_matches(proj, None, strategy='traditional', transform='gms', sort=sort)
        if rank == 0:
            check_against(g, g['runs'][sort], call, proj)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    with open(os.path.join(outdir, 'ok_%d' % rank), 'w') as f:
        f.write('ok')


@pytest.fixture
def _restore():
    from imageanalysis_amd import matcher
    names = ('_launch_batch', '_finish_batch', '_surface_device', 'the_matcher', 'PAIRS_PER_BATCH',
             'max_distance', 'min_pairs')
    saved = {n: getattr(matcher, n) for n in names}
    yield
    for n, v in saved.items():
        setattr(matcher, n, v)


@pytest.mark.parametrize('sort', [True, False])
@pytest.mark.parametrize('native', [True, False])
def test_pair_loop_equals_reference_loop_host_logic(tmp_path, _restore, sort, native, monkeypatch):
    """native: the pose feedback replayed by libiamx's host routine (iamx_yaw_feedback_*) or by
    the python form that stands in for it when the tree names partners outside the project"""
    from imageanalysis_amd import smart
    monkeypatch.setattr(smart, 'NATIVE_FEEDBACK', native)
    _run_cpu(0, 1, 0, str(tmp_path), sort)


def test_pair_loop_with_a_periodic_save_after_every_round(tmp_path, _restore, monkeypatch):
    """the periodic save (scripts/lib/matcher.py:1008-1026: .match files of the dirty images,
    smart.json, the descriptor cache flush) in the middle of the loop changes nothing: every round
    is followed by one here"""
    from imageanalysis_amd import matcher
    monkeypatch.setattr(matcher, 'SAVE_INTERVAL', 0)
    _run_cpu(0, 1, 0, str(tmp_path), True)


@pytest.mark.parametrize('sort', [True, False])
def test_pair_loop_equals_reference_loop_world2_gloo(tmp_path, sort):
    import torch.multiprocessing as mp
    from test_dist_cpu import _free_port
    mp.spawn(_run_cpu, args=(2, _free_port(), str(tmp_path), sort), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), 'ok_0')) and os.path.exists(os.path.join(str(tmp_path), 'ok_1'))


@pytest.mark.gpu
@pytest.mark.parametrize('sort', [True, False])
@pytest.mark.parametrize('ppb', [16384, 7])
def test_find_matches_equals_reference_loop(sort, ppb, monkeypatch):
    """the shipped path end to end on the device: k=2 NN, filters, similarity fits, the pose
    feedback in schedule order, per-pair-pose triangulation, the discard rule -- one round
    (ppb 16384) and many small ones (ppb 7: the feedback crosses round boundaries)"""
    from imageanalysis_amd import matcher
    g = _golden()
    proj = make_project(g, device=True)
    assert [_pose_record(im) for im in proj.image_list] == g['runs'][sort]['initial']
    monkeypatch.setattr(matcher, 'PAIRS_PER_BATCH', ppb)
    for call in range(2):
        matcher.find_matches(proj, None, strategy='traditional', transform='gms', sort=sort)
        check_against(g, g['runs'][sort], call, proj)


def test_pose_feedback_native_replay_equals_python_replay_on_random_schedules():
    """smart.PoseFeedback: libiamx's iamx_yaw_feedback_* against the python form it stands in for,
    on random rounds -- pairs with matches and quiet pairs mixed, fits missing on either side,
    yaw errors beyond the 30 degree gate, pairs closer than 0.5 m, large weights, entries an
    earlier call left in the tree -- every estimate handed out and every final value bit-equal."""
    from imageanalysis_amd import smart
    from imageanalysis_amd._deps import getNode
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    rng = np.random.default_rng(2026)
    n_img = 40
    names = ['Z%03d' % i for i in rng.permutation(n_img)]      # (name order != index order)
    for n_ in names:
        getNode('/images', True).__dict__.pop(n_, None)
    smart.smart_node.__dict__.clear()
    smart.load(None)
    proj = PoseProject(names)
    # what an earlier call left: a few yaw_pairs entries, written the way the module writes them
    for _ in range(25):
        a, b = rng.choice(n_img, 2, replace=False)
        smart._record_yaw(proj.image_list[a], proj.image_list[b], float(rng.normal(0, 12)),
                          float(rng.uniform(0.2, 60)), float(rng.uniform(0, 360)), float(rng.uniform(0, 50)))
    fbs = [smart.PoseFeedback(proj.image_list, native=True), smart.PoseFeedback(proj.image_list, native=False)]
    assert fbs[0]._native is not None and fbs[1]._native is None
    seq0 = 0
    for rnd in range(30):
        n = int(rng.integers(1, 60))
        pi = rng.integers(0, n_img - 1, n).astype(np.int32)
        pj = (pi + 1 + rng.integers(0, n_img - 1 - pi)).astype(np.int32)
        quiet = rng.random(n) < 0.4
        hit_rows = np.nonzero(~quiet)[0].astype(np.int64)
        h = len(hit_rows)
        yv = lambda: np.stack([rng.normal(0, 14, h), rng.choice([0.3, 5.0, 40.0, 123.4], h),
                               rng.uniform(0, 360, h), np.abs(rng.normal(0, 30, h)) ** rng.choice([1, 3], h)], 1)
        yv_f, yv_r = yv(), yv()
        ok = rng.random((h, 2)) < 0.85
        seq = seq0 + np.arange(n, dtype=np.int64)
        seq0 += n
        outs = [fb.feed(seq, pi, pj, quiet, hit_rows, yv_f, yv_r, ok) for fb in fbs]
        for a, b in zip(outs[0], outs[1]):
            assert list(a) == list(b), rnd
    fbs[0]._sync_native()
    assert list(fbs[0].value) == [float(v) for v in fbs[1].value]
    assert list(fbs[0].touched) == list(fbs[1].touched)


@pytest.mark.gpu
def test_triangulate_packed_equals_the_per_image_form():
    """iamx_triangulate_packed (one pair of projection matrices per PAIR, packed match rows) ==
    iamx_triangulate_pairs (one matrix per image) bit for bit when the matrices are the images'
    own, through the product's surface-stage entry (matcher._surface_device); uploaded rows and
    several jobs in one call"""
    from imageanalysis_amd import matcher, smart
    g = _golden()
    proj = make_project(g, device=True)
    il = proj.image_list
    rng = np.random.default_rng(5)
    jobs, want = [], []
    for pairs_of_job in (((0, 1), (1, 2), (0, 3)), ((5, 6),)):
        pi, pj, P, off, rows = [], [], [], [0], []
        for a, b in pairs_of_job:
            m = int(rng.integers(30, 400))
            pr = np.stack([rng.integers(0, len(il[a].kp_list), m), rng.integers(0, len(il[b].kp_list), m)], 1).astype(np.int32)
            want.append(smart.triangulate_down(il[a], il[b], pr))
            pi.append(a); pj.append(b); rows.append(pr); off.append(off[-1] + m)
            P.append(np.stack([smart.projection_matrix(il[a]).ravel(), smart.projection_matrix(il[b]).ravel()]))
        jobs.append(dict(pi=np.array(pi), pj=np.array(pj), proj=np.stack(P), m_off=np.array(off, np.int64),
                         pairs=np.concatenate(rows), src=None))
    # ... and a third job whose rows are read where an RCCL gather would have left them: the wire
    # form of a rank's share (matcher._Part.to_wire) on the DEVICE, rows addressed inside it
    import torch
    j0 = jobs[0]
    h = len(j0['pi'])
    R = matcher._RoundResult(h)
    R.n_fwd = R.n_rev = R.cc = np.zeros(h, np.int64)
    R.quiet = np.zeros(h, bool)
    R.hit_rows = np.arange(h, dtype=np.int64)
    R.lo, R.hi = j0['m_off'][:-1].copy(), j0['m_off'][1:].copy()
    R.fwd_all = j0['pairs']
    R.fit = True
    R.dist, R.same = np.ones(h), np.zeros(h, bool)
    R.yv_f = R.yv_r = np.zeros((h, 4))
    R.aff_ok = np.ones((h, 2), bool)
    part = matcher._Part(np.arange(h, dtype=np.int64), j0['pi'].astype(np.int32), j0['pj'].astype(np.int32),
                         np.zeros(h), np.zeros(h, np.int64), np.zeros(h, np.int64), R)
    wire = part.to_wire()
    back = matcher._Part.from_wire(wire, torch.from_numpy(wire).cuda())
    assert back.R.src['kind'] == 'device' and np.array_equal(back.R.src['pairs'].cpu().numpy(), j0['pairs'])
    assert np.array_equal(back.R.fwd_all, j0['pairs']) and np.array_equal(back.R.lo, R.lo)
    jobs.append(dict(pi=j0['pi'], pj=j0['pj'], proj=j0['proj'], m_off=j0['m_off'], pairs=None, src=back.R.src))
    want += want[:h]
    got = matcher._surface_device(il, jobs)
    k = 0
    for job, z in zip(jobs, got):
        for t in range(len(job['pi'])):
            assert np.array_equal(z[job['m_off'][t]:job['m_off'][t + 1]], want[k]), k
            k += 1
    assert k == 7


def _run_gpu_rank(rank, world, port, outdir, sort):
    """one of two ranks on the SAME GPU over gloo (RCCL refuses two ranks per device; only the
    collectives differ): the shipped device path on every rank, rank 0 books and runs the surface
    stage for both"""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import torch
    from test_dist_cpu import _init
    torch.cuda.set_device(0)
    _init(rank, world, port)
    from imageanalysis_amd import matcher
    g = _golden()
    proj = make_project(g, device=True)
    matcher.PAIRS_PER_BATCH = 4                       # several rounds, both ranks busy in each
    for call in range(2):
        matcher.find_matches(proj, None, strategy='traditional', transform='gms', sort=sort)
        if rank == 0:
            check_against(g, g['runs'][sort], call, proj)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
    with open(os.path.join(outdir, 'ok_%d' % rank), 'w') as f:
        f.write('ok')


@pytest.mark.gpu
@pytest.mark.parametrize('sort', [True, False])
def test_find_matches_two_ranks_on_one_gpu_equals_reference_loop(tmp_path, sort):
    """the N > 1 product path on the device: sharded feature exchange, round-robin deal, one byte
    tensor per rank and round to rank 0, rank 0's pose feedback + triangulation of BOTH ranks'
    pairs -- against the reference's own loop (G9)"""
    import torch.multiprocessing as mp
    from test_dist_cpu import _free_port
    mp.spawn(_run_gpu_rank, args=(2, _free_port(), str(tmp_path), sort), nprocs=2, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), 'ok_0')) and os.path.exists(os.path.join(str(tmp_path), 'ok_1'))
