"""CPU, world_size 2 over gloo: the N>1 sharding / exchange logic (no kernels run here; the
device step of find_matches is replaced by the oracle INSIDE THE TEST ONLY)."""
import os
import pickle
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO


@pytest.fixture(autouse=True)
def _restore_matcher_injections():
    """the helpers below replace the device halves of find_matches (and _deps.smart) by assignment
    -- they also run in spawned workers, where monkeypatch is not at hand; the world-1 runs happen
    in THIS process, so what they replaced is put back for the tests that follow"""
    from imageanalysis_amd import _deps, matcher
    names = ('_launch_batch', '_finish_batch', 'the_matcher', 'PAIRS_PER_BATCH', 'max_distance', 'min_pairs')
    saved = {n: getattr(matcher, n) for n in names}
    smart = _deps.smart
    yield
    for n, v in saved.items():
        setattr(matcher, n, v)
    _deps.smart = smart


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    return dist


def _strip(n_img=6, seed=5):
    from test_match_gpu import _sift_like
    rng = np.random.default_rng(seed)
    W, H = 5472, 3648
    des, xy = [], []
    for i in range(n_img):
        n = int(rng.integers(300, 500))
        d = _sift_like(rng, n)
        p = np.stack([rng.uniform(600, W - 600, n), rng.uniform(400, H - 400, n)], 1)
        if i:
            k = min(int(0.5 * n), len(des[i - 1]))
            src, dst = rng.permutation(len(des[i - 1]))[:k], rng.permutation(n)[:k]
            d[dst] = np.clip(des[i - 1][src].astype(int) + rng.integers(-5, 6, (k, 128)), 0, 255)
            p[dst] = xy[i - 1][src] + [250.0, -120.0] + rng.normal(0, 0.6, (k, 2))
        des.append(d)
        xy.append(np.clip(p, 0, [W - 1, H - 1]).astype(np.float32))
    return des, xy


def _oracle_match_batch(batch, match_ratio):
    """stand-in for matcher._match_batch (device) used only by this CPU test: oracle k=2 NN +
    threshold, then the host filters; same return format (fwd, rev, n_fwd, n_rev)"""
    from imageanalysis_amd import matcher
    from oracle import match_oracle as mo
    out = []
    for a, b in batch:
        res = []
        for q, t in ((a, b), (b, a)):
            idx, d2 = mo.knn2_l2(q.des_list, t.des_list)
            dist = mo.distances_f32(d2).astype(np.float64)
            metric = dist[:, 0] * (dist[:, 0] / dist[:, 1])
            keep = np.nonzero(metric < matcher.max_distance * match_ratio)[0]
            res.append((matcher._threshold_sort_clip(keep.astype(np.int32), idx[keep, 0],
                                                     metric[keep]), len(keep)))
        fwd = matcher._post_filter(a, b, res[0][0])
        rev = matcher._post_filter(b, a, res[1][0]) if len(fwd) >= matcher.min_pairs else []
        fwd, rev = matcher.filter_cross_check(fwd, rev)
        out.append((fwd, rev, res[0][1], res[1][1]))
    return out


def _run_find_matches(rank, world, port, outdir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    if world > 1:
        _init(rank, world, port)
    from imageanalysis_amd import matcher
    from imageanalysis_amd.hostlib import camera
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    from test_host_logic import _image
    des, xy = _strip()
    names = ['D%02d' % i for i in range(len(des))]
    proj = PoseProject(names)
    for i, im in enumerate(proj.image_list):
        im.set_camera_pose([-3.2727 * i, -6.8182 * i, -100.0], 0.0, -90.0, 0.0)
        f = _image(names[i], des[i], xy[i])
        im.des_list, im.kp_list = f.des_list, f.kp_list
    matcher.detector_node.setString('detector', 'SIFT')
    matcher.detector_node.setFloat('scale', 0.4)
    matcher.matcher_node.setFloat('match_ratio', 0.75)
    matcher.matcher_node.setInt('min_pairs', 25)
    camera.set_image_params(5472, 3648)
    matcher.max_distance, matcher.min_pairs = 270.0, 25.0
    matcher.the_matcher = object()                  # configure() would need the GPU library
    # TEST-ONLY injection: the two halves of the device batch (launch / finish)
    matcher._launch_batch = lambda batch, ratio, **kw: _oracle_match_batch(batch, ratio)
    matcher._finish_batch = lambda handle: handle
    matcher._deps.smart = lambda: None              # the pairwise surface estimate is a GPU kernel
    matcher.PAIRS_PER_BATCH = 3                     # several rounds
    matcher.find_matches(proj, None, strategy='traditional', sort=True)
    with open(os.path.join(outdir, 'r%d_of_%d.pkl' % (rank, world)), 'wb') as f:
        pickle.dump({im.name: im.match_list for im in proj.image_list}, f)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def test_find_matches_two_ranks_equals_one_rank():
    with tempfile.TemporaryDirectory() as d:
        _run_find_matches(0, 1, 0, d)
        mp.spawn(_run_find_matches, args=(2, _free_port(), d), nprocs=2, join=True)
        one = pickle.load(open(os.path.join(d, 'r0_of_1.pkl'), 'rb'))
        # rank 0 holds the whole survey's bookkeeping; the other rank is a worker (its share of every
        # round travels to rank 0 as one byte tensor, nothing is booked there)
        two = pickle.load(open(os.path.join(d, 'r0_of_2.pkl'), 'rb'))
        assert two == one
        part = pickle.load(open(os.path.join(d, 'r1_of_2.pkl'), 'rb'))
        assert all(len(ml) == 0 for ml in part.values())


def _run_helpers(rank, world, port):
    sys.path.insert(0, REPO)
    dist = _init(rank, world, port)
    from imageanalysis_amd import dist as D
    assert D.world() == (rank, world)
    # ragged store gather: images of 3, 1, 4, 2 "rows"
    rows = np.array([3, 1, 4, 2])
    off = np.concatenate([[0], np.cumsum(rows)])
    owner = D.owner_of_images(4, world)
    assert owner.tolist() == [0, 0, 1, 1]
    desc = torch.zeros(int(off[-1]) * 128, dtype=torch.int8)
    norm = torch.zeros(int(off[-1]), dtype=torch.int32)
    for i in np.nonzero(owner == rank)[0]:
        desc[off[i] * 128:off[i + 1] * 128] = i + 1
        norm[off[i]:off[i + 1]] = 100 + i
    D.gather_store_shards([(desc, 128), (norm, 1)], off, owner, rank, world)
    want = np.repeat(np.arange(4) + 1, rows)
    assert np.array_equal(desc.view(-1, 128)[:, 0].numpy(), want)
    assert np.array_equal(norm.numpy(), 100 + np.repeat(np.arange(4), rows))
    # objects
    got = D.allgather_objects({'rank': rank, 'pairs': [[rank, 1]]})
    assert [g['rank'] for g in got] == [0, 1]
    # per-round results: everything on rank 0, own part elsewhere; a failure on ONE rank is
    # re-raised on every rank instead of leaving the others in the collective
    got = D.gather_results([('pair', rank)])
    assert got == ([[('pair', 0)], [('pair', 1)]] if rank == 0 else [[('pair', 1)]])
    with pytest.raises(ZeroDivisionError):
        D.gather_results([], ZeroDivisionError("float division by zero") if rank == 1 else None)
    with pytest.raises((RuntimeError, ValueError)):
        D.gather_results([], ValueError("bad image") if rank == 0 else None)
    # BA: observation sharding by point + all-reduce of camera-side sums == unsharded sums
    rng = np.random.default_rng(3)
    P, C, O = 50, 6, 400
    pt = rng.integers(0, P, O)
    cam = np.sort(rng.integers(0, C, O))
    val = rng.normal(size=O)
    mine = D.shard_observations_by_point(pt, P, rank, world)
    assert np.all(np.diff(mine) > 0)
    acc = torch.zeros(C, dtype=torch.float64)
    acc.index_add_(0, torch.from_numpy(cam[mine]), torch.from_numpy(val[mine]))
    D.allreduce_sum_(acc)
    full = np.bincount(cam, weights=val, minlength=C)
    assert np.allclose(acc.numpy(), full, atol=1e-12)
    owners = D.allgather_objects(sorted(set(pt[mine].tolist())))
    assert not (set(owners[0]) & set(owners[1]))          # a point lives on exactly one rank
    assert sum(len(D.shard_observations_by_point(pt, P, r, world)) for r in range(world)) == O
    dist.barrier()
    dist.destroy_process_group()


def test_dist_helpers_world2_gloo():
    mp.spawn(_run_helpers, args=(2, _free_port()), nprocs=2, join=True)


def test_shard_pairs_partition():
    from imageanalysis_amd import dist as D
    pairs = [(i, j) for j in range(9) for i in range(j)]
    for ws in (1, 2, 3, 8):
        parts = [D.shard_pairs(pairs, r, ws) for r in range(ws)]
        assert sum(parts, []) == pairs
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _run_sharded_detect(rank, world, port, outdir):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    _init(rank, world, port)
    from imageanalysis_amd import matcher
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    from test_host_logic import _image
    des, xy = _strip(n_img=7, seed=9)
    names = ['E%02d' % i for i in range(len(des))]
    proj = PoseProject(names)
    calls = []

    def fake_detect(self, scale, use_cache=True):           # TEST-ONLY stand-in for the GPU detector
        i = names.index(self.name)
        calls.append(i)
        f = _image(self.name, des[i], xy[i])
        self.des_list, self.kp_list = f.des_list, f.kp_list
    for im in proj.image_list:
        im.detect_features = fake_detect.__get__(im)
        im.des_list, im.kp_list = None, None
    matcher.detect_scale = 0.4
    matcher.the_matcher = dm = matcher.DeviceMatcher()      # adopt() needs no GPU
    counts = matcher.detect_features_sharded(proj)
    assert counts.tolist() == [len(d) for d in des]
    from imageanalysis_amd import dist as D
    owner = D.owner_of_images(len(des), world)
    assert sorted(calls) == np.nonzero(owner == rank)[0].tolist()      # detected once, by its owner
    for i, im in enumerate(proj.image_list):
        if owner[i] == rank:
            assert im.name not in dm._adopted and len(im.des_list) == len(des[i])
            continue
        assert im.des_list is None and matcher._have_features(im) and matcher._rows_of(im) == len(des[i])
        slot = dm.slot_of(im)
        got = [d for s_, d in dm._pending if s_ == slot][0]
        assert np.array_equal(got.numpy(), des[i])                      # uint8 rows of the owner
        assert np.array_equal(dm._kp[slot][0], xy[i])
        assert np.array_equal(dm._kp[slot][1], matcher.kp_key2(xy[i]))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_detection_and_feature_exchange_world2_gloo(tmp_path):
    """SURVEY.md 8e 'SIFT detect': each image is detected by one rank, descriptors + keypoint
    positions reach the other rank's device matcher through dist.exchange_features."""
    mp.spawn(_run_sharded_detect, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def _run_find_matches_again(rank, world, port, outdir):
    """a SECOND find_matches call in the same process (results live on rank 0 only): the ranks
    must still agree on the work list -- rank 0's -- or their collectives no longer line up"""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    _init(rank, world, port)
    from imageanalysis_amd import matcher
    from imageanalysis_amd.hostlib import camera
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    from test_host_logic import _image
    des, xy = _strip()
    names = ['D%02d' % i for i in range(len(des))]
    proj = PoseProject(names)
    for i, im in enumerate(proj.image_list):
        im.set_camera_pose([-3.2727 * i, -6.8182 * i, -100.0], 0.0, -90.0, 0.0)
        f = _image(names[i], des[i], xy[i])
        im.des_list, im.kp_list = f.des_list, f.kp_list
    matcher.detector_node.setString('detector', 'SIFT')
    matcher.detector_node.setFloat('scale', 0.4)
    matcher.matcher_node.setFloat('match_ratio', 0.75)
    matcher.matcher_node.setInt('min_pairs', 25)
    camera.set_image_params(5472, 3648)
    matcher.max_distance, matcher.min_pairs = 270.0, 25.0
    matcher.the_matcher = object()
    calls = []

    def launch(batch, ratio, **kw):
        calls.append(len(batch))
        return _oracle_match_batch(batch, ratio)
    matcher._launch_batch = launch
    matcher._finish_batch = lambda handle: handle
    matcher._deps.smart = lambda: None
    matcher.PAIRS_PER_BATCH = 3
    matcher.find_matches(proj, None, strategy='traditional', sort=True)
    first = {im.name: {k: list(map(list, v)) for k, v in im.match_list.items()}
             for im in proj.image_list}
    n_first = sum(calls)
    del calls[:]
    matcher.find_matches(proj, None, strategy='traditional', sort=True)       # must not hang
    # the second call only retries the pairs rank 0 recorded as empty, dealt over both ranks
    retried = sum(1 for k, im in enumerate(proj.image_list) for o, v in im.match_list.items()
                  if len(v) == 0) // 2 if rank == 0 else None
    import torch.distributed as dist
    from imageanalysis_amd import dist as D
    total = sum(D.allgather_objects(sum(calls)))
    n_retry = D.broadcast_object(retried, src=0)
    assert total == n_retry and total < n_first * world
    if rank == 0:
        again = {im.name: {k: list(map(list, v)) for k, v in im.match_list.items()}
                 for im in proj.image_list}
        assert again == first
    dist.barrier()
    dist.destroy_process_group()


def test_find_matches_second_call_same_process_world2_gloo(tmp_path):
    mp.spawn(_run_find_matches_again, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def _run_round0_failure(rank, world, port, where):
    """an exception raised on ONE rank before the first gather (launch of round 0, the sharded
    detection) is re-raised on every rank instead of leaving the other in a collective"""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    _init(rank, world, port)
    from imageanalysis_amd import matcher
    from imageanalysis_amd.hostlib import camera
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    from test_host_logic import _image
    des, xy = _strip()
    names = ['D%02d' % i for i in range(len(des))]
    proj = PoseProject(names)
    for i, im in enumerate(proj.image_list):
        im.set_camera_pose([-3.2727 * i, -6.8182 * i, -100.0], 0.0, -90.0, 0.0)
        f = _image(names[i], des[i], xy[i])
        im.des_list, im.kp_list = f.des_list, f.kp_list
    matcher.detector_node.setString('detector', 'SIFT')
    matcher.detector_node.setFloat('scale', 0.4)
    matcher.matcher_node.setFloat('match_ratio', 0.75)
    matcher.matcher_node.setInt('min_pairs', 25)
    camera.set_image_params(5472, 3648)
    matcher.max_distance, matcher.min_pairs = 270.0, 25.0
    matcher._deps.smart = lambda: None
    matcher.PAIRS_PER_BATCH = 64                       # one round: the failure is in round 0
    if where == 'launch':
        matcher.the_matcher = object()

        def launch(batch, ratio, **kw):
            if rank == 1:
                raise ZeroDivisionError("float division by zero")
            return _oracle_match_batch(batch, ratio)
        matcher._launch_batch = launch
        matcher._finish_batch = lambda handle: handle
        with pytest.raises(ZeroDivisionError):
            matcher.find_matches(proj, None, strategy='traditional', sort=True)
    else:
        matcher.the_matcher = matcher.DeviceMatcher()

        def bad_detect(self, scale, use_cache=True):
            raise SystemExit("image size mismatch")
        for k, im in enumerate(proj.image_list):
            im.des_list, im.kp_list = None, None
            if k >= 3:                                 # the second rank's images
                im.detect_features = bad_detect.__get__(im)
            else:
                def ok(self, scale, use_cache=True, k=k):
                    f = _image(names[k], des[k], xy[k])
                    self.des_list, self.kp_list = f.des_list, f.kp_list
                im.detect_features = ok.__get__(im)
        with pytest.raises(SystemExit):
            matcher.detect_features_sharded(proj)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('where', ['launch', 'detect'])
def test_failure_before_first_gather_reaches_every_rank_world2_gloo(where):
    mp.spawn(_run_round0_failure, args=(2, _free_port(), where), nprocs=2, join=True)


def test_round_robin_deal_balances_the_overlapping_pairs():
    """find_matches deals every round's pairs round-robin (dist.round_slice).  On a
    distance-sorted all-pairs schedule the overlapping pairs -- all the exact-stage, filter and
    match-list work -- come first: contiguous blocks (what rounds 1-5 did) give all of them to
    rank 0, the round-robin deal gives every rank the same share to within a few per cent."""
    from imageanalysis_amd import dist as D
    from imageanalysis_amd import matcher
    from imageanalysis_amd.hostlib.image_pose import PoseProject
    n = 20 * 24
    proj = PoseProject(['G%03d' % i for i in range(n)])
    for i, im in enumerate(proj.image_list):
        r, c = divmod(i, 24)
        im.set_camera_pose([30.0 * r, 25.0 * (c if r % 2 == 0 else 23 - c), -100.0], 0.0, -90.0, 0.0)
    matcher.matcher_node.setString('schedule', 'all-pairs')
    try:
        wd, wi, wj = matcher._work_arrays(proj, True)
    finally:
        matcher.matcher_node.__dict__.pop('schedule', None)
    assert len(wd) == n * (n - 1) // 2 and np.all(np.diff(wd) >= 0)
    near = wd <= 70.0                                        # pairs whose footprints overlap
    world, ppb = 8, 512
    n_rounds = (len(wd) + world * ppb - 1) // (world * ppb)
    share = np.zeros(world, np.int64)
    seen = np.zeros(len(wd), np.int64)
    for rank in range(world):
        for rnd in range(n_rounds):
            sl = D.round_slice(len(wd), rnd, ppb, rank, world)
            share[rank] += int(near[sl].sum())
            seen[sl] += 1
    assert np.all(seen == 1)                                 # a partition of the schedule
    assert share.sum() == near.sum() > 1000
    assert share.max() - share.min() <= 0.15 * share.mean()
    blocks = np.array([near[D.shard_bounds(len(wd), r, world)[0]:D.shard_bounds(len(wd), r, world)[1]].sum()
                       for r in range(world)])
    assert blocks[0] == near.sum() and blocks[1:].sum() == 0   # what the contiguous deal did
