"""GPU: the shipped matching forms against the oracle -- the symmetric sweep (one MFMA pass for
both directions of an image pair + exact re-scan of the candidate rows) in all three workgroup
shapes, and the one-direction fast forms at the sizes that select their 512- and 1024-row
instantiations.  Integer work -> bit-exact: survivor rows, train rows, metrics, squared
distances of the survivors (scripts/lib/matcher.py:203-216,253-269)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

MATCH_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'match_*.npz')))


def _sift_like(rng, n):
    g = rng.gamma(0.6, 1.0, size=(n, 128))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    g = np.minimum(g, 0.2)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    return np.clip(np.rint(g * 512.0), 0, 255).astype(np.uint8)


def _run(store, pairs, thresh, fast=True, sym=None):
    import torch
    from imageanalysis_amd import kernels
    pb = kernels.PairBatch(store, np.asarray(pairs, np.int32), sym=sym)
    ws = kernels.PairWorkspace(pb.rows, pb.n_pairs)
    pb.run(ws, thresh, fast=fast)
    torch.cuda.synchronize()
    first, count, q, t, m = ws.survivors(pb.n_pairs)
    take = np.concatenate([np.arange(f, f + c) for f, c in zip(first, count)]
                          + [np.zeros(0, np.int64)]).astype(np.int64)
    soff = np.zeros(pb.n_pairs + 1, np.int64)
    np.cumsum(count, out=soff[1:])
    return dict(d2=ws.d2[:pb.rows].cpu().numpy(), soff=soff, sq=q[take], st=t[take], sm=m[take],
                unresolved=int(ws.unresolved.item()), zero_div=int(ws.zero_div.item()),
                off=pb.out_off, pb=pb)


def _oracle_survivors(q, t, thresh):
    from oracle import cpu_ref
    ridx, rd2 = cpu_ref.knn2_l2_u8(q, t)
    d = np.sqrt(rd2.astype(np.float32)).astype(np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        metric = d[:, 0] * (d[:, 0] / d[:, 1])
    keep = np.nonzero(metric < thresh)[0]              # NaN (d1 == 0) compares false
    return keep, ridx[keep, 0], metric[keep], rd2[keep], int((rd2[:, 1] == 0).sum())


def _same_survivors(a, b):
    assert a['zero_div'] == b['zero_div']
    for k in ('soff', 'sq', 'st', 'sm'):
        assert np.array_equal(a[k], b[k]), k
    # squared distances of the survivors
    for p in range(len(a['soff']) - 1):
        rows = a['off'][p] + a['sq'][a['soff'][p]:a['soff'][p + 1]]
        assert np.array_equal(a['d2'][rows], b['d2'][rows]), p


def test_desc3_store_layout():
    """sorted store of the symmetric sweep: stable sort by |a-128|^2, padding, inverse maps,
    C-operand terms"""
    from imageanalysis_amd import kernels
    rng = np.random.default_rng(4)
    for n in (2, 127, 128, 129, 1000, 4096, 5000):
        a = _sift_like(rng, n)
        if n >= 1000:
            a[500:600] = a[100:200]                   # equal keys: the sort must be stable
        st = kernels.DescriptorStore.from_arrays([a])
        cap = -(-n // 128) * 128
        assert int(st.offsets3[1]) == cap
        s = a.astype(np.int64) - 128
        n2 = (s * s).sum(1)
        nb = n2 + 2 * s.sum(1)
        order = np.argsort(n2, kind='stable')
        perm = st.sperm.cpu().numpy()[:cap]
        assert np.array_equal(perm[:n], order) and (perm[n:] == -1).all()
        inv = st.sinv.cpu().numpy()[:cap]
        assert np.array_equal(inv[:n][order], np.arange(n)) and (inv[n:] == -1).all()
        rows = st.desc3.cpu().numpy()[:cap].astype(np.int64)
        assert np.array_equal(rows[:n], s[order]) and (rows[n:] == 0).all()
        assert np.array_equal(st.sn2.cpu().numpy()[:n], n2[order])
        sct = st.sct.cpu().numpy()[:cap]
        assert np.array_equal(sct[:n], nb[order] >> 1) and (sct[n:] == 0x3F000000).all()
    # float32 input (what the reference hands over) packs like uint8
    a = _sift_like(rng, 300)
    s1 = kernels.DescriptorStore.from_arrays([a])
    s2 = kernels.DescriptorStore.from_arrays([a.astype(np.float32)])
    for name in ('desc3', 'sn2', 'sct', 'sperm', 'sinv'):
        assert np.array_equal(getattr(s1, name).cpu().numpy(), getattr(s2, name).cpu().numpy()), name


@pytest.mark.parametrize('path', MATCH_CASES, ids=os.path.basename)
def test_sym_equals_general_path_golden(path):
    from imageanalysis_amd import kernels
    g = np.load(path)
    store = kernels.DescriptorStore.from_arrays([g['des1'], g['des2']])
    thresh = 270.0 * float(g['match_ratio'])
    a = _run(store, [[0, 1], [1, 0]], thresh, sym=True)
    b = _run(store, [[0, 1], [1, 0]], thresh, fast=False, sym=False)
    assert a['pb'].sym and not b['pb'].sym
    _same_survivors(a, b)
    # and == the reference's pre-GMS list after the host's stable sort + clip
    for p, tag in enumerate(['fwd', 'rev']):
        lo, hi = a['soff'][p], a['soff'][p + 1]
        order = np.argsort(a['sm'][lo:hi], kind='stable')[:2000]
        got = np.stack([a['sq'][lo:hi][order], a['st'][lo:hi][order]], 1)
        if len(g['pregms_%s' % tag]):
            assert np.array_equal(got, g['pregms_%s' % tag])


def _survey(rng, sizes, overlap=0.3):
    imgs = [_sift_like(rng, n) for n in sizes]
    for k in range(1, len(imgs)):                    # neighbours overlap: real survivors
        m = int(min(len(imgs[k]), len(imgs[k - 1])) * overlap)
        src = rng.permutation(len(imgs[k - 1]))[:m]
        dst = rng.permutation(len(imgs[k]))[:m]
        imgs[k][dst] = np.clip(imgs[k - 1][src].astype(int) + rng.integers(-6, 7, (m, 128)), 0, 255)
    return imgs


@pytest.mark.parametrize('sizes,form', [
    ([4096, 4097, 5000, 4224, 4608, 4096], 2),       # 1024-row workgroups (BASELINE configs[1] shape)
    ([2048, 2049, 3000, 4095, 2500], 1),             # 512-row workgroups
    ([300, 2, 257, 1024, 129, 640, 3], 0),           # 256-row workgroups, ragged and tiny images
], ids=['rows1024', 'rows512', 'rows256'])
def test_sym_forms_vs_general_kernel_and_oracle(sizes, form):
    """every ordered pair of a small survey: survivors of the symmetric path == the general
    kernel's; a sample of pairs == oracle/cpu_ref.c.  Planted: near copies between neighbours,
    exact duplicates inside an image (lowest train row must win), duplicates of the best (second
    == best), identical rows across images (distance 0 -> ZeroDivisionError count), an image
    whose rows all share one parity of |a-128|^2."""
    from imageanalysis_amd import kernels
    rng = np.random.default_rng(100 + form)
    imgs = _survey(rng, sizes)
    big = int(np.argmax(sizes))
    n = sizes[big]
    if n >= 600:
        imgs[big][n // 2:n // 2 + 200] = imgs[big][:200]          # duplicates 'n/2' rows apart
        other = (big + 2) % len(sizes)
        m = min(100, sizes[other])
        imgs[other][:m] = imgs[big][300:300 + m]                   # identical rows across images
    if form == 0:
        imgs[1][:] = 128                                           # two identical rows, one parity
    else:
        par = imgs[0].astype(np.int64).sum(1) & 1                  # parity of |a-128|^2 = parity of sum
        imgs[0][:, 5] = np.where(par == 1, imgs[0][:, 5] ^ 1, imgs[0][:, 5])
        assert (((imgs[0].astype(np.int64) - 128) ** 2).sum(1) & 1).max() == 0
    store = kernels.DescriptorStore.from_arrays(imgs)
    pairs = [(i, j) for i in range(len(sizes)) for j in range(len(sizes)) if i != j]
    for thresh in (270.0 * 0.75, 1e9):
        if thresh > 1e6 and form == 2:
            sub = [p for p in pairs if p[0] < 3 and p[1] < 3]      # all rows are candidates here
        else:
            sub = pairs
        a = _run(store, sub, thresh, sym=True)
        assert a['pb'].sym and a['pb'].sym_form == form and a['unresolved'] == 0
        b = _run(store, sub, thresh, fast=False, sym=False)
        _same_survivors(a, b)
        if thresh < 1e6:
            assert (np.diff(a['soff']) > 50).sum() >= 2            # the planted overlaps survive
        # oracle on a sample of ordered pairs (both roles of the sweep: a pair and its mirror)
        sample = [0, 1, len(sub) // 2, len(sub) - 1]
        sample += [sub.index((j, i)) for (i, j) in [sub[k] for k in sample]]
        for p in sorted(set(sample)):
            i, j = sub[p]
            keep, tr, mt, d2, zd = _oracle_survivors(imgs[i], imgs[j], thresh)
            lo, hi = a['soff'][p], a['soff'][p + 1]
            assert np.array_equal(a['sq'][lo:hi], keep), (i, j)
            assert np.array_equal(a['st'][lo:hi], tr), (i, j)
            assert np.array_equal(a['sm'][lo:hi], mt), (i, j)
            assert np.array_equal(a['d2'][a['off'][p] + keep], d2), (i, j)


def _narrow_pairs(ws, n_pairs):
    """ordered pairs of the last batch whose candidates went through the narrow exact stage
    (include/iamx.h: the int32 table behind the 256 control bytes of the narrow workspace)"""
    return int((ws.nar[256:256 + 4 * n_pairs].view(__import__('torch').int32) >= 0).sum().item())


@pytest.mark.parametrize('sizes,form', [
    ([6000, 5500, 4096, 9000], 2),
    ([2500, 3000, 4000, 2048], 1),
    ([1024, 1500, 2047, 1100], 0),
], ids=['rows1024', 'rows512', 'rows256'])
def test_narrow_exact_stage_equals_full_scan(sizes, form, monkeypatch):
    """Overlapping images (hundreds to thousands of candidates per ordered pair): the narrow exact
    stage -- one class of train rows per task, chosen by the sweep's group minima / block bounds --
    must leave what the full scan leaves (IAMX_EXACT_NARROW=0), what the general kernel leaves,
    and what oracle/cpu_ref.c says.  Planted: noisy copies between ALL images (best and second in
    arbitrary classes), exact duplicates inside an image far apart in the sorted order's terms
    (ties on the best: lowest ORIGINAL row wins; second == best), triples of identical rows
    (three-way ties inside one lane's rows), identical rows across images (zero distances)."""
    import torch
    from imageanalysis_amd import kernels
    rng = np.random.default_rng(300 + form)
    imgs = [_sift_like(rng, n) for n in sizes]
    base = imgs[0]
    for k in range(1, len(imgs)):
        m = int(min(len(imgs[k]), len(base)) * 0.5)
        src = rng.permutation(len(base))[:m]
        dst = rng.permutation(len(imgs[k]))[:m]
        imgs[k][dst] = np.clip(base[src].astype(int) + rng.integers(-5, 6, (m, 128)), 0, 255)
    n1 = sizes[1]
    imgs[1][n1 - 150:n1 - 50] = imgs[1][10:110]                    # duplicates: ties on the best
    imgs[1][n1 - 50:n1 - 25] = imgs[1][200:225]                    # triples
    imgs[1][n1 - 25:n1] = imgs[1][200:225]
    imgs[2][:60] = imgs[3][500:560]                                # zero distances across images
    store = kernels.DescriptorStore.from_arrays(imgs)
    pairs = [(i, j) for i in range(len(sizes)) for j in range(len(sizes)) if i != j]
    thresh = 270.0 * 0.75
    pb = kernels.PairBatch(store, np.asarray(pairs, np.int32), sym=True)
    ws = kernels.PairWorkspace(pb.rows, pb.n_pairs)
    pb.run(ws, thresh)
    torch.cuda.synchronize()
    assert pb.sym_form == form
    assert _narrow_pairs(ws, pb.n_pairs) == len(pairs)            # every pair has > 64 candidates
    assert int(ws.nar[:16].view(torch.int32).abs().sum().item()) == 0    # control words back to 0
    a = _run(store, pairs, thresh, sym=True)
    assert a['unresolved'] == 0
    monkeypatch.setenv('IAMX_EXACT_NARROW', '0')
    f = _run(store, pairs, thresh, sym=True)
    monkeypatch.delenv('IAMX_EXACT_NARROW')
    b = _run(store, pairs, thresh, fast=False, sym=False)
    _same_survivors(a, f)
    _same_survivors(a, b)
    assert np.diff(a['soff']).min() > 64
    for p in (0, 3, 4, 7, len(pairs) - 1):
        i, j = pairs[p]
        keep, tr, mt, d2, zd = _oracle_survivors(imgs[i], imgs[j], thresh)
        lo, hi = a['soff'][p], a['soff'][p + 1]
        assert np.array_equal(a['sq'][lo:hi], keep), (i, j)
        assert np.array_equal(a['st'][lo:hi], tr), (i, j)
        assert np.array_equal(a['sm'][lo:hi], mt), (i, j)
        assert np.array_equal(a['d2'][a['off'][p] + keep], d2), (i, j)
    # a second batch on the same workspace (control words and buckets reused)
    pb.run(ws, thresh)
    torch.cuda.synchronize()
    first, count, q, t, m = ws.survivors(pb.n_pairs)
    assert np.array_equal(np.cumsum(count), a['soff'][1:]) and a['unresolved'] == 0


def test_sym_second_inside_best_group_and_loose_bounds():
    """Adversarial for the bounds of the sweep: the true second neighbour next to the best (same
    group of train rows), on both sides of the threshold; duplicates of the best; exact-zero
    seconds; and an image whose |a-128|^2 values spread widely (loose row-direction bounds)."""
    from imageanalysis_amd import kernels
    rng = np.random.default_rng(77)
    n = 640
    train = _sift_like(rng, n)
    query = _sift_like(rng, 500)
    rows = rng.choice(np.arange(0, n - 1, 2), 200, replace=False)
    for k, r in enumerate(rows):
        query[k] = np.clip(train[r].astype(int) + rng.integers(-2, 3, 128), 0, 255)
        amp = (1, 3, 8, 20)[k % 4]
        train[r + 1] = np.clip(train[r].astype(int) + rng.integers(-amp, amp + 1, 128), 0, 255)
    for k in range(200, 230):
        r = rows[k - 200]
        query[k] = train[r]
        train[r + 1] = train[r]
    query[300:400] = rng.integers(0, 256, (100, 128))              # wide norm spread
    train[500:600] = rng.integers(0, 256, (100, 128))
    store = kernels.DescriptorStore.from_arrays([query, train])
    for thresh in (270.0 * 0.75, 270.0 * 0.6, 1e9):
        a = _run(store, [[0, 1], [1, 0]], thresh, sym=True)
        b = _run(store, [[0, 1], [1, 0]], thresh, fast=False, sym=False)
        assert b['zero_div'] > 0
        _same_survivors(a, b)
        for p, (x, y) in enumerate(((query, train), (train, query))):
            keep, tr, mt, d2, zd = _oracle_survivors(x, y, thresh)
            lo, hi = a['soff'][p], a['soff'][p + 1]
            assert np.array_equal(a['sq'][lo:hi], keep) and np.array_equal(a['st'][lo:hi], tr)
            assert np.array_equal(a['sm'][lo:hi], mt)


def test_sym_extreme_values_and_uniform_images():
    """all-0 / all-255 / constant images: maximum distances, every distance equal, zero seconds"""
    from imageanalysis_amd import kernels
    a0 = np.zeros((200, 128), np.uint8)
    a0[1::2] = 255
    a1 = np.zeros((300, 128), np.uint8)
    a1[::3] = 255
    a1[1::3] = 128
    a2 = np.full((130, 128), 255, np.uint8)
    store = kernels.DescriptorStore.from_arrays([a0, a1, a2])
    pairs = [(i, j) for i in range(3) for j in range(3) if i != j]
    for thresh in (202.5, 1e9):
        a = _run(store, pairs, thresh, sym=True)
        b = _run(store, pairs, thresh, fast=False, sym=False)
        _same_survivors(a, b)
        assert a['zero_div'] > 0


@pytest.mark.parametrize('sizes,rows', [([4096, 4097, 5000, 4224], 1024), ([2048, 2049, 3000, 4095], 512)],
                         ids=['rows1024', 'rows512'])
def test_one_direction_fast_forms_at_their_sizes(sizes, rows):
    """iamx_knn2v2_pairs in its 1024-row (8 waves, direct global->LDS staging) and 512-row
    instantiations -- what batches WITHOUT mirrored pairs run -- against the general kernel on
    every ordered pair and the oracle on four of them."""
    from imageanalysis_amd import kernels
    rng = np.random.default_rng(rows)
    imgs = _survey(rng, sizes)
    imgs[1][2000:2040] = imgs[1][:40]                              # ties: lowest row wins
    store = kernels.DescriptorStore.from_arrays(imgs)
    pairs = [(i, j) for i in range(len(sizes)) for j in range(len(sizes)) if i != j]
    thresh = 270.0 * 0.75
    for form in (True, 'exact'):
        a = _run(store, pairs, thresh, fast=form, sym=False)
        assert not a['pb'].sym and a['unresolved'] == 0
        assert a['pb'].fast_rows == (rows if form is True else 512)
        b = _run(store, pairs, thresh, fast=False, sym=False)
        _same_survivors(a, b)
        assert np.array_equal(a['d2'][:, 0], b['d2'][:, 0])        # best exact on every row
        assert (a['d2'][:, 1] >= b['d2'][:, 1]).all()
        for p in (0, 3, 7, len(pairs) - 1):
            i, j = pairs[p]
            keep, tr, mt, d2, zd = _oracle_survivors(imgs[i], imgs[j], thresh)
            lo, hi = a['soff'][p], a['soff'][p + 1]
            assert np.array_equal(a['sq'][lo:hi], keep) and np.array_equal(a['st'][lo:hi], tr)
            assert np.array_equal(a['sm'][lo:hi], mt)
            assert np.array_equal(a['d2'][a['off'][p] + keep], d2)


def test_config2_arena_pairs_at_both_ends_of_the_store():
    """BASELINE configs[2] store: 2812 images x 4096 descriptors resident in HBM (1.5 GB per
    layout, row offsets up to 11.5 M).  The shipped symmetric form on image pairs that touch the
    first, the middle and the last images of the arena, both directions: bit-equal to
    oracle/cpu_ref.c on 12 ordered pairs; planted overlaps (image j repeats 30 % of image j-1 with
    +-6 noise) show up as hundreds of survivors, unrelated pairs as a handful."""
    import torch
    from imageanalysis_amd import kernels
    dev = kernels.require_gpu()
    n_img, kpts = 2812, 4096
    store = kernels.DescriptorStore([kpts] * n_img)
    alpha = torch.full((kpts, 128), 0.6, device=dev)
    named = [(0, 1), (1, 2811), (2810, 2811), (1405, 1406), (0, 2811), (7, 1406)]
    keep_ids = sorted({i for p in named for i in p})
    kept, prev, pending = {}, None, []
    for j in range(n_img):
        g = torch.Generator(device=dev)
        g.manual_seed(5000 + j)
        x = torch._standard_gamma(alpha, generator=g)
        x = x / x.norm(dim=1, keepdim=True)
        x = x.clamp(max=0.2)
        x = x / x.norm(dim=1, keepdim=True)
        cur = (x * 512.0).round().clamp(0, 255)
        base = cur.clone()
        if prev is not None:
            k = int(0.3 * kpts)
            src = torch.randperm(kpts, generator=g, device=dev)[:k]
            dst = torch.randperm(kpts, generator=g, device=dev)[:k]
            cur[dst] = (prev[src] + torch.randint(-6, 7, (k, 128), generator=g, device=dev)).clamp(0, 255)
        prev = base
        u8 = cur.to(torch.uint8)
        if j in keep_ids:
            kept[j] = u8.cpu().numpy()
        pending.append(store.set_image(j, u8, sync=False))
        if len(pending) >= 64:
            torch.cuda.synchronize()
            pending = []
    torch.cuda.synchronize()
    rng = np.random.default_rng(11)
    und = list(named)
    while len(und) < 1500:
        a, b = (int(v) for v in rng.integers(0, n_img, 2))
        if a != b and (a, b) not in und and (b, a) not in und:
            und.append((min(a, b), max(a, b)))
    ordered = np.array(und + [(b, a) for a, b in und], np.int32)
    thresh = 270.0 * 0.75
    res = _run(store, ordered, thresh, sym=True)
    assert res['pb'].sym and res['pb'].sym_form == 2 and res['unresolved'] == 0
    count = np.diff(res['soff'])
    for p, (a, b) in enumerate(ordered):
        if (int(a), int(b)) in named or (int(b), int(a)) in named:
            keep, tidx, metric, rd2, _z = _oracle_survivors(kept[int(a)], kept[int(b)], thresh)
            lo, hi = res['soff'][p], res['soff'][p + 1]
            assert np.array_equal(res['sq'][lo:hi], keep), (a, b)
            assert np.array_equal(res['st'][lo:hi], tidx) and np.array_equal(res['sm'][lo:hi], metric), (a, b)
            assert np.array_equal(res['d2'][res['off'][p] + keep], rd2), (a, b)
    adjacent = np.abs(ordered[:, 0] - ordered[:, 1]) == 1
    assert adjacent.sum() >= 6 and count[adjacent].min() > 300          # the planted 30 %
    assert count[~adjacent].max() < 60


def test_sym_keypoint_counts_of_full_resolution_frames():
    """20 MP frames at detect scale 0.4 carry ~50 k keypoints: images of 20 000 and 17 777 rows
    (20 / 18 workgroups per pair, 157 chunks per sweep, ragged tail) through the symmetric form,
    both directions, against oracle/cpu_ref.c."""
    from imageanalysis_amd import kernels
    rng = np.random.default_rng(77)
    a, b = _sift_like(rng, 20000), _sift_like(rng, 17777)
    b[rng.permutation(17777)[:3000]] = np.clip(
        a[rng.permutation(20000)[:3000]].astype(int) + rng.integers(-5, 6, (3000, 128)), 0, 255)
    store = kernels.DescriptorStore.from_arrays([a, b])
    thresh = 270.0 * 0.75
    res = _run(store, [[0, 1], [1, 0]], thresh, sym=True)
    assert res['pb'].sym and res['pb'].sym_form == 2 and res['unresolved'] == 0
    for p, (q, t) in enumerate(((a, b), (b, a))):
        keep, tidx, metric, rd2, _z = _oracle_survivors(q, t, thresh)
        lo, hi = res['soff'][p], res['soff'][p + 1]
        assert len(keep) > 2000
        assert np.array_equal(res['sq'][lo:hi], keep) and np.array_equal(res['st'][lo:hi], tidx)
        assert np.array_equal(res['sm'][lo:hi], metric)
        assert np.array_equal(res['d2'][res['off'][p] + keep], rd2)


@pytest.mark.parametrize('prune,sets', [('1', '4'), ('1', '2'), ('0', '2')],
                         ids=['pruned-4-sets', 'pruned-2-sets', 'unpruned'])
@pytest.mark.parametrize('sizes', [(5000, 4097), (700, 20000), (4096, 4096, 333)])
def test_exact_stage_with_many_candidates(sizes, prune, sets, monkeypatch):
    """Pairs whose rows mostly MATCH (real frames of one scene: a third to two thirds of a pair's
    rows are candidates) go through the workgroup form of the exact stage -- 256 candidates per
    task, train tiles shared through LDS -- and pairs with <= 64 candidates through the wave form;
    both against oracle/cpu_ref.c: survivor rows, train rows, metrics, squared distances.  Ragged
    sizes (last tile masked, images starting at odd multiples of 128 rows in the store), planted
    exact duplicates (ties -> lowest train row), a candidate count that is not a multiple of 32.
    All scans of the workgroup form: the one that skips tiles no lane can have its best or second
    in (the candidate test's upper bound of the second distance) with two candidate sets per wave
    and tasks of 256 (shipped) or four and 512, and the full scan."""
    from imageanalysis_amd import kernels
    monkeypatch.setenv('IAMX_EXACT_PRUNE', prune)
    monkeypatch.setenv('IAMX_EXACT_SETS', sets)
    rng = np.random.default_rng(sum(sizes))
    imgs = [_sift_like(rng, sizes[0])]
    for n in sizes[1:]:
        # most rows of the next image are noisy copies of rows of the first one
        src = rng.integers(0, sizes[0], n)
        im = np.clip(imgs[0][src].astype(int) + rng.integers(-3, 4, (n, 128)), 0, 255).astype(np.uint8)
        fresh = rng.random(n) < 0.35
        im[fresh] = _sift_like(rng, int(fresh.sum()))
        imgs.append(im)
    imgs[1][10] = imgs[0][7]                 # an exact copy: distance 0
    imgs[1][11] = imgs[1][10]                # ... twice: the lower train row wins, second distance 0
    imgs.append(_sift_like(rng, 200))        # an unrelated small image: few or no candidates (wave form)
    store = kernels.DescriptorStore.from_arrays(imgs)
    k = len(imgs)
    pairs = [(a, b) for a in range(k) for b in range(k) if a != b]
    thresh = 270.0 * 0.75
    got = _run(store, pairs, thresh, sym=True)
    assert got['pb'].sym and got['unresolved'] == 0
    many = 0
    zero_div = 0
    for p, (a, b) in enumerate(pairs):
        keep, t, m, d2, zd = _oracle_survivors(imgs[a], imgs[b], thresh)
        zero_div += zd
        lo, hi = got['soff'][p], got['soff'][p + 1]
        assert np.array_equal(got['sq'][lo:hi], keep), (a, b)
        assert np.array_equal(got['st'][lo:hi], t), (a, b)
        assert np.array_equal(got['sm'][lo:hi], m), (a, b)
        assert np.array_equal(got['d2'][got['off'][p] + keep], d2), (a, b)
        many += len(keep) > 256
    assert many >= 2 and got['zero_div'] == zero_div


def test_bulk_ingest_equals_image_by_image():
    """DescriptorStore.set_images (one uint8 block from libiamx's threads, one upload, the batched
    pack kernels) fills every layout of the store exactly like set_image() image by image: ragged
    row counts, float32 (the reference's des_list) and uint8 sources"""
    import torch
    from imageanalysis_amd import kernels
    rng = np.random.default_rng(12)
    counts = [4096, 4097, 130, 5000, 2, 3333, 4096, 257, 1024, 700]
    for dtype in (np.float32, np.uint8):
        arrays = [_sift_like(rng, n).astype(dtype) for n in counts]
        a = kernels.DescriptorStore(counts)
        for i, d in enumerate(arrays):
            a.set_image(i, d)
        b = kernels.DescriptorStore(counts)
        keep = b.set_images(0, arrays)
        torch.cuda.synchronize()
        del keep
        for i, n in enumerate(counts):
            o, o2, o3 = int(a.offsets[i]), int(a.offsets2[i]), int(a.offsets3[i])
            for name, lo, m in (('desc', o, n), ('norm_q', o, n), ('norm_t', o, n),
                                ('desc3', o3, int(a.caps3[i])), ('sn2', o3, n), ('sct', o3, n),
                                ('sperm', o3, n), ('sinv', o3, n)):
                assert torch.equal(getattr(a, name)[lo:lo + m], getattr(b, name)[lo:lo + m]), (name, i)
            n2 = int(a.offsets2[i + 1]) - o2
            for name in ('desc2', 'norm2', 'cinit', 'perm'):
                assert torch.equal(getattr(a, name)[o2:o2 + n2], getattr(b, name)[o2:o2 + n2]), (name, i)
        assert torch.equal(a.meta, b.meta)


@pytest.mark.parametrize('train_layout', [True, False], ids=['with_desc2', 'without_desc2'])
def test_store_extended_in_place_equals_store_built_at_once(train_layout):
    """DescriptorStore(reserve_rows, reserve_images) + try_extend (find_matches meeting undetected
    images: a few hundred new images per round go behind the old rows, the arena is not rebuilt):
    every layout of the extended store equals a store built with all images at once, the matches
    of pairs across old and new images are the same, and try_extend says no -- changing nothing --
    when the capacity does not suffice"""
    import torch
    from imageanalysis_amd import kernels
    rng = np.random.default_rng(21)
    counts = [4096, 4097, 130, 5000, 3333, 257, 1024, 700]
    arrays = [_sift_like(rng, n) for n in counts]
    k = 3
    full = kernels.DescriptorStore(counts, train_layout=train_layout)
    full.set_images(0, arrays)
    ext = kernels.DescriptorStore(counts[:k], train_layout=train_layout,
                                  reserve_rows=sum(counts) + 256 * len(counts), reserve_images=len(counts))
    keep = [ext.set_images(0, arrays[:k])]
    ptr = ext.desc.data_ptr()
    assert ext.try_extend(counts[k:])
    assert ext.desc.data_ptr() == ptr and ext.counts == counts
    for i in range(k, len(counts)):
        keep.append(ext.set_image(i, arrays[i], sync=False))
    torch.cuda.synchronize()
    assert np.array_equal(ext.offsets, full.offsets) and np.array_equal(ext.offsets3, full.offsets3)
    assert ext.rows_used == full.rows_used
    n, n3 = int(full.offsets[-1]), int(full.offsets3[-1])
    for name, m in (('desc', n), ('norm_q', n), ('norm_t', n)):
        assert torch.equal(getattr(ext, name)[:m], getattr(full, name)[:m]), name
    for i, c in enumerate(counts):
        o3 = int(full.offsets3[i])
        for name, m in (('desc3', int(full.caps3[i])), ('sn2', c), ('sct', c), ('sperm', c), ('sinv', c)):
            assert torch.equal(getattr(ext, name)[o3:o3 + m], getattr(full, name)[o3:o3 + m]), (name, i)
    assert torch.equal(ext.img_off, full.img_off) and torch.equal(ext.img_n, full.img_n)
    assert torch.equal(ext.img_off3, full.img_off3)
    if train_layout:
        n2 = int(full.offsets2[-1])
        for name in ('desc2', 'norm2', 'cinit', 'perm'):
            assert torch.equal(getattr(ext, name)[:n2], getattr(full, name)[:n2]), name
        assert torch.equal(ext.meta[:len(counts)], full.meta[:len(counts)])
    pairs = [(0, 4), (4, 0), (1, 7), (7, 1), (3, 5), (5, 3), (2, 6), (6, 2)]
    _same_survivors(_run(ext, pairs, 0.7, sym=True), _run(full, pairs, 0.7, sym=True))
    _same_survivors(_run(ext, pairs, 0.7, fast=False, sym=False), _run(full, pairs, 0.7, fast=False, sym=False))
    # no room: nothing changes
    before = list(ext.counts)
    assert not ext.try_extend([1 << 20])
    assert ext.counts == before and ext.desc.data_ptr() == ptr
