"""GPU: the HIP matching kernels (through the C ABI) against the oracle and the golden
vectors.  Integer work -> bit-exact."""
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


def test_single_hip_runtime_and_native_lib_loaded():
    import torch
    from imageanalysis_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()
    maps = open('/proc/self/maps').read()
    hips = {l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}
    assert len(hips) == 1, hips
    assert 'libiamx.so' in maps


@pytest.mark.parametrize('path', MATCH_CASES, ids=os.path.basename)
def test_knn2_golden(path):
    from imageanalysis_amd import kernels
    from oracle import match_oracle as mo
    g = np.load(path)
    for tag, (a, b) in dict(fwd=(g['des1'], g['des2']), rev=(g['des2'], g['des1'])).items():
        idx, d2 = kernels.knn2(a, b)
        assert np.array_equal(idx.cpu().numpy(), g['knn_%s_idx' % tag])
        assert np.array_equal(mo.distances_f32(d2.cpu().numpy()), g['knn_%s_dist' % tag])
        # the reference hands float32 descriptors to knnMatch
        idx, d2b = kernels.knn2(a.astype(np.float32), b.astype(np.float32))
        assert np.array_equal(idx.cpu().numpy(), g['knn_%s_idx' % tag])
        assert np.array_equal(d2b.cpu().numpy(), d2.cpu().numpy())


@pytest.mark.parametrize('nq,nt', [(1, 2), (2, 2), (31, 5), (33, 127), (64, 128), (255, 129),
                                   (257, 300), (1000, 257), (300, 4096), (4096, 1031)])
def test_knn2_ragged_sizes_vs_oracle(nq, nt):
    from imageanalysis_amd import kernels
    from oracle import cpu_ref
    rng = np.random.default_rng(nq * 7919 + nt)
    q = rng.integers(0, 256, (nq, 128), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 128), dtype=np.uint8)
    idx, d2 = kernels.knn2(q, t)
    ridx, rd2 = cpu_ref.knn2_l2_u8(q, t)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(d2.cpu().numpy(), rd2)


def test_knn2_ties_lowest_train_index_wins():
    from imageanalysis_amd import kernels
    from oracle import cpu_ref
    rng = np.random.default_rng(3)
    t = _sift_like(rng, 1500)
    t[700:1400] = t[:700]                  # exact duplicates 700 rows apart (other lane half,
    t[5] = t[4]                            # other 256-row epoch, neighbouring rows)
    q = t[rng.permutation(1500)[:600]].copy()
    idx, d2 = kernels.knn2(q, t)
    ridx, rd2 = cpu_ref.knn2_l2_u8(q, t)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(d2.cpu().numpy(), rd2)
    assert (rd2[:, 0] == 0).all()


def test_knn2_extreme_values_no_overflow():
    from imageanalysis_amd import kernels
    from oracle import cpu_ref
    q = np.zeros((40, 128), np.uint8)
    q[1::2] = 255
    t = np.zeros((300, 128), np.uint8)
    t[::3] = 255
    t[1::3] = 128
    idx, d2 = kernels.knn2(q, t)
    ridx, rd2 = cpu_ref.knn2_l2_u8(q, t)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(d2.cpu().numpy(), rd2)
    # only far neighbours: every distance is the maximum 128*255^2
    idx, d2 = kernels.knn2(np.zeros((5, 128), np.uint8), np.full((9, 128), 255, np.uint8))
    assert (d2.cpu().numpy() == 128 * 255 * 255).all()
    assert (idx.cpu().numpy() == [0, 1]).all()


def test_knn2_rejects_single_row_train_like_reference():
    from imageanalysis_amd import kernels, _lib
    with pytest.raises(_lib.IamxError):
        kernels.knn2(np.zeros((4, 128), np.uint8), np.zeros((1, 128), np.uint8))


def test_knn2_pairs_batch_ragged_images():
    from imageanalysis_amd import kernels
    from oracle import cpu_ref
    rng = np.random.default_rng(9)
    sizes = [300, 2, 257, 1024, 129, 640, 77]
    imgs = [_sift_like(rng, n) for n in sizes]
    imgs[3][:200] = np.clip(imgs[0][:200].astype(int) + rng.integers(-5, 6, (200, 128)), 0, 255)
    store = kernels.DescriptorStore.from_arrays(imgs)
    pairs = [(i, j) for i in range(len(sizes)) for j in range(len(sizes)) if i != j]
    idx, d2, off = kernels.knn2_pairs(store, np.array(pairs, np.int32))
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    for p, (i, j) in enumerate(pairs):
        ridx, rd2 = cpu_ref.knn2_l2_u8(imgs[i], imgs[j])
        assert np.array_equal(idx[off[p]:off[p + 1]], ridx), (i, j)
        assert np.array_equal(d2[off[p]:off[p + 1]], rd2), (i, j)


def test_metric_kernel_sqrt_and_division_exact():
    """float32 sqrt + float64 divide/multiply on the device == numpy (IEEE) for every
    representable squared distance class we can meet."""
    import torch
    from imageanalysis_amd import kernels
    rng = np.random.default_rng(1)
    n = 1 << 20
    d0 = rng.integers(0, 128 * 255 * 255 + 1, n)
    d1 = np.maximum(d0, rng.integers(1, 128 * 255 * 255 + 1, n))
    d2 = np.stack([d0, d1], 1).astype(np.int32)
    d2[:4096, 0] = np.arange(4096)               # small values incl. perfect squares
    d2[:4096, 1] = np.maximum(d2[:4096, 0], 1)
    seg = np.array([0, 1000, 1000, n], np.int64)
    metric, keep, cnt, zd = kernels.match_metric(torch.from_numpy(d2).cuda(), seg, 202.5)
    f = np.sqrt(d2.astype(np.float32)).astype(np.float64)
    ref = f[:, 0] * (f[:, 0] / f[:, 1])
    assert np.array_equal(metric.cpu().numpy(), ref)
    rk = ref < 202.5
    assert np.array_equal(keep.cpu().numpy().astype(bool), rk)
    assert cnt.cpu().numpy().tolist() == [int(rk[:1000].sum()), 0, int(rk[1000:].sum())]
    assert int(zd.item()) == 0


def test_metric_kernel_sqrt_exhaustive():
    """every possible squared distance 0..128*255^2: device float32 sqrt == IEEE sqrt."""
    import torch
    from imageanalysis_amd import kernels
    n = 128 * 255 * 255 + 1
    d = np.arange(n, dtype=np.int64)
    d2 = np.stack([d, np.maximum(d, 1)], 1).astype(np.int32)     # ratio 1 -> metric = d0
    metric, keep, cnt, zd = kernels.match_metric(torch.from_numpy(d2).cuda(),
                                                 np.array([0, n], np.int64), 202.5)
    f = np.sqrt(d2.astype(np.float32)).astype(np.float64)
    assert np.array_equal(metric.cpu().numpy(), f[:, 0] * (f[:, 0] / f[:, 1]))


def test_metric_kernel_flags_zero_division():
    import torch
    from imageanalysis_amd import kernels
    d2 = torch.tensor([[0, 0], [4, 9], [0, 5]], dtype=torch.int32).cuda()
    metric, keep, cnt, zd = kernels.match_metric(d2, np.array([0, 3], np.int64), 202.5)
    assert int(zd.item()) == 1
    m = metric.cpu().numpy()
    assert np.isnan(m[0]) and m[1] == 2.0 * (2.0 / 3.0) and m[2] == 0.0
    assert keep.cpu().numpy().tolist() == [0, 1, 1]


@pytest.mark.parametrize('path', MATCH_CASES, ids=os.path.basename)
def test_metric_filter_and_compaction_golden(path):
    """device top-2 -> metric -> threshold -> compaction, then the host's stable sort + clip
    == the list the reference hands to matchGMS (lib/matcher.py:253-269)."""
    import torch
    from imageanalysis_amd import kernels
    g = np.load(path)
    imgs = [g['des1'], g['des2']]
    store = kernels.DescriptorStore.from_arrays(imgs)
    idx, d2, off = kernels.knn2_pairs(store, np.array([[0, 1], [1, 0]], np.int32))
    thresh = 270.0 * float(g['match_ratio'])
    metric, keep, cnt, zd = kernels.match_metric(d2, off, thresh)
    soff = kernels.exclusive_scan(cnt)
    total = int(soff[-1].item())
    sq, st, sm = kernels.match_compact(idx, metric, keep, off, soff, total)
    soff = soff.cpu().numpy()
    sq, st, sm = sq.cpu().numpy(), st.cpu().numpy(), sm.cpu().numpy()
    for p, tag in enumerate(['fwd', 'rev']):
        a, b = soff[p], soff[p + 1]
        order = np.argsort(sm[a:b], kind='stable')[:2000]
        got = np.stack([sq[a:b][order], st[a:b][order]], 1)
        want = g['pregms_%s' % tag]
        if len(want) == 0:
            assert len(got) < int(g['min_pairs'])
        else:
            assert np.array_equal(got, want)


def test_config2_shape_properties():
    """BASELINE config 2 shape (4096 x 4096 x 128) where the oracle is too slow to run for
    every pair: size-independent properties.  (a) a planted copy is found at distance 0 with
    the right index, (b) query-order permutation invariance, (c) one pair vs the C oracle."""
    from imageanalysis_amd import kernels
    from oracle import cpu_ref
    rng = np.random.default_rng(42)
    imgs = [_sift_like(rng, 4096) for _ in range(3)]
    perm = rng.permutation(4096)
    imgs[1][perm[:1200]] = imgs[0][:1200]
    store = kernels.DescriptorStore.from_arrays(imgs)
    idx, d2, off = kernels.knn2_pairs(store, np.array([[0, 1], [1, 0], [2, 1], [0, 2]], np.int32))
    idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
    assert (d2[:1200, 0] == 0).all() and np.array_equal(idx[:1200, 0], perm[:1200])
    ridx, rd2 = cpu_ref.knn2_l2_u8(imgs[2], imgs[1])
    assert np.array_equal(idx[off[2]:off[3]], ridx) and np.array_equal(d2[off[2]:off[3]], rd2)
    # permuting the queries permutes the answers
    p2 = rng.permutation(4096)
    i2, dd2 = kernels.knn2(imgs[0][p2], imgs[2])
    assert np.array_equal(i2.cpu().numpy(), idx[off[3]:off[4]][p2])
    assert np.array_equal(dd2.cpu().numpy(), d2[off[3]:off[4]][p2])


def _run_batch(store, pairs, thresh, fast, sym=False):
    """survivor lists are returned densely packed per pair (soff = exclusive scan of counts);
    sym=False: one sweep per ordered pair (the symmetric sweep has its own tests in
    tests/test_match_sym_gpu.py)"""
    import torch
    from imageanalysis_amd import kernels
    pb = kernels.PairBatch(store, np.asarray(pairs, np.int32), sym=sym)
    ws = kernels.PairWorkspace(pb.rows, pb.n_pairs)
    pb.run(ws, thresh, fast=fast)
    torch.cuda.synchronize()
    first, count, q, t, m = ws.survivors(pb.n_pairs)
    take = np.concatenate([np.arange(f, f + c) for f, c in zip(first, count)] + [np.zeros(0, np.int64)])
    take = take.astype(np.int64)
    soff = np.zeros(pb.n_pairs + 1, np.int64)
    np.cumsum(count, out=soff[1:])
    return dict(d2=ws.d2[:pb.rows].cpu().numpy(), soff=soff, sq=q[take], st=t[take], sm=m[take],
                unresolved=int(ws.unresolved.item()), zero_div=int(ws.zero_div.item()),
                off=pb.out_off)


def test_desc2_store_layout():
    """parity-partitioned train store: stable partition, padding, C-operand terms"""
    from imageanalysis_amd import kernels
    rng = np.random.default_rng(4)
    for n in (2, 127, 128, 129, 1000, 4096):
        a = _sift_like(rng, n)
        st = kernels.DescriptorStore.from_arrays([a])
        meta = st.meta.cpu().numpy()[0]
        s = a.astype(np.int64) - 128
        nb = (s * s).sum(1) + 2 * s.sum(1)
        even = np.nonzero(nb % 2 == 0)[0]
        odd = np.nonzero(nb % 2 != 0)[0]
        ne_pad, no_pad = -(-len(even) // 128) * 128, -(-len(odd) // 128) * 128
        assert meta.tolist() == [n, ne_pad // 128, no_pad // 128, len(even)]
        perm = st.perm.cpu().numpy()[:ne_pad + no_pad]
        assert np.array_equal(perm[:len(even)], even) and (perm[len(even):ne_pad] == -1).all()
        assert np.array_equal(perm[ne_pad:ne_pad + len(odd)], odd)
        assert (perm[ne_pad + len(odd):] == -1).all()
        rows = st.desc2.cpu().numpy()[:ne_pad + no_pad].astype(np.int64)
        ok = perm >= 0
        assert np.array_equal(rows[ok], s[perm[ok]]) and (rows[~ok] == 0).all()
        cin = st.cinit.cpu().numpy()[:ne_pad + no_pad]
        assert np.array_equal(cin[ok], nb[perm[ok]] >> 1) and (cin[~ok] == 0x3F000000).all()
        assert np.array_equal(st.norm2.cpu().numpy()[:ne_pad + no_pad][ok], (s * s).sum(1)[perm[ok]])


@pytest.mark.parametrize('form', [True, 'exact'], ids=['bound', 'exact-second'])
@pytest.mark.parametrize('path', MATCH_CASES, ids=os.path.basename)
def test_fast_path_equals_general_path_golden(path, form):
    from imageanalysis_amd import kernels
    g = np.load(path)
    store = kernels.DescriptorStore.from_arrays([g['des1'], g['des2']])
    thresh = 270.0 * float(g['match_ratio'])
    a = _run_batch(store, [[0, 1], [1, 0]], thresh, fast=form)
    b = _run_batch(store, [[0, 1], [1, 0]], thresh, fast=False)
    assert a['unresolved'] == 0 and a['zero_div'] == b['zero_div']
    for k in ('soff', 'sq', 'st', 'sm'):
        assert np.array_equal(a[k], b[k]), k
    if form == 'exact':
        assert np.array_equal(a['d2'], b['d2'])
    else:
        # bound form: best exact everywhere; second exact on every survivor, an upper bound
        # (smallest distance outside the best's 16-row group) elsewhere
        assert np.array_equal(a['d2'][:, 0], b['d2'][:, 0])
        assert (a['d2'][:, 1] >= b['d2'][:, 1]).all()
        rows = np.concatenate([a['off'][p] + a['sq'][a['soff'][p]:a['soff'][p + 1]] for p in (0, 1)])
        assert np.array_equal(a['d2'][rows], b['d2'][rows])
    # and == the reference's pre-GMS list after the host's stable sort + clip
    for p, tag in enumerate(['fwd', 'rev']):
        lo, hi = a['soff'][p], a['soff'][p + 1]
        order = np.argsort(a['sm'][lo:hi], kind='stable')[:2000]
        got = np.stack([a['sq'][lo:hi][order], a['st'][lo:hi][order]], 1)
        if len(g['pregms_%s' % tag]):
            assert np.array_equal(got, g['pregms_%s' % tag])


def test_bound_form_second_inside_best_group():
    """Adversarial for the bound form: the true second neighbour sits in the same 16-row group
    as the best (so the sweep only sees an upper bound), on both sides of the threshold, with
    duplicates of the best (second == best) and exact-zero seconds (ZeroDivisionError path)."""
    from imageanalysis_amd import kernels
    from oracle import cpu_ref, match_oracle
    rng = np.random.default_rng(77)
    n = 640
    train = _sift_like(rng, n)
    query = _sift_like(rng, 500)
    # queries 0..199: best = train row r, second = a near copy placed right next to it
    rows = rng.choice(np.arange(0, n - 1, 2), 200, replace=False)
    for k, r in enumerate(rows):
        query[k] = np.clip(train[r].astype(int) + rng.integers(-2, 3, 128), 0, 255)
        amp = (1, 3, 8, 20)[k % 4]                     # some pass the ratio test, some fail it
        train[r + 1] = np.clip(train[r].astype(int) + rng.integers(-amp, amp + 1, 128), 0, 255)
    for k in range(200, 230):                          # duplicates of the best: second == best
        r = rows[k - 200]
        query[k] = train[r]
        train[r + 1] = train[r]                        # exact zero second for these queries
    store = kernels.DescriptorStore.from_arrays([query, train])
    for thresh in (270.0 * 0.75, 270.0 * 0.6, 1e9):
        a = _run_batch(store, [[0, 1], [1, 0]], thresh, fast=True)
        b = _run_batch(store, [[0, 1], [1, 0]], thresh, fast=False)
        assert a['unresolved'] == 0
        assert a['zero_div'] == b['zero_div'] and b['zero_div'] > 0
        for k in ('soff', 'sq', 'st', 'sm'):
            assert np.array_equal(a[k], b[k]), (thresh, k)
        ridx, rd2 = cpu_ref.knn2_l2_u8(query, train)
        lo, hi = a['soff'][0], a['soff'][1]
        assert np.array_equal(a['d2'][:500][a['sq'][lo:hi]], rd2[a['sq'][lo:hi]])
        assert np.array_equal(a['st'][lo:hi], ridx[a['sq'][lo:hi], 0])
    # the bound really was loose somewhere (otherwise this test exercises nothing)
    pb = kernels.PairBatch(store, np.array([[0, 1]], np.int32), sym=False)
    ws = kernels.PairWorkspace(pb.rows, pb.n_pairs)
    pb.run_knn2_fast(ws)
    d2b = ws.d2[:500].cpu().numpy()
    assert (d2b[:, 1] > rd2[:, 1]).sum() >= 30 and np.array_equal(d2b[:, 0], rd2[:, 0])


def test_fast_path_ragged_ties_and_tiny_classes():
    from imageanalysis_amd import kernels
    from oracle import cpu_ref
    rng = np.random.default_rng(21)
    sizes = [300, 2, 257, 1024, 129, 640, 3]
    imgs = [_sift_like(rng, n) for n in sizes]
    imgs[3][:200] = np.clip(imgs[0][:200].astype(int) + rng.integers(-5, 6, (200, 128)), 0, 255)
    imgs[3][500:700] = imgs[3][:200]                 # exact duplicates: lowest row must win
    imgs[5][:129] = imgs[4]                          # identical rows across images (distance 0)
    imgs[1][:] = 128                                 # an image whose rows all share one parity
    store = kernels.DescriptorStore.from_arrays(imgs)
    pairs = [(i, j) for i in range(len(sizes)) for j in range(len(sizes)) if i != j]
    # a generous threshold keeps most rows so the index resolve is exercised everywhere
    a = _run_batch(store, pairs, 1e9, fast=True)
    assert a['unresolved'] == 0
    for p, (i, j) in enumerate(pairs):
        ridx, rd2 = cpu_ref.knn2_l2_u8(imgs[i], imgs[j])
        assert np.array_equal(a['d2'][a['off'][p]:a['off'][p + 1]], rd2), (i, j)
        lo, hi = a['soff'][p], a['soff'][p + 1]
        keep = rd2[:, 1] > 0                         # d1 == 0 -> NaN metric, never kept
        assert np.array_equal(a['sq'][lo:hi], np.nonzero(keep)[0]), (i, j)
        assert np.array_equal(a['st'][lo:hi], ridx[keep, 0]), (i, j)


@pytest.mark.parametrize('sym', [True, False], ids=['symmetric', 'one-direction'])
def test_overlapped_sweeps_equal_sequential_runs(sym):
    """kernels.OverlappedSweeps (filter kernels of launch k on a second stream beside the sweep
    of launch k+1, two workspaces) leaves exactly what the one-stream sequence leaves"""
    import torch
    from imageanalysis_amd import kernels
    rng = np.random.default_rng(5)
    sizes = [700, 513, 1024, 300, 900, 640]
    imgs = [_sift_like(rng, n) for n in sizes]
    for k in range(1, len(imgs)):                    # overlapping neighbours: real survivors
        m = min(len(imgs[k]), len(imgs[k - 1])) // 3
        imgs[k][:m] = np.clip(imgs[k - 1][:m].astype(int) + rng.integers(-4, 5, (m, 128)), 0, 255)
    store = kernels.DescriptorStore.from_arrays(imgs)
    und = np.array([(i, j) for j in range(len(sizes)) for i in range(j)], np.int32)
    batches = [kernels.PairBatch(store, np.concatenate([und[s:s + 3], und[s:s + 3, ::-1]]), sym=sym)
               for s in range(0, len(und), 3)]                                      # 5 launches
    assert all(b.sym == sym for b in batches)
    rows, npairs = max(b.rows for b in batches), max(b.n_pairs for b in batches)
    thresh = 270.0 * 0.75

    def snapshot(b, w):
        return (w.surv_cnt[:b.n_pairs].clone(), w.surv_off[:b.n_pairs + 1].clone(),
                w.surv_q.clone(), w.surv_t.clone(), w.surv_metric.clone())

    def segments(snap):
        """the survivor slices of every pair (what lies between them is scratch)"""
        cnt, off, q, t, m = snap
        out = []
        for p in range(len(cnt)):
            lo, hi = int(off[p]), int(off[p]) + int(cnt[p])
            out.append((q[lo:hi].cpu().numpy(), t[lo:hi].cpu().numpy(), m[lo:hi].cpu().numpy()))
        return out

    ws = kernels.PairWorkspace(rows, npairs)
    want = []
    for b in batches:
        b.run_knn2_fast(ws)
        b.run_filter_fast(ws, thresh)
        torch.cuda.synchronize()
        want.append(segments(snapshot(b, ws)))
    runner = kernels.OverlappedSweeps(rows, npairs)
    # snapshots are taken on the side stream right behind each launch's filter kernels
    lazy = []
    runner.run(batches, thresh, after_filter=lambda b, w: lazy.append(snapshot(b, w)))
    torch.cuda.synchronize()
    assert len(lazy) == len(batches)
    n_surv = 0
    for snap, ref in zip(lazy, want):
        got = segments(snap)
        assert len(got) == len(ref)
        for (q, t, m), (wq, wt, wm) in zip(got, ref):
            assert np.array_equal(q, wq) and np.array_equal(t, wt) and np.array_equal(m, wm)
            n_surv += len(q)
    assert n_surv > 0
    assert sum(int(w.unresolved.item()) for w in runner.ws) == 0
