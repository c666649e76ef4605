"""CPU: the oracle (numpy + plain C restatement) against the golden vectors that were produced
by the reference's own scripts/lib/{matcher,optimizer}.py (oracle/gen_golden.py)."""
import glob
import os
import pickle

import numpy as np
import pytest

from oracle import ba_oracle as bo
from oracle import cpu_ref
from oracle import match_oracle as mo

from conftest import GOLDEN

MATCH_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'match_*.npz')))
BA_CASES = sorted(glob.glob(os.path.join(GOLDEN, 'ba_*.npz')))


def test_fixtures_present():
    assert len(MATCH_CASES) >= 5 and len(BA_CASES) >= 4


@pytest.mark.parametrize('path', MATCH_CASES, ids=os.path.basename)
def test_match_pipeline_matches_reference(path):
    g = np.load(path)
    size = (int(g['width']), int(g['height']))
    ratio, min_pairs = float(g['match_ratio']), int(g['min_pairs'])
    for tag, (da, xa, db, xb) in dict(fwd=(g['des1'], g['xy1'], g['des2'], g['xy2']),
                                      rev=(g['des2'], g['xy2'], g['des1'], g['xy1'])).items():
        st = {}
        basic = mo.basic_pair_matches(da, xa, db, xb, ratio, min_pairs, size, st)
        assert np.array_equal(st['knn_idx'], g['knn_%s_idx' % tag])
        assert np.array_equal(mo.distances_f32(st['knn_d2']), g['knn_%s_dist' % tag])
        if len(g['pregms_%s' % tag]):
            assert np.array_equal(st['pregms'], g['pregms_%s' % tag])
            assert np.array_equal(st['postgms'], g['postgms_%s' % tag])
        else:
            assert len(st['pregms']) < min_pairs         # reference quit before matchGMS
        assert np.array_equal(basic, g['basic_%s' % tag])
    f, r = mo.bidirectional_pair_matches(g['des1'], g['xy1'], g['des2'], g['xy2'], ratio,
                                         min_pairs, size)
    assert np.array_equal(f, g['bidir_fwd']) and np.array_equal(r, g['bidir_rev'])


@pytest.mark.parametrize('path', MATCH_CASES, ids=os.path.basename)
def test_c_knn2_matches_reference(path):
    g = np.load(path)
    for tag, (a, b) in dict(fwd=(g['des1'], g['des2']), rev=(g['des2'], g['des1'])).items():
        idx, d2 = cpu_ref.knn2_l2_u8(a, b)
        assert np.array_equal(idx, g['knn_%s_idx' % tag])
        assert np.array_equal(mo.distances_f32(d2), g['knn_%s_dist' % tag])
        idx1, d21 = cpu_ref.knn2_l2_u8(a, b, nthreads=1)
        assert np.array_equal(idx, idx1) and np.array_equal(d2, d21)


def test_c_knn2_ties_and_extremes():
    rng = np.random.default_rng(5)
    t = rng.integers(0, 256, (300, 128), dtype=np.uint8)
    t[100:200] = t[:100]                      # every row has an exact duplicate
    q = t[rng.permutation(300)[:64]].copy()
    idx, d2 = cpu_ref.knn2_l2_u8(q, t)
    idx2, d22 = mo.knn2_l2(q, t)
    assert np.array_equal(idx, idx2) and np.array_equal(d2, d22)
    q = np.zeros((3, 128), np.uint8)
    t = np.full((4, 128), 255, np.uint8)
    idx, d2 = cpu_ref.knn2_l2_u8(q, t)
    assert (d2 == 128 * 255 * 255).all() and (idx == [0, 1]).all()


def test_metric_zero_division_like_python():
    idx = np.array([[0, 1]], np.int32)
    with pytest.raises(ZeroDivisionError):
        mo.metric_filter(idx, np.array([[0, 0]], np.int32), 0.75)


def _calib(g):
    K = g['K']
    return [K[0, 0], K[1, 1], K[0, 2], K[1, 2]]


@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
def test_ba_residual_matches_reference(path):
    g = np.load(path)
    C, P = int(g['n_cameras']), int(g['n_points'])
    r = bo.residuals(g['x0'], C, P, g['camera_indices'], g['point_indices'], g['points_2d'],
                     g['K'], g['dist'], bool(g['cam_calib']))
    scale = np.abs(g['f0']).max()
    assert np.abs(r - g['f0']).max() <= 1e-9 * scale
    # reference's rvec/tvec (lib/optimizer.py:120-126) <-> oracle's direct R, t
    import math
    for c in range(C):
        R, t = bo.camera_rt(g['x0'][c * 7:c * 7 + 7])
        assert np.allclose(t, g['tvecs'][c], rtol=0, atol=1e-9)
        rv = g['rvecs'][c]
        th = math.sqrt(float(rv @ rv))
        k = rv / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        Rr = math.cos(th) * np.identity(3) + (1 - math.cos(th)) * np.outer(k, k) + math.sin(th) * Kx
        assert np.allclose(R, Rr, rtol=0, atol=1e-12)
    if not bool(g['cam_calib']):
        x = g['x0']
        rc = cpu_ref.ba_residual(x[:C * 7], x[C * 7:C * 7 + P * 3], g['camera_indices'],
                                 g['point_indices'], g['points_2d'], _calib(g), g['dist'])
        assert np.abs(rc - g['f0']).max() <= 1e-9 * scale


@pytest.mark.parametrize('path', BA_CASES, ids=os.path.basename)
def test_ba_setup_matches_reference(path):
    g = np.load(path)
    with open(path.replace('.npz', '_in.pkl'), 'rb') as f:
        inp = pickle.load(f)
    s = bo.setup(inp['names'], inp['groups'], 0, inp['matches'])
    assert np.array_equal(s['camera_map_fwd'], g['camera_map_fwd'])
    assert np.array_equal(s['feat_map_rev'], g['feat_map_rev'])
    assert np.array_equal(s['camera_indices'], g['camera_indices'])
    assert np.array_equal(s['point_indices'], g['point_indices'])
    assert np.array_equal(s['points_2d'], g['points_2d'])
    assert np.array_equal(s['by_camera_counts'], g['by_camera_counts'])


def test_c_knn2_batch_equals_single():
    rng = np.random.default_rng(2)
    imgs = rng.integers(0, 256, (4, 300, 128), dtype=np.uint8)
    pairs = np.array([[0, 1], [1, 0], [2, 3], [3, 1]], np.int32)
    idx, d2 = cpu_ref.knn2_l2_u8_batch(imgs, pairs)
    for p, (a, b) in enumerate(pairs):
        i, d = cpu_ref.knn2_l2_u8(imgs[a], imgs[b])
        assert np.array_equal(i, idx[p]) and np.array_equal(d, d2[p])


# ---- oracle/sift_ref.c (the hot loops of sift_oracle.py in C) against its numpy twins -----------
def _texture(h, w, seed):
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.float64)
    for s in (2, 4, 8, 16):
        g = rng.normal(size=(h // s + 2, w // s + 2))
        img += np.kron(g, np.ones((s, s)))[:h, :w] * s ** 0.7
    img = (img - img.min()) / (img.max() - img.min()) * 255
    return img.clip(0, 255).astype(np.uint8)


def test_fma32_emulation_is_the_fused_multiply_add():
    """sift_oracle.fma32 (the definition of the oracle's Gaussian taps in numpy) == fmaf, also
    where the float64 sum lands exactly on a float32 tie (double rounding)"""
    from oracle import sift_oracle as so
    rng = np.random.default_rng(0)
    a = rng.normal(size=200000).astype(np.float32) * np.float32(100)
    c = rng.normal(size=200000).astype(np.float32)
    k = np.float32(0.12345678)
    want = np.array([np.float32(__import__('math').fma(float(x), float(k), float(y)))
                     for x, y in zip(a[:2000], c[:2000])]) if hasattr(__import__('math'), 'fma') else None
    got = so.fma32(a, k, c)
    if want is not None:
        assert np.array_equal(got[:2000], want)
    # against the C library's fmaf through the oracle's blur: a 1 x n image and a one-tap kernel
    # is n independent fmaf(v, k, 0); a 3-tap kernel chains them
    from oracle import cpu_ref
    img = a.reshape(400, 500)
    taps = np.array([0.25, 0.5, 0.25], np.float32)
    assert np.array_equal(cpu_ref.sift_blur(img, taps), _blur_py(img, taps))
    # a constructed double-rounding case: a*b exact = 1 + 2^-24 + 2^-60-ish relative to c
    x = np.array([1.0 + 2.0 ** -12], np.float32)
    y = np.float32(1.0 + 2.0 ** -12)
    cc = np.array([2.0 ** -30], np.float32)
    p = float(x[0]) * float(y) + float(cc[0])          # not representable: reference by fractions
    from fractions import Fraction
    exact = Fraction(float(x[0])) * Fraction(float(y)) + Fraction(float(cc[0]))
    lo = np.float32(float(exact))
    cands = [lo, np.nextafter(lo, np.float32(np.inf)), np.nextafter(lo, np.float32(-np.inf))]
    best = min(cands, key=lambda v: abs(Fraction(float(v)) - exact))
    assert so.fma32(x, y, cc)[0] == best and p == p


def _blur_py(img, taps):
    from oracle import sift_oracle as so
    r = len(taps) // 2
    h, w = img.shape
    xs, ys = np.arange(w), np.arange(h)
    tmp = np.zeros_like(img)
    for t in range(-r, r + 1):
        tmp = so.fma32(img[:, so._reflect101(xs + t, w)], taps[t + r], tmp)
    out = np.zeros_like(img)
    for t in range(-r, r + 1):
        out = so.fma32(tmp[so._reflect101(ys + t, h), :], taps[t + r], out)
    return out


@pytest.mark.parametrize('shape', [(61, 83), (9, 40), (128, 20)])
def test_c_blur_equals_numpy_twin(shape):
    """BORDER_REFLECT_101 with radii larger than the image included"""
    from oracle import sift_oracle as so
    img = _texture(shape[0], shape[1], 3).astype(np.float32)
    for sigma in (1.2263, 3.09, 1.6):
        assert np.array_equal(so.gaussian_blur(img, sigma), so.gaussian_blur_py(img, sigma))


def test_c_keypoints_and_descriptors_equal_numpy_twins():
    from oracle import sift_oracle as so
    img = _texture(120, 150, 7)
    ka, da = so.detect_and_compute(img, use_c=True)
    kb, db = so.detect_and_compute(img, use_c=False)
    assert len(ka) == len(kb) > 150
    assert np.array_equal(ka, kb) and np.array_equal(da, db)   # the same float32 operations


def test_opencv_float32_helpers():
    """the scalar conventions the oracle restates: cv::fastAtan2's polynomial (good to ~0.3 deg,
    45 deg comes out as 44.99...), exp32 (an expf within ~1 ulp), Matx33f::solve = Cramer's rule
    (zero vector for a singular matrix)"""
    import math
    from oracle import sift_oracle as so
    rng = np.random.default_rng(5)
    for y, x in rng.normal(size=(2000, 2)) * 10:
        want = math.degrees(math.atan2(np.float32(y), np.float32(x))) % 360.0
        got = float(so.fast_atan2(y, x))
        assert abs(((got - want) + 180.0) % 360.0 - 180.0) < 0.3
    assert so.fast_atan2(0, 0) == 0 and so.fast_atan2(1, 0) == 90 and so.fast_atan2(0, -1) == 180
    assert abs(float(so.fast_atan2(1, 1)) - 44.99046) < 1e-4       # the published value of the polynomial
    for x in np.concatenate([-rng.uniform(0, 12, 3000), [0.0, -1e-8, -86.9, -88.0]]):
        x = np.float32(x)
        want = math.exp(float(x))
        got = float(so.exp32(x))
        assert got == 0.0 if x < -87 else abs(got - want) <= 1.5 * float(np.spacing(np.float32(want)))
    A = rng.normal(size=(3, 3)).astype(np.float32)
    b = rng.normal(size=3).astype(np.float32)
    x = np.array(so._solve3_cramer(A.tolist() and [[np.float32(v) for v in r] for r in A],
                                   [np.float32(v) for v in b]), np.float64)
    assert np.allclose(x, np.linalg.solve(A.astype(np.float64), b.astype(np.float64)), rtol=1e-3, atol=1e-4)
    sing = [[np.float32(1), np.float32(2), np.float32(3)]] * 3
    assert [float(v) for v in so._solve3_cramer(sing, [np.float32(1)] * 3)] == [0.0, 0.0, 0.0]


def test_remove_duplicated_sorted_semantics():
    """KeyPointsFilter::removeDuplicatedSorted: KeyPoint12_LessThan order (x, y up; size down;
    angle up; response down; octave down, compared BEFORE the first-octave adjustment), then
    everything equal to the last kept row in (x, y, size, angle) goes -- numpy twin == C"""
    import ctypes
    from oracle import sift_oracle as so
    oc = lambda o, layer: ((o - 1) & 255) | (layer << 8)          # as detectAndCompute reports it
    rows = np.array([
        [5.0, 1.0, 2.0, 30.0, 0.02, oc(0, 1)],
        [5.0, 1.0, 2.0, 30.0, 0.05, oc(0, 1)],      # duplicate of row 0 with the HIGHER response: survives
        [5.0, 1.0, 2.0, 10.0, 0.01, oc(0, 1)],      # same point, smaller angle: before both
        [5.0, 1.0, 3.0, 10.0, 0.01, oc(0, 2)],      # larger size first
        [4.0, 9.0, 2.0, 0.0, 0.01, oc(1, 1)],       # smaller x first of all
        [5.0, 1.0, 2.0, 30.0, 0.05, oc(1, 1)],      # ties to row 1 up to the octave: higher octave first
    ])
    kept, idx, removed = so.remove_duplicated_sorted(rows)
    assert idx.tolist() == [4, 3, 2, 5] and removed == 2
    rng = np.random.default_rng(1)
    big = rng.integers(0, 6, (4000, 6)).astype(np.float64)
    big[:, 5] = [oc(int(a) % 4, 1 + int(b) % 3) for a, b in zip(big[:, 5], big[:, 4])]
    big[:, 4] = (big[:, 4] + 1) / 64
    kept, idx, removed = so.remove_duplicated_sorted(big)
    buf = np.ascontiguousarray(big.copy())
    # (the C comparator sees the packed octave the way findScaleSpaceExtrema wrote it)
    pre = buf[:, 5].astype(np.int64)
    buf[:, 5] = (pre & ~255) | ((pre + 1) & 255)
    L = cpu_ref.lib()
    L.oracle_sift_remove_duplicated_sorted.restype = ctypes.c_int
    n = L.oracle_sift_remove_duplicated_sorted(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(len(buf)))
    assert n == len(kept) and removed == len(big) - n
    assert np.array_equal(buf[:n, :5], kept[:, :5])


def test_all_c_sift_equals_the_oracle():
    """oracle_sift_detect (the CPU baseline of bench.py's SIFT section: everything in C) ==
    sift_oracle.detect_and_compute (numpy pyramid glue + the C loops), which in turn equals the
    numpy twins (test above)"""
    from oracle import sift_oracle as so
    for shape, seed in (((120, 150), 7), ((97, 203), 11)):
        img = _texture(shape[0], shape[1], seed)
        ka, da = so.detect_and_compute(img)
        kc, dc = so.detect_and_compute_c(img)
        assert len(ka) == len(kc) > 100
        assert np.array_equal(ka, kc) and np.array_equal(da, dc)


def test_simd_knn2_equals_the_plain_c_loop():
    """oracle/knn2_simd.c (the AVX-512 VNNI form bench.py times as cpu_baseline) == cpu_ref.c:
    ragged row counts, duplicates (ties -> lowest train index), extreme values"""
    if not cpu_ref.knn2_simd_available():
        pytest.skip("no AVX-512 VNNI on this host")
    rng = np.random.default_rng(3)
    for n_rows in (2, 17, 63, 64, 65, 500):
        imgs = rng.integers(0, 256, (3, n_rows, 128), dtype=np.uint8)
        imgs[2, :n_rows // 2] = imgs[2, n_rows - n_rows // 2:]
        imgs[0, 0] = 0
        imgs[1, -1] = 255
        pairs = np.array([[0, 1], [1, 0], [2, 2], [0, 2], [2, 1]], np.int32)
        a = cpu_ref.knn2_l2_u8_batch(imgs, pairs)
        b = cpu_ref.knn2_l2_u8_batch_simd(imgs, pairs)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), n_rows


def test_identities_the_device_sift_kernels_rely_on():
    """Three rewrites inside csrc/sift.hip that must not change a bit, checked on the host:
    (1) orient_kernel's window index e // side as a float32 product ((e + 0.5) * (1 / side), side <
    512); (2) descriptor_kernel's orientation-bin wrap as `o0 & 7` for every bin floor() can give
    (-8 .. 8); (3) run combining -- a lane adds the terms of consecutive samples that fall into one
    histogram cell in float64 registers first and sends the partial sums to the bin afterwards:
    with sums of float32 terms exact in float64 the bins, rounded to float32 once, are the oracle's
    whatever the grouping (the oracle's descriptor() bins terms one by one)"""
    from oracle import sift_oracle as so
    for side in range(1, 512):
        e = np.arange(side * side + 64, dtype=np.int64)
        q = ((e.astype(np.float32) + np.float32(0.5)) * (np.float32(1) / np.float32(side))).astype(np.int64)
        assert np.array_equal(q, e // side), side
    for o0 in range(-8, 9):
        want = o0 + 8 if o0 < 0 else (o0 - 8 if o0 >= 8 else o0)
        assert (o0 & 7) == want
    rng = np.random.default_rng(5)
    for _ in range(20):
        n_terms, n_bins = 4000, 360
        # magnitudes like mag * weight * trilinear fractions: products of float32 factors in (0, 1]
        terms = (rng.random(n_terms) * rng.random(n_terms) * rng.random(n_terms) * 300).astype(np.float32)
        terms[rng.random(n_terms) < 0.05] *= np.float32(1e-6)
        cell = np.sort(rng.integers(0, n_bins, n_terms))[rng.permutation(n_terms) // 7 * 7 % n_terms]
        one_by_one = so._bin_sums(n_bins, cell.tolist(), terms.tolist())
        # grouped: runs of equal cell in a lane-like blocked walk, partial sums in float64, then the bins
        grouped = np.zeros(n_bins, np.float64)
        for lane in np.array_split(np.arange(n_terms), 64):
            run_cell, run_sum = -1, 0.0
            for k in lane.tolist():
                if cell[k] != run_cell:
                    if run_cell >= 0:
                        grouped[run_cell] += run_sum
                    run_cell, run_sum = int(cell[k]), 0.0
                run_sum += float(terms[k])
            if run_cell >= 0:
                grouped[run_cell] += run_sum
        assert np.array_equal(np.asarray(one_by_one, np.float32), grouped.astype(np.float32))
