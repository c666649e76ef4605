"""GPU: the SIFT kernels (csrc/sift.hip) against the CPU restatement of the same published
algorithm (oracle/sift_oracle.py).  PARITY UNPINNED against cv2 (absent); the bar here is
agreement with the oracle: same keypoint set, descriptors equal up to u8 rounding ties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def texture(h, w, seed=0):
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.float64)
    for s in (2, 4, 8, 16, 32):
        g = rng.normal(size=(h // s + 2, w // s + 2))
        img += np.kron(g, np.ones((s, s)))[:h, :w] * s ** 0.7
    k = np.array([1, 4, 6, 4, 1.]) / 16
    img = np.apply_along_axis(lambda r: np.convolve(r, k, 'same'), 1, img)
    img = np.apply_along_axis(lambda r: np.convolve(r, k, 'same'), 0, img)
    img = (img - img.min()) / (img.max() - img.min()) * 255
    return np.stack([img, img * 0.9 + 10, img * 0.8 + 20], 2).clip(0, 255).astype(np.uint8)


def _match(ka, oa, kb, ob):
    """greedy one-to-one matching of keypoints with identical packed octave"""
    used = np.zeros(len(kb), bool)
    pairs = []
    for i in range(len(ka)):
        cand = np.nonzero((ob == oa[i]) & ~used & (np.abs(kb[:, 0] - ka[i, 0]) < 0.02)
                          & (np.abs(kb[:, 1] - ka[i, 1]) < 0.02))[0]
        if len(cand) == 0:
            continue
        da = np.abs(((kb[cand, 3] - ka[i, 3]) + 180.0) % 360.0 - 180.0)
        j = cand[np.argmin(da)]
        if da.min() < 0.2:
            used[j] = True
            pairs.append((i, j))
    return np.array(pairs).reshape(-1, 2)


@pytest.mark.parametrize('shape,seed,gray', [((200, 260), 0, False), ((167, 301), 3, False),
                                             ((240, 180), 5, True)])
def test_sift_equals_oracle(shape, seed, gray):
    from imageanalysis_amd import kernels
    from oracle import sift_oracle as so
    img = texture(shape[0], shape[1], seed)
    if gray:
        img = so.bgr_to_gray(img)
    kps, des = so.detect_and_compute(img)
    kp, octv, d = kernels.sift_detect(img)
    assert len(kps) > 100
    assert abs(len(kp) - len(kps)) <= max(2, len(kps) // 100)
    pairs = _match(kps, kps[:, 5].astype(np.int64), kp.astype(np.float64), octv.astype(np.int64))
    assert len(pairs) >= 0.99 * len(kps)
    a, b = kps[pairs[:, 0]], kp[pairs[:, 1]].astype(np.float64)
    assert np.abs(a[:, 0] - b[:, 0]).max() < 2e-3 and np.abs(a[:, 1] - b[:, 1]).max() < 2e-3
    assert np.abs(a[:, 2] - b[:, 2]).max() < 1e-3 * a[:, 2].max()
    assert np.abs(a[:, 4] - b[:, 4]).max() < 1e-5
    da, db = des[pairs[:, 0]].astype(int), d[pairs[:, 1]].astype(int)
    diff = np.abs(da - db)
    assert diff.max() <= 2 and (diff == 0).mean() > 0.99
    # descriptor invariants of the format (x512, clip 0.2): u8, L2 norm ~ 512
    nrm = np.sqrt((d.astype(float) ** 2).sum(1))
    assert np.all(np.abs(nrm - 512) < 40)


def test_sift_translation_repeatability():
    """a crop shifted by whole pixels yields the same features shifted (interior ones)."""
    from imageanalysis_amd import kernels
    big = texture(260, 320, 9)
    a = big[10:250, 10:310]
    b = big[18:258, 26:326]                     # content of b at (x,y) = a at (x+16, y+8)
    ka, oa, da = kernels.sift_detect(a)
    kb, ob, db = kernels.sift_detect(b)
    inner = (ka[:, 0] > 60) & (ka[:, 0] < 240) & (ka[:, 1] > 50) & (ka[:, 1] < 190) & (ka[:, 2] < 8)
    ka2 = ka[inner].astype(np.float64)
    ka2[:, 0] -= 16
    ka2[:, 1] -= 8
    pairs = _match(ka2, oa[inner].astype(np.int64), kb.astype(np.float64), ob.astype(np.int64))
    assert len(pairs) >= 0.85 * inner.sum()
    dd = np.abs(da[inner][pairs[:, 0]].astype(int) - db[pairs[:, 1]].astype(int))
    assert np.median(dd.sum(1)) < 60


@pytest.mark.parametrize('k', [1, 2, 3])
def test_sift_rotation_invariance(k):
    """np.rot90 is an exact pixel permutation: the same scene must give the same keypoints at
    the rotated positions, orientations turned by k*90 degrees and (orientation-normalised)
    descriptors that agree -- what matching between the opposite headings of a lawn-mower
    survey relies on."""
    from imageanalysis_amd import kernels
    img = texture(240, 240, 11)[:, :, 0].copy()
    rot = np.ascontiguousarray(np.rot90(img, k))
    ka, oa, da = kernels.sift_detect(img)
    kb, ob, db = kernels.sift_detect(rot)
    assert len(ka) > 150 and abs(len(ka) - len(kb)) <= len(ka) // 10
    n = img.shape[0]
    # pixel (x, y) of img lands at rot90^k: k=1 -> (y, n-1-x), k=2 -> (n-1-x, n-1-y), k=3 -> (n-1-y, x)
    x, y = ka[:, 0].astype(np.float64), ka[:, 1].astype(np.float64)
    xr, yr = {1: (y, n - 1 - x), 2: (n - 1 - x, n - 1 - y), 3: (n - 1 - y, x)}[k]
    matched, dist, dang = 0, [], []
    for i in range(len(ka)):
        d2 = (kb[:, 0] - xr[i]) ** 2 + (kb[:, 1] - yr[i]) ** 2
        # (cv2 convention: no half-pixel shift for octave -1 => up to 0.5 px between the frames;
        #  the packed octave also carries the sub-layer offset in its top bits: compare octave+layer)
        cand = np.nonzero((d2 < 2.0) & ((ob & 0xFFFF) == (oa[i] & 0xFFFF)))[0]
        if len(cand) == 0:
            continue
        # image rotated counter-clockwise by k*90 deg; keypoint angles are clockwise (y down)
        want = (ka[i, 3] - 90.0 * k) % 360.0
        da_ = np.abs(((kb[cand, 3] - want) + 180.0) % 360.0 - 180.0)
        j = cand[np.argmin(da_)]
        if da_.min() < 3.0:
            matched += 1
            dang.append(da_.min())
            dist.append(np.sqrt(((da[i].astype(float) - db[j].astype(float)) ** 2).sum()))
    print('rot90 x%d: %d / %d keypoints re-found, median angle error %.3f deg, descriptor distance '
          'median %.1f p90 %.1f' % (k, matched, len(ka), np.median(dang), np.median(dist),
                                  np.percentile(dist, 90)))
    assert matched >= 0.8 * len(ka), (matched, len(ka))
    # |desc| = 512, unrelated descriptors are ~500 apart; the half-pixel frame shift costs a little
    assert np.median(dist) < 80 and np.percentile(dist, 90) < 200
    # and they are each other's nearest neighbours among all descriptors
    from oracle import cpu_ref
    idx, dd = cpu_ref.knn2_l2_u8(da, db)
    ratio_ok = (np.sqrt(dd[:, 0].astype(float)) < 0.6 * np.sqrt(dd[:, 1].astype(float))).mean()
    assert ratio_ok > 0.6


def _device_level(img_shape, ws, octave, kind, index):
    import ctypes
    from imageanalysis_amd import _lib
    off, lh, lw, no = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib().iamx_sift_pyramid_level(img_shape[0], img_shape[1], octave, kind, index,
                                                  ctypes.byref(off), ctypes.byref(lh), ctypes.byref(lw),
                                                  ctypes.byref(no)), 'iamx_sift_pyramid_level')
    n = lh.value * lw.value
    lvl = ws[off.value:off.value + 4 * n].view(dtype=__import__('torch').float32)
    return lvl.cpu().numpy().reshape(lh.value, lw.value), no.value


def _match_sorted(ka, oa, kb, ob):
    """_match() for whole frames: both lists are in the canonical (octave, layer, y, x, angle)
    order, so a keypoint's partner is within a few positions of where a merge would put it"""
    seg = lambda o: (((o & 255) + 1) & 255) * 4 + ((o >> 8) & 255)
    sa, sb = seg(oa), seg(ob)
    pairs = []
    for s_ in np.unique(sa):
        ia, ib = np.nonzero(sa == s_)[0], np.nonzero(sb == s_)[0]
        if len(ib) == 0:
            continue
        yb = kb[ib, 1]
        lo = np.searchsorted(yb, ka[ia, 1] - 0.02, 'left')
        hi = np.searchsorted(yb, ka[ia, 1] + 0.02, 'right')
        used = np.zeros(len(ib), bool)
        for i, a, b in zip(ia, lo, hi):
            cand = np.arange(a, b)
            cand = cand[(~used[cand]) & (np.abs(kb[ib[cand], 0] - ka[i, 0]) < 0.02) & (ob[ib[cand]] == oa[i])]
            if len(cand) == 0:
                continue
            da = np.abs(((kb[ib[cand], 3] - ka[i, 3]) + 180.0) % 360.0 - 180.0)
            if da.min() < 0.2:
                j = cand[np.argmin(da)]
                used[j] = True
                pairs.append((i, ib[j]))
    return np.array(pairs).reshape(-1, 2)


def test_config_size_pyramid_bit_equal_and_whole_frame():
    """BASELINE configs[1] detect size (5472x3648 at scale 0.4 -> 2189x1459): every Gaussian and
    DoG level of every octave is BIT-identical to the oracle's float32 pyramid; keypoints and
    descriptors are compared on the WHOLE frame (the oracle's per-keypoint loops run in C,
    oracle/sift_ref.c) and, as before, on four crops; the share of descriptor bytes the float32
    exp / atan2 of the device moves is reported (the oracle is float64 throughout)."""
    from imageanalysis_amd import kernels
    from oracle import sift_oracle as so
    frame = texture(1459, 2189, 21)
    gray = so.bgr_to_gray(frame)
    kp, octv, d = kernels.sift_detect(gray)
    assert len(kp) > 5000
    # canonical order (octave, layer, y, x, angle): strictly ascending keys
    oi = ((octv & 255) + 1) & 255
    key = np.stack([oi * 4 + ((octv >> 8) & 255), kp[:, 1].view(np.int32), kp[:, 0].view(np.int32),
                    kp[:, 3].view(np.int32)], 1).astype(np.int64)
    order = np.lexsort((d[:, 0], key[:, 3], key[:, 2], key[:, 1], key[:, 0]))
    assert np.array_equal(order, np.arange(len(kp)))
    import torch
    ws = kernels._sift_ws[(torch.cuda.current_device(), 0)]
    gauss, dog = so.build_pyramids(gray)
    _lvl, n_oct = _device_level(gray.shape, ws, 0, 0, 0)
    assert n_oct == len(gauss) >= 10
    for o in range(n_oct):
        for i in range(6):
            if o == 0 and i == 0:
                pass                                   # (also checked: the doubled, pre-blurred base)
            got, _ = _device_level(gray.shape, ws, o, 0, i)
            assert got.shape == gauss[o][i].shape and np.array_equal(got, gauss[o][i]), ('gauss', o, i)
        for i in range(5):
            if o == 0 and i == 0:
                continue                               # DoG 0 of octave 0 doubles as scratch before it is written
            got, _ = _device_level(gray.shape, ws, o, 1, i)
            assert np.array_equal(got, dog[o][i]), ('dog', o, i)
    got, _ = _device_level(gray.shape, ws, 0, 1, 0)
    assert np.array_equal(got, dog[0][0])
    # keypoints / descriptors of the whole frame, one to one
    kps, des = so.detect_and_compute(gray)
    assert len(kps) > 5000 and abs(len(kp) - len(kps)) <= len(kps) // 200
    pairs = _match_sorted(kps, kps[:, 5].astype(np.int64), kp.astype(np.float64), octv.astype(np.int64))
    assert len(pairs) >= 0.995 * len(kps), (len(pairs), len(kps), len(kp))
    a, b = kps[pairs[:, 0]], kp[pairs[:, 1]].astype(np.float64)
    assert np.abs(a[:, 0] - b[:, 0]).max() < 2e-3 and np.abs(a[:, 1] - b[:, 1]).max() < 2e-3
    assert np.abs(a[:, 2] - b[:, 2]).max() < 1e-3 * a[:, 2].max()
    assert np.abs(a[:, 4] - b[:, 4]).max() < 1e-5
    diff = np.abs(des[pairs[:, 0]].astype(int) - d[pairs[:, 1]].astype(int))
    print('whole frame: %d oracle / %d device keypoints, %d paired; descriptor bytes that differ: '
          '%.4f %%, max %d' % (len(kps), len(kp), len(pairs), 100.0 * (diff != 0).mean(), diff.max()))
    assert diff.max() <= 2 and (diff != 0).mean() < 0.01
    # keypoints / descriptors on crops of the frame
    moved, total = 0, 0
    for (y0, x0) in ((0, 0), (500, 900), (1159, 1888), (300, 1500)):
        crop = np.ascontiguousarray(gray[y0:y0 + 300, x0:x0 + 301])
        kps, des = so.detect_and_compute(crop)
        ck, co, cd = kernels.sift_detect(crop)
        assert len(kps) > 100 and abs(len(ck) - len(kps)) <= max(2, len(kps) // 100)
        pairs = _match(kps, kps[:, 5].astype(np.int64), ck.astype(np.float64), co.astype(np.int64))
        assert len(pairs) >= 0.99 * len(kps)
        diff = np.abs(des[pairs[:, 0]].astype(int) - cd[pairs[:, 1]].astype(int))
        assert diff.max() <= 2
        moved += int((diff != 0).sum())
        total += diff.size
    print('descriptor bytes that differ from the float64 oracle: %d of %d (%.4f %%)'
          % (moved, total, 100.0 * moved / total))
    assert moved / total < 0.01
