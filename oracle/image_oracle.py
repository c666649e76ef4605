"""ORACLE / TEST INFRASTRUCTURE ONLY -- numpy restatement of the image preparation the
reference does with OpenCV before SIFT (scripts/lib/image.py:99-121,313):

    hsv = cv2.cvtColor(bgr, cv2.COLOR_BGR2HSV); v' = cv2.createCLAHE(3.0, (8,8)).apply(v)
    bgr' = cv2.cvtColor(merge(h, s, v'), cv2.COLOR_HSV2BGR); scaled = cv2.resize(bgr', (0,0), fx=s, fy=s)

PARITY UNPINNED (cv2 absent): the published 8-bit algorithms -- fixed-point HSV with 12-bit
division tables (H in [0,180)), CLAHE with clip + uniform redistribution and bilinear blending
of the 8x8 tile look-up tables, float HSV->BGR, 11-bit fixed-point bilinear resize with
dsize = round(size*scale) -- which the HIP kernels (csrc/image_prep.hip) are tested against.
"""
import numpy as np

HSV_SHIFT = 12


def _div_tables():
    sdiv = np.zeros(256, np.int64)
    hdiv = np.zeros(256, np.int64)
    i = np.arange(1, 256)
    sdiv[1:] = np.rint((255 << HSV_SHIFT) / (1.0 * i)).astype(np.int64)
    hdiv[1:] = np.rint((180 << HSV_SHIFT) / (6.0 * i)).astype(np.int64)
    return sdiv, hdiv


def bgr_to_hsv(bgr):
    sdiv, hdiv = _div_tables()
    b = bgr[..., 0].astype(np.int64)
    g = bgr[..., 1].astype(np.int64)
    r = bgr[..., 2].astype(np.int64)
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    s = (diff * sdiv[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = np.where(h < 0, h + 180, h)
    return np.stack([h, s, v], -1).astype(np.uint8)


def hsv_to_bgr(hsv):
    h = hsv[..., 0].astype(np.float32) * np.float32(6.0 / 180.0)
    s = hsv[..., 1].astype(np.float32) * np.float32(1.0 / 255.0)
    v = hsv[..., 2].astype(np.float32) * np.float32(1.0 / 255.0)
    sector = np.floor(h).astype(np.int64)
    f = h - sector.astype(np.float32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    f = np.where(bad, np.float32(0), f)
    t0 = v
    t1 = v * (np.float32(1) - s)
    t2 = v * (np.float32(1) - s * f)
    t3 = v * (np.float32(1) - s * (np.float32(1) - f))
    tab = np.stack([t0, t1, t2, t3], -1)
    sd = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    idx = sd[sector]                                         # [...,3] -> b, g, r
    out = np.take_along_axis(tab, idx, axis=-1)
    out = np.where((hsv[..., 1] == 0)[..., None], v[..., None], out)
    return np.clip(np.rint(out * np.float32(255.0)), 0, 255).astype(np.uint8)


def _reflect101(idx, n):
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    idx = np.abs(idx) % period
    return np.where(idx >= n, period - idx, idx)


def clahe(v, clip_limit=3.0, tiles=(8, 8)):
    """cv2.createCLAHE(clipLimit, tileGridSize).apply on uint8."""
    h, w = v.shape
    tx, ty = tiles
    pw = w if w % tx == 0 else w + (tx - w % tx)
    ph = h if h % ty == 0 else h + (ty - h % ty)
    src = v[_reflect101(np.arange(ph), h)][:, _reflect101(np.arange(pw), w)]
    tw, th = pw // tx, ph // ty
    area = tw * th
    clip = max(int(clip_limit * area / 256.0), 1)
    lut_scale = np.float32(255.0) / np.float32(area)
    luts = np.zeros((ty, tx, 256), np.uint8)
    for j in range(ty):
        for i in range(tx):
            hist = np.bincount(src[j * th:(j + 1) * th, i * tw:(i + 1) * tw].ravel(),
                               minlength=256).astype(np.int64)
            clipped = int(np.maximum(hist - clip, 0).sum())
            hist = np.minimum(hist, clip)
            batch = clipped // 256
            residual = clipped - batch * 256
            hist += batch
            if residual:
                step = max(256 // residual, 1)
                k = 0
                while k < 256 and residual > 0:
                    hist[k] += 1
                    k += step
                    residual -= 1
            cs = np.cumsum(hist).astype(np.float32)
            luts[j, i] = np.clip(np.rint(cs * lut_scale), 0, 255).astype(np.uint8)

    def coords(n, tile, ntiles):
        f = np.arange(n, dtype=np.float32) * np.float32(1.0 / tile) - np.float32(0.5)
        t1 = np.floor(f).astype(np.int64)
        a = f - t1.astype(np.float32)
        t2 = np.minimum(t1 + 1, ntiles - 1)
        t1 = np.maximum(t1, 0)
        return t1, t2, a.astype(np.float32)

    y1, y2, ya = coords(h, th, ty)
    x1, x2, xa = coords(w, tw, tx)
    vv = v.astype(np.int64)
    l11 = luts[y1[:, None], x1[None, :], vv].astype(np.float32)
    l12 = luts[y1[:, None], x2[None, :], vv].astype(np.float32)
    l21 = luts[y2[:, None], x1[None, :], vv].astype(np.float32)
    l22 = luts[y2[:, None], x2[None, :], vv].astype(np.float32)
    xa1 = np.float32(1) - xa
    ya1 = np.float32(1) - ya
    res = (l11 * xa1[None, :] + l12 * xa[None, :]) * ya1[:, None] + \
          (l21 * xa1[None, :] + l22 * xa[None, :]) * ya[:, None]
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def equalize_bgr(bgr):
    """scripts/lib/image.py:105-112"""
    hsv = bgr_to_hsv(bgr)
    hsv[..., 2] = clahe(hsv[..., 2])
    return hsv_to_bgr(hsv)


def resize_linear_u8(img, scale):
    """cv2.resize(img, (0,0), fx=scale, fy=scale) (INTER_LINEAR, uint8): dsize =
    round(size*scale); 11-bit fixed-point coefficients; ((b0*(row0>>4))>>16 + ... + 2) >> 2."""
    h, w = img.shape[:2]
    dw, dh = int(round(w * scale)), int(round(h * scale))

    def taps(n_src, n_dst):
        sc = float(n_src) / n_dst
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * sc - 0.5
        s0 = np.floor(f).astype(np.int64)
        t = (f - s0).astype(np.float32)
        lo = s0 < 0
        s0[lo], t[lo] = 0, 0.0
        hi = s0 >= n_src - 1
        s0[hi], t[hi] = n_src - 1, 0.0
        a1 = np.clip(np.rint(t * 2048.0), -32768, 32767).astype(np.int64)
        a0 = np.clip(np.rint((np.float32(1) - t) * 2048.0), -32768, 32767).astype(np.int64)
        return s0, np.minimum(s0 + 1, n_src - 1), a0, a1

    x0, x1, ax0, ax1 = taps(w, dw)
    y0, y1, ay0, ay1 = taps(h, dh)
    src = img.astype(np.int64)
    if src.ndim == 2:
        src = src[..., None]
    rows0 = src[y0][:, x0] * ax0[None, :, None] + src[y0][:, x1] * ax1[None, :, None]
    rows1 = src[y1][:, x0] * ax0[None, :, None] + src[y1][:, x1] * ax1[None, :, None]
    out = (((ay0[:, None, None] * (rows0 >> 4)) >> 16) + ((ay1[:, None, None] * (rows1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[..., 0] if img.ndim == 2 else out
