"""ORACLE / TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the SIFT detector +
descriptor that the reference obtains from OpenCV:

    detector = cv2.SIFT_create()                       scripts/lib/image.py:235-237
    kp_list, des_list = detector.detectAndCompute(scaled, None)        :324

PARITY UNPINNED: OpenCV (third party; environment.yml pins 4.0.1 but the code needs >= 4.4
for cv2.SIFT_create) is neither in /root/reference nor installed here, and the reference has
no SIFT fixtures.  This file restates the published algorithm (D. Lowe, IJCV 2004) with
OpenCV 4.x's defaults and conventions -- nfeatures=0, nOctaveLayers=3, contrastThreshold=0.04,
edgeThreshold=10, sigma=1.6, first octave -1 (image doubled, assumed pre-blur 0.5), border 5,
36-bin orientation histogram with 0.8 peak ratio, 4x4x8 descriptor with 0.2 clipping and
x512 u8 quantisation, keypoint `octave` packing octave | layer<<8 | round((xi+0.5)*255)<<16 --
and is what the HIP kernels (csrc/sift.hip) are tested against.  Agreement with a real
cv2.SIFT would have to be statistical (float filters differ in the last bits).

Arithmetic conventions (this file is their definition; oracle/sift_ref.c restates the hot loops
in C with OpenMP, and the *_py functions here are their numpy twins, compared in
tests/test_oracle.py): the Gaussian taps are accumulated with ONE rounding per tap, acc =
fma(v, k[t], acc) from 0 in ascending tap order -- a filter compiled for FMA hardware -- (rounds
1-2 rounded product and sum separately, the numpy expression `acc += v * k`); everything else in
the pyramid is separately rounded float32; refinement solves, orientation histograms and
descriptors are float64.
"""
import math

import numpy as np

F = np.float32
N_OCTAVE_LAYERS = 3
CONTRAST_THRESHOLD = 0.04
EDGE_THRESHOLD = 10.0
SIGMA = 1.6
IMG_BORDER = 5
MAX_INTERP_STEPS = 5
ORI_HIST_BINS = 36
ORI_SIG_FCTR = 1.5
ORI_RADIUS = 3 * ORI_SIG_FCTR
ORI_PEAK_RATIO = 0.8
DESCR_WIDTH = 4
DESCR_HIST_BINS = 8
DESCR_SCL_FCTR = 3.0
DESCR_MAG_THR = 0.2
INT_DESCR_FCTR = 512.0
FLT_EPSILON = 1.1920929e-07


def bgr_to_gray(bgr):
    """cv2.cvtColor(BGR2GRAY) on uint8: fixed point (B*1868 + G*9617 + R*4899 + 8192) >> 14."""
    b = bgr[..., 0].astype(np.int32)
    g = bgr[..., 1].astype(np.int32)
    r = bgr[..., 2].astype(np.int32)
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def resize_linear_2x(img):
    """cv2.resize(img, (2w, 2h), INTER_LINEAR) for float32: src = (dst + 0.5) / 2 - 0.5,
    clamped at the borders."""
    h, w = img.shape

    def taps(n_src, n_dst):
        f = (np.arange(n_dst, dtype=np.float32) + F(0.5)) * F(0.5) - F(0.5)
        i0 = np.floor(f).astype(np.int64)
        t = (f - i0.astype(np.float32)).astype(np.float32)
        lo = i0 < 0
        i0[lo], t[lo] = 0, 0.0
        hi = i0 >= n_src - 1
        i0[hi], t[hi] = n_src - 1, 0.0
        i1 = np.minimum(i0 + 1, n_src - 1)
        return i0, i1, t

    y0, y1, ty = taps(h, 2 * h)
    x0, x1, tx = taps(w, 2 * w)
    top = img[y0][:, x0] * (F(1) - tx) + img[y0][:, x1] * tx
    bot = img[y1][:, x0] * (F(1) - tx) + img[y1][:, x1] * tx
    return (top * (F(1) - ty)[:, None] + bot * ty[:, None]).astype(np.float32)


def gaussian_kernel(sigma):
    """cv2.GaussianBlur(img, Size(), sigma) for CV_32F: ksize = round(sigma*8 + 1) | 1,
    coefficients exp(-x^2 / 2 sigma^2) normalised to sum 1."""
    ksize = int(round(sigma * 8 + 1)) | 1
    r = ksize // 2
    x = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    total = 0.0
    for v in k.tolist():             # sequential double sum, first tap first (np.sum is pairwise:
        total += v                   # a different last bit would move taps by one float32 ulp)
    return (k / total).astype(np.float32)


def _reflect101(idx, n):
    """cv2 BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba), also for radii larger than the image."""
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    idx = np.abs(idx) % period
    return np.where(idx >= n, period - idx, idx)


def fma32(a, b, c):
    """float32 fused multiply-add a * b + c with ONE rounding (v_fma_f32 / fmaf), exactly: the
    product of two float32 is exact in float64; the float64 sum with c is rounded to 53 bits
    first, which only matters when it lands exactly on a float32 tie -- then the exact error of
    the sum (TwoSum) says on which side of the tie the true value lies."""
    p = a.astype(np.float64) * np.float64(b)
    c64 = np.asarray(c, np.float64)
    s = p + c64
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)                   # s + err = p + c exactly
    r = s.astype(np.float32)
    d = s - r.astype(np.float64)                        # exact
    other = np.nextafter(r, np.where(d > 0, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32))
    tie = (d != 0) & ((r.astype(np.float64) + other.astype(np.float64)) * 0.5 == s) & (err != 0)
    if np.any(tie):
        # ties-to-even picked r from s; the true value is s + err
        toward_other = np.sign(err) == np.sign(d)
        r = np.where(tie & toward_other, other, r)
        # (tie & ~toward_other: the true value is on r's side of the midpoint... unless
        #  ties-to-even had picked the far candidate, which float32(s) never does for r)
    return r.astype(np.float32)


def gaussian_blur_py(img, sigma):
    """separable, BORDER_REFLECT_101, fused multiply-add taps in ascending order (numpy twin of
    oracle/sift_ref.c blur_rows)."""
    k = gaussian_kernel(sigma)
    r = len(k) // 2
    h, w = img.shape
    xs = np.arange(w)
    tmp = np.zeros_like(img)
    for t in range(-r, r + 1):
        tmp = fma32(img[:, _reflect101(xs + t, w)], k[t + r], tmp)
    ys = np.arange(h)
    out = np.zeros_like(img)
    for t in range(-r, r + 1):
        out = fma32(tmp[_reflect101(ys + t, h), :], k[t + r], out)
    return out


def gaussian_blur(img, sigma):
    from . import cpu_ref
    return cpu_ref.sift_blur(img, gaussian_kernel(sigma))


def layer_sigmas():
    k = 2.0 ** (1.0 / N_OCTAVE_LAYERS)
    sig = [SIGMA]
    for i in range(1, N_OCTAVE_LAYERS + 3):
        sp = (k ** (i - 1)) * SIGMA
        st = sp * k
        sig.append(math.sqrt(st * st - sp * sp))
    return sig


def build_pyramids(bgr_or_gray):
    gray = bgr_to_gray(bgr_or_gray) if bgr_or_gray.ndim == 3 else bgr_or_gray
    base = resize_linear_2x(gray.astype(np.float32))
    sig_diff = math.sqrt(max(SIGMA * SIGMA - 4.0 * 0.5 * 0.5, 0.01))
    base = gaussian_blur(base, sig_diff)
    n_oct = int(round(math.log(min(base.shape)) / math.log(2.0) - 2)) + 1
    sig = layer_sigmas()
    gauss, dog = [], []
    for o in range(n_oct):
        lv = []
        for i in range(N_OCTAVE_LAYERS + 3):
            if o == 0 and i == 0:
                lv.append(base)
            elif i == 0:
                src = gauss[o - 1][N_OCTAVE_LAYERS]      # INTER_NEAREST to (w//2, h//2)
                lv.append(np.ascontiguousarray(src[:src.shape[0] // 2 * 2:2, :src.shape[1] // 2 * 2:2]))
            else:
                lv.append(gaussian_blur(lv[i - 1], sig[i]))
        gauss.append(lv)
        dog.append([lv[i + 1] - lv[i] for i in range(N_OCTAVE_LAYERS + 2)])
    return gauss, dog


def _solve3(A, b):
    """Gaussian elimination with partial pivoting (cv::Matx::solve(DECOMP_LU)): the operation
    sequence sift_ref.c and the device use, so the refined offsets agree to the last bit"""
    A = [[float(v) for v in row] for row in A]
    b = [float(v) for v in b]
    p = [0, 1, 2]
    for k in range(3):
        piv, best = k, abs(A[p[k]][k])
        for i in range(k + 1, 3):
            if abs(A[p[i]][k]) > best:
                best, piv = abs(A[p[i]][k]), i
        if best < 1e-300:
            return None
        p[k], p[piv] = p[piv], p[k]
        for i in range(k + 1, 3):
            f = A[p[i]][k] / A[p[k]][k]
            for j in range(k, 3):
                A[p[i]][j] -= f * A[p[k]][j]
            b[p[i]] -= f * b[p[k]]
    x = [0.0, 0.0, 0.0]
    for k in (2, 1, 0):
        s = b[p[k]]
        for j in range(k + 1, 3):
            s -= A[p[k]][j] * x[j]
        x[k] = s / A[p[k]][k]
    return x


def _adjust_local_extrema(dogs, layer, r, c):
    """3-D quadratic refinement; returns None or (layer, r, c, xi, xr, xc, contr)."""
    img_scale = F(1.0 / 255.0)
    deriv_scale = F(img_scale * 0.5)
    second_scale = img_scale
    cross_scale = F(img_scale * 0.25)
    h, w = dogs[0].shape
    xi = xr = xc = 0.0
    for it in range(MAX_INTERP_STEPS):
        img, prv, nxt = dogs[layer], dogs[layer - 1], dogs[layer + 1]
        dD = np.array([(img[r, c + 1] - img[r, c - 1]) * deriv_scale,
                       (img[r + 1, c] - img[r - 1, c]) * deriv_scale,
                       (nxt[r, c] - prv[r, c]) * deriv_scale], np.float32)
        v2 = img[r, c] * F(2)
        dxx = (img[r, c + 1] + img[r, c - 1] - v2) * second_scale
        dyy = (img[r + 1, c] + img[r - 1, c] - v2) * second_scale
        dss = (nxt[r, c] + prv[r, c] - v2) * second_scale
        dxy = (img[r + 1, c + 1] - img[r + 1, c - 1] - img[r - 1, c + 1] + img[r - 1, c - 1]) * cross_scale
        dxs = (nxt[r, c + 1] - nxt[r, c - 1] - prv[r, c + 1] + prv[r, c - 1]) * cross_scale
        dys = (nxt[r + 1, c] - nxt[r - 1, c] - prv[r + 1, c] + prv[r - 1, c]) * cross_scale
        H = np.array([[dxx, dxy, dxs], [dxy, dyy, dys], [dxs, dys, dss]], np.float32)
        X = _solve3(H.astype(np.float64), dD.astype(np.float64))
        if X is None:
            return None
        xc, xr, xi = -X[0], -X[1], -X[2]
        if abs(xi) < 0.5 and abs(xr) < 0.5 and abs(xc) < 0.5:
            break
        if abs(xi) > 2147483647 / 3 or abs(xr) > 2147483647 / 3 or abs(xc) > 2147483647 / 3:
            return None
        c += int(round(xc))
        r += int(round(xr))
        layer += int(round(xi))
        if layer < 1 or layer > N_OCTAVE_LAYERS or c < IMG_BORDER or c >= w - IMG_BORDER \
                or r < IMG_BORDER or r >= h - IMG_BORDER:
            return None
    else:
        return None
    img, prv, nxt = dogs[layer], dogs[layer - 1], dogs[layer + 1]
    dD = np.array([(img[r, c + 1] - img[r, c - 1]) * deriv_scale,
                   (img[r + 1, c] - img[r - 1, c]) * deriv_scale,
                   (nxt[r, c] - prv[r, c]) * deriv_scale], np.float64)
    t = dD[0] * xc + dD[1] * xr + dD[2] * xi
    contr = float(img[r, c]) * float(img_scale) + t * 0.5
    if abs(contr) * N_OCTAVE_LAYERS < CONTRAST_THRESHOLD:
        return None
    v2 = float(img[r, c]) * 2.0
    dxx = (float(img[r, c + 1]) + float(img[r, c - 1]) - v2) * float(second_scale)
    dyy = (float(img[r + 1, c]) + float(img[r - 1, c]) - v2) * float(second_scale)
    dxy = (float(img[r + 1, c + 1]) - float(img[r + 1, c - 1]) - float(img[r - 1, c + 1])
           + float(img[r - 1, c - 1])) * float(cross_scale)
    tr = dxx + dyy
    det = dxx * dyy - dxy * dxy
    if det <= 0 or tr * tr * EDGE_THRESHOLD >= (EDGE_THRESHOLD + 1) ** 2 * det:
        return None
    return layer, r, c, xi, xr, xc, contr


def _orientation_hist(img, r, c, radius, sigma):
    n = ORI_HIST_BINS
    h, w = img.shape
    expf_scale = -1.0 / (2.0 * sigma * sigma)
    hist = np.zeros(n, np.float64)
    for i in range(-radius, radius + 1):
        y = r + i
        if y <= 0 or y >= h - 1:
            continue
        for j in range(-radius, radius + 1):
            x = c + j
            if x <= 0 or x >= w - 1:
                continue
            dx = float(img[y, x + 1]) - float(img[y, x - 1])
            dy = float(img[y - 1, x]) - float(img[y + 1, x])
            wgt = math.exp((i * i + j * j) * expf_scale)
            ori = math.degrees(math.atan2(dy, dx)) % 360.0
            mag = math.sqrt(dx * dx + dy * dy)
            b = int(round((n / 360.0) * ori))
            if b >= n:
                b -= n
            if b < 0:
                b += n
            hist[b] += wgt * mag
    # circular smoothing [1 4 6 4 1] / 16
    t = np.concatenate([hist[-2:], hist, hist[:2]])
    sm = (t[:-4] + t[4:]) * (1.0 / 16) + (t[1:-3] + t[3:-1]) * (4.0 / 16) + t[2:-2] * (6.0 / 16)
    return sm


def detect(bgr_or_gray, use_c=True):
    """-> (keypoints [N,6] float64: x, y, size, angle, response, octave(packed int), pyramids).
    use_c: refinement + orientation through oracle/sift_ref.c (OpenMP), else the numpy twin."""
    gauss, dog = build_pyramids(bgr_or_gray)
    threshold = math.floor(0.5 * CONTRAST_THRESHOLD / N_OCTAVE_LAYERS * 255)
    kps = []
    for o, dogs in enumerate(dog):
        h, w = dogs[0].shape
        if h <= 2 * IMG_BORDER or w <= 2 * IMG_BORDER:
            continue
        cands = []
        for layer in range(1, N_OCTAVE_LAYERS + 1):
            cur = dogs[layer]
            core = cur[IMG_BORDER:h - IMG_BORDER, IMG_BORDER:w - IMG_BORDER]
            is_max = (np.abs(core) > threshold) & (core > 0)
            is_min = (np.abs(core) > threshold) & (core < 0)
            for dl in (-1, 0, 1):
                nb = dogs[layer + dl]
                for dr in (-1, 0, 1):
                    for dc in (-1, 0, 1):
                        if dl == 0 and dr == 0 and dc == 0:
                            continue
                        sh = nb[IMG_BORDER + dr:h - IMG_BORDER + dr, IMG_BORDER + dc:w - IMG_BORDER + dc]
                        is_max &= core >= sh
                        is_min &= core <= sh
            rr, cc = np.nonzero(is_max | is_min)
            if use_c:
                cands.append(np.stack([np.full(len(rr), layer), rr + IMG_BORDER, cc + IMG_BORDER], 1))
                continue
            for r, c in zip(rr + IMG_BORDER, cc + IMG_BORDER):
                res = _adjust_local_extrema(dogs, layer, int(r), int(c))
                if res is None:
                    continue
                l2, r2, c2, xi, xr, xc, contr = res
                size = SIGMA * (2.0 ** ((l2 + xi) / N_OCTAVE_LAYERS)) * (1 << o) * 2
                px = (c2 + xc) * (1 << o)
                py = (r2 + xr) * (1 << o)
                octave = o + (l2 << 8) + (int(round((xi + 0.5) * 255)) << 16)
                scl_octv = size * 0.5 / (1 << o)
                hist = _orientation_hist(gauss[o][l2], r2, c2, int(round(ORI_RADIUS * scl_octv)),
                                         ORI_SIG_FCTR * scl_octv)
                mag_thr = hist.max() * ORI_PEAK_RATIO
                n = ORI_HIST_BINS
                for j in range(n):
                    lft = hist[(j - 1) % n]
                    rgt = hist[(j + 1) % n]
                    if hist[j] > lft and hist[j] > rgt and hist[j] >= mag_thr:
                        b = j + 0.5 * (lft - rgt) / (lft - 2 * hist[j] + rgt)
                        b = b + n if b < 0 else (b - n if b >= n else b)
                        angle = 360.0 - (360.0 / n) * b
                        if abs(angle - 360.0) < FLT_EPSILON:
                            angle = 0.0
                        kps.append([px, py, size, angle, abs(contr), octave])
        if use_c and cands:
            from . import cpu_ref
            kps.extend(cpu_ref.sift_keypoints(dogs, gauss[o], o, np.concatenate(cands), SIGMA).tolist())
    kps = np.array(kps, np.float64).reshape(-1, 6)
    # first octave is -1: rescale to the input image (detectAndCompute)
    if len(kps):
        oc = kps[:, 5].astype(np.int64)
        oc = (oc & ~255) | ((oc - 1) & 255)
        kps[:, 5] = oc
        kps[:, 0] *= 0.5
        kps[:, 1] *= 0.5
        kps[:, 2] *= 0.5
    return kps, gauss


def unpack_octave(packed):
    packed = int(packed)
    octave = packed & 255
    layer = (packed >> 8) & 255
    if octave >= 128:
        octave |= -128
    scale = 1.0 / (1 << octave) if octave >= 0 else float(1 << -octave)
    return octave, layer, scale


def descriptor(img, ptx, pty, ori, scl):
    d, n = DESCR_WIDTH, DESCR_HIST_BINS
    h, w = img.shape
    px, py = int(round(ptx)), int(round(pty))
    cos_t = math.cos(math.radians(ori))
    sin_t = math.sin(math.radians(ori))
    bins_per_rad = n / 360.0
    exp_scale = -1.0 / (d * d * 0.5)
    hist_width = DESCR_SCL_FCTR * scl
    radius = int(round(hist_width * 1.4142135623730951 * (d + 1) * 0.5))
    radius = min(radius, int(math.sqrt(float(w) * w + float(h) * h)))
    cos_t /= hist_width
    sin_t /= hist_width
    hist = np.zeros((d + 2, d + 2, n + 2), np.float64)
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            c_rot = j * cos_t - i * sin_t
            r_rot = j * sin_t + i * cos_t
            rbin = r_rot + d / 2 - 0.5
            cbin = c_rot + d / 2 - 0.5
            r, c = py + i, px + j
            if not (-1 < rbin < d and -1 < cbin < d and 0 < r < h - 1 and 0 < c < w - 1):
                continue
            dx = float(img[r, c + 1]) - float(img[r, c - 1])
            dy = float(img[r - 1, c]) - float(img[r + 1, c])
            wgt = math.exp((c_rot * c_rot + r_rot * r_rot) * exp_scale)
            o = math.degrees(math.atan2(dy, dx)) % 360.0
            mag = math.sqrt(dx * dx + dy * dy) * wgt
            obin = (o - ori) * bins_per_rad
            r0, c0, o0 = math.floor(rbin), math.floor(cbin), math.floor(obin)
            fr, fc, fo = rbin - r0, cbin - c0, obin - o0
            if o0 < 0:
                o0 += n
            if o0 >= n:
                o0 -= n
            v_r1 = mag * fr
            v_r0 = mag - v_r1
            v_rc11 = v_r1 * fc
            v_rc10 = v_r1 - v_rc11
            v_rc01 = v_r0 * fc
            v_rc00 = v_r0 - v_rc01
            for (dr_, dc_, vv) in ((0, 0, v_rc00), (0, 1, v_rc01), (1, 0, v_rc10), (1, 1, v_rc11)):
                v1 = vv * fo
                hist[r0 + 1 + dr_, c0 + 1 + dc_, o0] += vv - v1
                hist[r0 + 1 + dr_, c0 + 1 + dc_, o0 + 1] += v1
    hist[:, :, 0] += hist[:, :, n]
    hist[:, :, 1] += hist[:, :, n + 1]
    dst = hist[1:d + 1, 1:d + 1, :n].reshape(-1)
    thr = math.sqrt(float((dst * dst).sum())) * DESCR_MAG_THR
    dst = np.minimum(dst, thr)
    nrm = INT_DESCR_FCTR / max(math.sqrt(float((dst * dst).sum())), FLT_EPSILON)
    return np.clip(np.rint(dst * nrm), 0, 255).astype(np.uint8)


def detect_and_compute(bgr_or_gray, use_c=True):
    """-> keypoints [N,6] (x, y, size, angle, response, packed octave), descriptors [N,128] u8,
    sorted canonically by (octave, layer, y, x, angle).  use_c: the per-keypoint loops through
    oracle/sift_ref.c (OpenMP), else their numpy twins (small images only)."""
    kps, gauss = detect(bgr_or_gray, use_c=use_c)
    des = np.zeros((len(kps), 128), np.uint8)
    # cv2.KeyPoint fields are float32: the descriptor stage sees the rounded values
    kps[:, :5] = kps[:, :5].astype(np.float32).astype(np.float64)
    par = np.zeros((len(kps), 4))
    level = np.zeros((len(kps), 2), np.int64)
    for k, (x, y, size, angle, resp, packed) in enumerate(kps):
        octave, layer, scale = unpack_octave(packed)
        a = 360.0 - angle
        if abs(a - 360.0) < FLT_EPSILON:
            a = 0.0
        par[k] = (x * scale, y * scale, a, size * scale * 0.5)
        level[k] = (octave + 1, layer)
        if not use_c:
            des[k] = descriptor(gauss[octave + 1][layer], *par[k])
    if use_c and len(kps):
        from . import cpu_ref
        for o, layer in sorted(set(map(tuple, level.tolist()))):
            sel = np.nonzero((level[:, 0] == o) & (level[:, 1] == layer))[0]
            des[sel] = cpu_ref.sift_descriptors(gauss[o][layer], par[sel])
    if len(kps):
        oc = kps[:, 5].astype(np.int64)
        order = np.lexsort((des[:, 0], kps[:, 3], kps[:, 0], kps[:, 1], (oc >> 8) & 255,
                            ((oc & 255) + 1) & 255))
        kps, des = kps[order], des[order]
    return kps, des


def detect_and_compute_c(bgr_or_gray):
    """detect_and_compute() with nothing but oracle/sift_ref.c underneath (OpenMP): the form
    bench.py times as the CPU baseline of the SIFT section"""
    from . import cpu_ref
    gray = bgr_to_gray(bgr_or_gray) if bgr_or_gray.ndim == 3 else bgr_or_gray
    kps, des = cpu_ref.sift_detect(gray)
    if len(kps):
        oc = kps[:, 5].astype(np.int64)
        order = np.lexsort((des[:, 0], kps[:, 3], kps[:, 0], kps[:, 1], (oc >> 8) & 255,
                            ((oc & 255) + 1) & 255))
        kps, des = kps[order], des[order]
    return kps, des
