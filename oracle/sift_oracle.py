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
the pyramid is separately rounded float32.  From round 4 on everything behind the pyramid follows
OpenCV's float32 scalar code paths operation by operation (sift.simd.hpp adjustLocalExtrema with
Matx33f::solve = Cramer's rule, calcOrientationHist, calcSIFTDescriptor, hal::fastAtan2's
polynomial, KeyPointsFilter::removeDuplicatedSorted and its output order), with three stated
exceptions: the Gaussian weights use exp32() below instead of hal::exp32f's
table, cosf / sinf / powf are the correctly rounded float32 values, and every histogram bin is
the exact sum of OpenCV's float32 terms rounded once instead of a running float32 sum.
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


# --------------------------------------------------------------------------------------------
# float32 scalar conventions of OpenCV's sift.simd.hpp / mathfuncs (every operation below is ONE
# IEEE float32 operation unless it says float64; np.float32 scalars keep python floats "weak")
# --------------------------------------------------------------------------------------------
_DBL_EPS_F = F(2.220446049250313e-16)
_ATAN2_P1 = F(F(0.9997878412794807) * F(57.29577951308232))
_ATAN2_P3 = F(F(-0.3258083974640975) * F(57.29577951308232))
_ATAN2_P5 = F(F(0.1555786518463281) * F(57.29577951308232))
_ATAN2_P7 = F(F(-0.04432655554792128) * F(57.29577951308232))


def fast_atan2(y, x):
    """cv::fastAtan2 (degrees, [0, 360]; scalar form of hal::fastAtan32f, mathfuncs_core): a
    7th-order odd polynomial, good to ~0.3 deg -- what calcOrientationHist and calcSIFTDescriptor
    bin their gradients with."""
    y, x = F(y), F(x)
    ax, ay = F(abs(x)), F(abs(y))
    if ax >= ay:
        c = F(ay / F(ax + _DBL_EPS_F))
        c2 = F(c * c)
        a = F(F(F(F(F(F(F(_ATAN2_P7 * c2) + _ATAN2_P5) * c2) + _ATAN2_P3) * c2) + _ATAN2_P1) * c)
    else:
        c = F(ax / F(ay + _DBL_EPS_F))
        c2 = F(c * c)
        a = F(F(90.0) - F(F(F(F(F(F(F(_ATAN2_P7 * c2) + _ATAN2_P5) * c2) + _ATAN2_P3) * c2) + _ATAN2_P1) * c))
    if x < 0:
        a = F(F(180.0) - a)
    if y < 0:
        a = F(F(360.0) - a)
    return a


_LOG2E = 1.4426950408889634
_EXP_C = [F(0.6931471805599453 ** k / math.factorial(k)) for k in range(8)]


def exp32(x):
    """float32 exp of a float32 argument <= 0 (the Gaussian weights): range reduction in
    float64 (t = x log2 e, n = rint(t), f = float32(t - n)), 2^f by a degree-7 Horner polynomial
    in separately rounded float32, scaled by 2^n.  Within ~1 ulp of the true value like any expf;
    OpenCV's hal::exp32f is a 64-entry table + cubic of the same accuracy whose constants are not
    restated here."""
    x = F(x)
    if x < F(-87.0):
        return F(0.0)
    t = float(x) * _LOG2E
    n = float(np.rint(t))
    f = F(t - n)
    p = _EXP_C[7]
    for k in range(6, -1, -1):
        p = F(F(p * f) + _EXP_C[k])
    return F(math.ldexp(float(p), int(n)))


def cv_round(v):
    """cvRound: nearest integer, ties to even (lrint)"""
    return int(np.rint(v))


def _solve3_cramer(a, b):
    """Matx33f::solve(Vec3f, DECOMP_LU): for 3 x 3 with one right-hand side OpenCV's
    Matx_FastSolveOp specialisation ignores the method and applies Cramer's rule in float32;
    a singular matrix (determinant exactly 0) yields the zero vector."""
    det = F(F(F(a[0][0] * F(F(a[1][1] * a[2][2]) - F(a[2][1] * a[1][2])))
              - F(a[0][1] * F(F(a[1][0] * a[2][2]) - F(a[2][0] * a[1][2]))))
            + F(a[0][2] * F(F(a[1][0] * a[2][1]) - F(a[2][0] * a[1][1]))))
    if det == 0:
        return [F(0), F(0), F(0)]
    d = F(F(1.0) / det)
    x0 = F(d * F(F(F(b[0] * F(F(a[1][1] * a[2][2]) - F(a[1][2] * a[2][1])))
                   - F(a[0][1] * F(F(b[1] * a[2][2]) - F(a[1][2] * b[2]))))
                 + F(a[0][2] * F(F(b[1] * a[2][1]) - F(a[1][1] * b[2])))))
    x1 = F(d * F(F(F(a[0][0] * F(F(b[1] * a[2][2]) - F(a[1][2] * b[2])))
                   - F(b[0] * F(F(a[1][0] * a[2][2]) - F(a[1][2] * a[2][0]))))
                 + F(a[0][2] * F(F(a[1][0] * b[2]) - F(b[1] * a[2][0])))))
    x2 = F(d * F(F(F(a[0][0] * F(F(a[1][1] * b[2]) - F(b[1] * a[2][1])))
                   - F(a[0][1] * F(F(a[1][0] * b[2]) - F(b[1] * a[2][0]))))
                 + F(b[0] * F(F(a[1][0] * a[2][1]) - F(a[1][1] * a[2][0])))))
    return [x0, x1, x2]


def _adjust_local_extrema(dogs, layer, r, c):
    """adjustLocalExtrema (sift.simd.hpp): 3-D quadratic refinement, contrast and edge tests, all
    float32.  Returns None or (layer, r, c, xi, xr, xc, contr) with float32 offsets."""
    img_scale = F(1.0 / 255.0)
    deriv_scale = F(img_scale * F(0.5))
    second_scale = img_scale
    cross_scale = F(img_scale * F(0.25))
    h, w = dogs[0].shape
    xi = xr = xc = F(0)
    for it in range(MAX_INTERP_STEPS):
        img, prv, nxt = dogs[layer], dogs[layer - 1], dogs[layer + 1]
        dD = [F(F(img[r, c + 1] - img[r, c - 1]) * deriv_scale),
              F(F(img[r + 1, c] - img[r - 1, c]) * deriv_scale),
              F(F(nxt[r, c] - prv[r, c]) * deriv_scale)]
        v2 = F(img[r, c] * F(2))
        dxx = F(F(F(img[r, c + 1] + img[r, c - 1]) - v2) * second_scale)
        dyy = F(F(F(img[r + 1, c] + img[r - 1, c]) - v2) * second_scale)
        dss = F(F(F(nxt[r, c] + prv[r, c]) - v2) * second_scale)
        dxy = F(F(F(F(img[r + 1, c + 1] - img[r + 1, c - 1]) - img[r - 1, c + 1]) + img[r - 1, c - 1]) * cross_scale)
        dxs = F(F(F(F(nxt[r, c + 1] - nxt[r, c - 1]) - prv[r, c + 1]) + prv[r, c - 1]) * cross_scale)
        dys = F(F(F(F(nxt[r + 1, c] - nxt[r - 1, c]) - prv[r + 1, c]) + prv[r - 1, c]) * cross_scale)
        X = _solve3_cramer([[dxx, dxy, dxs], [dxy, dyy, dys], [dxs, dys, dss]], dD)
        xc, xr, xi = F(-X[0]), F(-X[1]), F(-X[2])
        if abs(xi) < F(0.5) and abs(xr) < F(0.5) and abs(xc) < F(0.5):
            break
        big = F(2147483647 // 3)
        if abs(xi) > big or abs(xr) > big or abs(xc) > big:
            return None
        c += cv_round(xc)
        r += cv_round(xr)
        layer += cv_round(xi)
        if layer < 1 or layer > N_OCTAVE_LAYERS or c < IMG_BORDER or c >= w - IMG_BORDER \
                or r < IMG_BORDER or r >= h - IMG_BORDER:
            return None
    else:
        return None
    img, prv, nxt = dogs[layer], dogs[layer - 1], dogs[layer + 1]
    dD = [F(F(img[r, c + 1] - img[r, c - 1]) * deriv_scale),
          F(F(img[r + 1, c] - img[r - 1, c]) * deriv_scale),
          F(F(nxt[r, c] - prv[r, c]) * deriv_scale)]
    t = F(F(F(F(0) + F(dD[0] * xc)) + F(dD[1] * xr)) + F(dD[2] * xi))           # Matx::dot
    contr = F(F(img[r, c] * img_scale) + F(t * F(0.5)))
    if F(abs(contr) * F(N_OCTAVE_LAYERS)) < F(CONTRAST_THRESHOLD):
        return None
    v2 = F(img[r, c] * F(2))
    dxx = F(F(F(img[r, c + 1] + img[r, c - 1]) - v2) * second_scale)
    dyy = F(F(F(img[r + 1, c] + img[r - 1, c]) - v2) * second_scale)
    dxy = F(F(F(F(img[r + 1, c + 1] - img[r + 1, c - 1]) - img[r - 1, c + 1]) + img[r - 1, c - 1]) * cross_scale)
    tr = F(dxx + dyy)
    det = F(F(dxx * dyy) - F(dxy * dxy))
    e = F(EDGE_THRESHOLD)
    if det <= 0 or F(F(tr * tr) * e) >= F(F(F(e + F(1)) * F(e + F(1))) * det):
        return None
    return layer, r, c, xi, xr, xc, contr


def _keypoint_fields(o, layer, r, c, xi, xr, xc, contr):
    """the KeyPoint adjustLocalExtrema fills in (float32 fields, octave-0 = doubled image
    coordinates): x, y, size, |contrast|, packed octave"""
    scale = F(1 << o)
    px = F(F(F(c) + xc) * scale)
    py = F(F(F(r) + xr) * scale)
    octave = o + (layer << 8) + (cv_round((float(xi) + 0.5) * 255) << 16)
    e = F(F(F(layer) + xi) / F(N_OCTAVE_LAYERS))
    size = F(F(F(F(SIGMA) * F(2.0 ** float(e))) * scale) * F(2))              # powf(2.f, e)
    return px, py, size, F(abs(contr)), octave


def _bin_sums(n_bins, bins, terms):
    """per-bin sums of float32 terms, each sum rounded to float32 ONCE: the order-independent
    form of OpenCV's running float32 sums (`hist[bin] += term` in scan order) that a parallel
    implementation can reproduce -- float64 accumulation of float32 terms is exact for these
    counts and magnitudes"""
    acc = np.zeros(n_bins, np.float64)
    np.add.at(acc, np.asarray(bins, np.int64), np.asarray(terms, np.float64))
    return acc.astype(np.float32)


def _orientation_hist(img, r, c, radius, sigma):
    """calcOrientationHist: raw 36-bin histogram of W * Mag, circular [1 4 6 4 1] / 16 smoothing"""
    n = ORI_HIST_BINS
    h, w = img.shape
    sigma = F(sigma)
    expf_scale = F(F(-1.0) / F(F(F(2.0) * sigma) * sigma))
    bins, terms = [], []
    for i in range(-radius, radius + 1):
        y = r + i
        if y <= 0 or y >= h - 1:
            continue
        for j in range(-radius, radius + 1):
            x = c + j
            if x <= 0 or x >= w - 1:
                continue
            dx = F(img[y, x + 1] - img[y, x - 1])
            dy = F(img[y - 1, x] - img[y + 1, x])
            wgt = exp32(F(F(i * i + j * j) * expf_scale))
            ori = fast_atan2(dy, dx)
            mag = F(np.sqrt(F(F(dx * dx) + F(dy * dy))))
            b = cv_round(F(F(n / 360.0) * ori))
            if b >= n:
                b -= n
            if b < 0:
                b += n
            bins.append(b)
            terms.append(F(wgt * mag))
    t = _bin_sums(n, bins, terms)
    sm = np.zeros(n, np.float32)
    for i in range(n):
        sm[i] = F(F(F(F(t[i - 2] + t[(i + 2) % n]) * F(1.0 / 16)) + F(F(t[i - 1] + t[(i + 1) % n]) * F(4.0 / 16)))
                  + F(t[i] * F(6.0 / 16)))
    return sm


def _orientation_peaks(hist):
    """the peak search of findScaleSpaceExtrema: float32 angles (degrees) in bin order"""
    n = ORI_HIST_BINS
    mag_thr = F(hist.max() * F(ORI_PEAK_RATIO))
    out = []
    for j in range(n):
        lft, rgt = hist[(j - 1) % n], hist[(j + 1) % n]
        if hist[j] > lft and hist[j] > rgt and hist[j] >= mag_thr:
            b = F(F(j) + F(F(F(0.5) * F(lft - rgt)) / F(F(lft - F(F(2) * hist[j])) + rgt)))
            b = F(F(n) + b) if b < 0 else (F(b - F(n)) if b >= n else b)
            angle = F(F(360.0) - F(F(360.0 / n) * b))
            if abs(F(angle - F(360.0))) < F(FLT_EPSILON):
                angle = F(0)
            out.append(angle)
    return out


def detect(bgr_or_gray, use_c=True):
    """findScaleSpaceExtrema + the first-octave rescale of detectAndCompute -> (keypoints [N,6]
    float64 holding float32 values: x, y, size, angle, response, octave(packed int) in the
    order candidates are visited (octave, layer, row major, peak) -- NOT yet de-duplicated or
    sorted, see remove_duplicated_sorted() --, Gaussian pyramid).
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
                px, py, size, resp, octave = _keypoint_fields(o, l2, r2, c2, xi, xr, xc, contr)
                scl_octv = F(F(size * F(0.5)) / F(1 << o))
                hist = _orientation_hist(gauss[o][l2], r2, c2, cv_round(F(F(ORI_RADIUS) * scl_octv)),
                                         F(F(ORI_SIG_FCTR) * scl_octv))
                for angle in _orientation_peaks(hist):
                    kps.append([px, py, size, angle, resp, octave])
        if use_c and cands:
            from . import cpu_ref
            kps.extend(cpu_ref.sift_keypoints(dogs, gauss[o], o, np.concatenate(cands), SIGMA).tolist())
    kps = np.array(kps, np.float64).reshape(-1, 6)
    # first octave is -1: rescale to the input image (detectAndCompute; x 0.5 is exact in float32)
    if len(kps):
        oc = kps[:, 5].astype(np.int64)
        oc = (oc & ~255) | ((oc - 1) & 255)
        kps[:, 5] = oc
        kps[:, 0] *= 0.5
        kps[:, 1] *= 0.5
        kps[:, 2] *= 0.5
    return kps, gauss


def opencv_sort_keys(kps):
    """KeyPoint12_LessThan of KeyPointsFilter::removeDuplicatedSorted (features2d/keypoint.cpp):
    x, y ascending, size DESCENDING, angle ascending, response DESCENDING, octave DESCENDING (the
    packed value BEFORE detectAndCompute's first-octave adjustment; class_id is -1 everywhere).
    Returns the np.lexsort key tuple (last key most significant)."""
    oc = kps[:, 5].astype(np.int64)
    pre = (oc & ~255) | ((oc + 1) & 255)
    return (-pre, -kps[:, 4], kps[:, 3], -kps[:, 2], kps[:, 1], kps[:, 0])


def remove_duplicated_sorted(kps):
    """KeyPointsFilter::removeDuplicatedSorted: sort with KeyPoint12_LessThan, then drop every
    keypoint equal to the last KEPT one in (x, y, size, angle).  -> (kept keypoints in OpenCV's
    output order, indices into the input, number removed)"""
    if len(kps) < 2:
        return kps, np.arange(len(kps)), 0
    order = np.lexsort(opencv_sort_keys(kps))
    s = kps[order]
    same = np.all(s[1:, :4] == s[:-1, :4], axis=1)       # equal to its predecessor => to the kept one
    keep = np.concatenate([[True], ~same])
    return s[keep], order[keep], int((~keep).sum())


def canonical_order(kps, des=None):
    """the (octave, layer, y, x, angle) order of rounds 1-3 (pyramid-local: what the device's
    descriptor stage walks); ties by descriptor[0] when descriptors are given"""
    oc = kps[:, 5].astype(np.int64)
    keys = (kps[:, 3], kps[:, 0], kps[:, 1], (oc >> 8) & 255, ((oc & 255) + 1) & 255)
    if des is not None:
        keys = (des[:, 0],) + keys
    return np.lexsort(keys)


def unpack_octave(packed):
    packed = int(packed)
    octave = packed & 255
    layer = (packed >> 8) & 255
    if octave >= 128:
        octave |= -128
    scale = 1.0 / (1 << octave) if octave >= 0 else float(1 << -octave)
    return octave, layer, scale


def descriptor(img, ptx, pty, ori, scl):
    """calcSIFTDescriptor (sift.simd.hpp), float32 throughout; ptx, pty, ori, scl are float32
    values in the coordinates of `img` (the keypoint's Gaussian level)."""
    d, n = DESCR_WIDTH, DESCR_HIST_BINS
    h, w = img.shape
    ptx, pty, ori, scl = F(ptx), F(pty), F(ori), F(scl)
    px, py = cv_round(ptx), cv_round(pty)
    ang = F(ori * F(math.pi / 180.0))
    cos_t = F(math.cos(float(ang)))                     # cosf / sinf: correctly rounded float32
    sin_t = F(math.sin(float(ang)))
    bins_per_rad = F(n / F(360.0))
    exp_scale = F(F(-1.0) / F(d * d * 0.5))
    hist_width = F(F(DESCR_SCL_FCTR) * scl)
    radius = cv_round(F(F(F(hist_width * F(1.4142135623730951)) * F(d + 1)) * F(0.5)))
    radius = min(radius, int(math.sqrt(float(w) * w + float(h) * h)))
    cos_t = F(cos_t / hist_width)
    sin_t = F(sin_t / hist_width)
    nb = (d + 2) * (d + 2) * (n + 2)
    bins, terms = [], []
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            c_rot = F(F(F(j) * cos_t) - F(F(i) * sin_t))
            r_rot = F(F(F(j) * sin_t) + F(F(i) * cos_t))
            rbin = F(F(r_rot + F(d // 2)) - F(0.5))
            cbin = F(F(c_rot + F(d // 2)) - F(0.5))
            r, c = py + i, px + j
            if not (rbin > -1 and rbin < d and cbin > -1 and cbin < d and 0 < r < h - 1 and 0 < c < w - 1):
                continue
            dx = F(img[r, c + 1] - img[r, c - 1])
            dy = F(img[r - 1, c] - img[r + 1, c])
            wgt = exp32(F(F(F(c_rot * c_rot) + F(r_rot * r_rot)) * exp_scale))
            og = fast_atan2(dy, dx)
            mag = F(F(np.sqrt(F(F(dx * dx) + F(dy * dy)))) * wgt)
            obin = F(F(og - ori) * bins_per_rad)
            r0, c0, o0 = int(math.floor(rbin)), int(math.floor(cbin)), int(math.floor(obin))
            fr, fc, fo = F(rbin - F(r0)), F(cbin - F(c0)), F(obin - F(o0))
            if o0 < 0:
                o0 += n
            if o0 >= n:
                o0 -= n
            v_r1 = F(mag * fr)
            v_r0 = F(mag - v_r1)
            v_rc11 = F(v_r1 * fc)
            v_rc10 = F(v_r1 - v_rc11)
            v_rc01 = F(v_r0 * fc)
            v_rc00 = F(v_r0 - v_rc01)
            for (dr_, dc_, vv) in ((0, 0, v_rc00), (0, 1, v_rc01), (1, 0, v_rc10), (1, 1, v_rc11)):
                v1 = F(vv * fo)
                base = ((r0 + 1 + dr_) * (d + 2) + (c0 + 1 + dc_)) * (n + 2) + o0
                bins += [base, base + 1]
                terms += [F(vv - v1), v1]
    hist = _bin_sums(nb, bins, terms).reshape(d + 2, d + 2, n + 2)
    hist[:, :, 0] = hist[:, :, 0] + hist[:, :, n]
    hist[:, :, 1] = hist[:, :, 1] + hist[:, :, n + 1]
    raw = hist[1:d + 1, 1:d + 1, :n].reshape(-1).copy()
    nrm2 = F(0)
    for v in raw:                                       # (the scalar loop: sequential float32)
        nrm2 = F(nrm2 + F(v * v))
    thr = F(F(np.sqrt(nrm2)) * F(DESCR_MAG_THR))
    nrm2 = F(0)
    for k in range(len(raw)):
        raw[k] = min(raw[k], thr)
        nrm2 = F(nrm2 + F(raw[k] * raw[k]))
    nrm = F(F(INT_DESCR_FCTR) / max(F(np.sqrt(nrm2)), F(FLT_EPSILON)))
    return np.clip(np.rint((raw * nrm).astype(np.float32)), 0, 255).astype(np.uint8)


def descriptor_params(kps):
    """calcDescriptors: per keypoint the Gaussian level (octave index, layer) and the float32
    arguments of calcSIFTDescriptor (ptf.x, ptf.y, angle, size * 0.5) in that level's pixels"""
    par = np.zeros((len(kps), 4), np.float32)
    level = np.zeros((len(kps), 2), np.int64)
    for k, (x, y, size, angle, _resp, packed) in enumerate(kps):
        octave, layer, scale = unpack_octave(packed)
        scale = F(scale)
        a = F(F(360.0) - F(angle))
        if abs(F(a - F(360.0))) < F(FLT_EPSILON):
            a = F(0)
        par[k] = (F(F(x) * scale), F(F(y) * scale), a, F(F(F(size) * scale) * F(0.5)))
        level[k] = (octave + 1, layer)
    return par, level


def detect_and_compute(bgr_or_gray, use_c=True, order='opencv', return_removed=False):
    """cv2.SIFT_create().detectAndCompute(img, None) -> keypoints [N,6] (x, y, size, angle,
    response, packed octave; float32 values) and descriptors [N,128] u8, duplicates removed
    (KeyPointsFilter::removeDuplicatedSorted) and in OpenCV's output order (order='opencv') or in
    the pyramid-local (octave, layer, y, x, angle) order (order='canonical').
    use_c: the per-keypoint loops through oracle/sift_ref.c (OpenMP), else their numpy twins
    (small images only)."""
    kps, gauss = detect(bgr_or_gray, use_c=use_c)
    kps, _idx, removed = remove_duplicated_sorted(kps)
    des = np.zeros((len(kps), 128), np.uint8)
    par, level = descriptor_params(kps)
    if use_c and len(kps):
        from . import cpu_ref
        for o, layer in sorted(set(map(tuple, level.tolist()))):
            sel = np.nonzero((level[:, 0] == o) & (level[:, 1] == layer))[0]
            des[sel] = cpu_ref.sift_descriptors(gauss[o][layer], par[sel])
    else:
        for k in range(len(kps)):
            des[k] = descriptor(gauss[level[k, 0]][level[k, 1]], *par[k])
    if order == 'canonical' and len(kps):
        sel = canonical_order(kps, des)
        kps, des = kps[sel], des[sel]
    return (kps, des, removed) if return_removed else (kps, des)


def detect_and_compute_c(bgr_or_gray, order='opencv'):
    """detect_and_compute() with nothing but oracle/sift_ref.c underneath (OpenMP): the form
    bench.py times as the CPU baseline of the SIFT section"""
    from . import cpu_ref
    gray = bgr_to_gray(bgr_or_gray) if bgr_or_gray.ndim == 3 else bgr_or_gray
    kps, des = cpu_ref.sift_detect(gray)
    if order == 'canonical' and len(kps):
        sel = canonical_order(kps, des)
        kps, des = kps[sel], des[sel]
    return kps, des
