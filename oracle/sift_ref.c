/* ORACLE / TEST INFRASTRUCTURE ONLY -- plain-C restatement of the hot loops of
 * oracle/sift_oracle.py (the SIFT detector + descriptor the reference obtains from
 * cv2.SIFT_create().detectAndCompute, scripts/lib/image.py:235-237,324), so that the
 * oracle runs on whole 2189 x 1459 frames in seconds and doubles as the OpenMP CPU baseline of
 * bench.py's SIFT section ("kind": "port").  PARITY UNPINNED like sift_oracle.py: OpenCV is not
 * in /root/reference.  Every function here has a numpy twin in sift_oracle.py (the *_py
 * functions); tests/test_oracle.py checks that the two agree on small images.
 * Never linked or loaded by the product (imageanalysis_amd/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NL 3
#define IMG_BORDER 5
#define MAX_INTERP_STEPS 5
#define ORI_BINS 36
#define DESCR_W 4
#define DESCR_N 8
#define FLT_EPS 1.1920929e-07

static int reflect101(int p, int n)
{
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    p = p < 0 ? -p : p;
    p %= period;
    return p >= n ? period - p : p;
}

/* sift_oracle.gaussian_blur: separable, BORDER_REFLECT_101, taps added in ascending order with
 * one rounding per tap -- acc = fmaf(v, k[t], acc) from 0 -- horizontally, then vertically.
 * fmaf is the correctly rounded fused multiply-add whether the compiler emits the hardware
 * instruction (target clone "fma") or calls libm. */
__attribute__((target_clones("fma", "default")))
static void blur_rows(const float *src, int h, int w, const float *k, int r, float *tmp, float *dst)
{
#pragma omp parallel
    {
        int *xi = (int *)malloc(sizeof(int) * (size_t)(w + 2 * r));
        for (int i = 0; i < w + 2 * r; ++i) xi[i] = reflect101(i - r, w);
#pragma omp for schedule(static)
        for (int y = 0; y < h; ++y) {
            const float *s = src + (size_t)y * w;
            float *d = tmp + (size_t)y * w;
            for (int x = 0; x < w; ++x) {
                float acc = 0.f;
                if (x >= r && x + r < w) {
                    for (int t = 0; t <= 2 * r; ++t) acc = fmaf(s[x - r + t], k[t], acc);
                } else {
                    for (int t = 0; t <= 2 * r; ++t) acc = fmaf(s[xi[x + t]], k[t], acc);
                }
                d[x] = acc;
            }
        }
#pragma omp for schedule(static)
        for (int y = 0; y < h; ++y) {
            float *d = dst + (size_t)y * w;
            for (int x = 0; x < w; ++x) d[x] = 0.f;
            for (int t = 0; t <= 2 * r; ++t) {
                const float *s = tmp + (size_t)reflect101(y - r + t, h) * w;
                const float kt = k[t];
                for (int x = 0; x < w; ++x) d[x] = fmaf(s[x], kt, d[x]);
            }
        }
        free(xi);
    }
}

int oracle_sift_blur(const float *src, int h, int w, const float *k, int r, float *dst, int nthreads)
{
    if (!src || !k || !dst || h < 1 || w < 1 || r < 0) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    float *tmp = (float *)malloc(sizeof(float) * (size_t)h * w);
    if (!tmp) return -2;
    blur_rows(src, h, w, k, r, tmp, dst);
    free(tmp);
    return 0;
}

/* ---- float32 scalar conventions of OpenCV's sift.simd.hpp / mathfuncs, twins of the functions of
 * the same names in sift_oracle.py.  The file is compiled with -ffp-contract=off: every float
 * operator below is one IEEE float32 operation. */
static float fast_atan2f_cv(float y, float x)
{
    /* cv::fastAtan2, scalar form (mathfuncs_core): degrees in [0, 360] */
    static const float P1 = 0.9997878412794807f * 57.29577951308232f, P3 = -0.3258083974640975f * 57.29577951308232f,
                       P5 = 0.1555786518463281f * 57.29577951308232f, P7 = -0.04432655554792128f * 57.29577951308232f;
    const float eps = 2.220446049250313e-16f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((P7 * c2 + P5) * c2 + P3) * c2 + P1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((P7 * c2 + P5) * c2 + P3) * c2 + P1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static float exp32(float x)
{
    /* sift_oracle.exp32: float64 range reduction, degree-7 float32 Horner polynomial for 2^f */
    static const float C[8] = {1.0f, 0.6931471805599453f, 0.2402265069591007f, 0.05550410866482158f,
                               0.009618129107628477f, 0.0013333558146428443f, 0.00015403530393381608f,
                               1.5252733804059841e-05f};
    if (x < -87.f) return 0.f;
    const double t = (double)x * 1.4426950408889634;
    const double n = rint(t);
    const float f = (float)(t - n);
    float p = C[7];
    for (int k = 6; k >= 0; --k) p = p * f + C[k];
    return ldexpf(p, (int)n);
}

/* Matx33f::solve(Vec3f, DECOMP_LU) = Cramer's rule in float32 (Matx_FastSolveOp<float, 3, 3, 1>);
 * determinant exactly 0 -> the zero vector */
static void solve3_cramer(const float a[3][3], const float b[3], float x[3])
{
    const float det = a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2]) -
                      a[0][1] * (a[1][0] * a[2][2] - a[2][0] * a[1][2]) +
                      a[0][2] * (a[1][0] * a[2][1] - a[2][0] * a[1][1]);
    if (det == 0) { x[0] = x[1] = x[2] = 0.f; return; }
    const float d = 1.f / det;
    x[0] = d * (b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) -
                a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2]) +
                a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
    x[1] = d * (a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) -
                b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
    x[2] = d * (a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) -
                a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0]) +
                b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
}

typedef struct {
    int layer, r, c;
    float xi, xr, xc, contr;
} refined_t;

/* sift_oracle._adjust_local_extrema = adjustLocalExtrema of sift.simd.hpp (float32 throughout) */
static int adjust_local_extrema(const float *const *dogs, int h, int w, int layer, int r, int c,
                                refined_t *out)
{
    const float img_scale = 1.f / 255.f;
    const float deriv_scale = img_scale * 0.5f, second_scale = img_scale, cross_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int it = 0;
#define AT(im, rr, cc) ((im)[(size_t)(rr) * w + (cc)])
    for (; it < MAX_INTERP_STEPS; ++it) {
        const float *img = dogs[layer], *prv = dogs[layer - 1], *nxt = dogs[layer + 1];
        const float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale,
                             (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                             (AT(nxt, r, c) - AT(prv, r, c)) * deriv_scale};
        const float v2 = AT(img, r, c) * 2.f;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_scale;
        const float dss = (AT(nxt, r, c) + AT(prv, r, c) - v2) * second_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_scale;
        const float dxs = (AT(nxt, r, c + 1) - AT(nxt, r, c - 1) - AT(prv, r, c + 1) + AT(prv, r, c - 1)) * cross_scale;
        const float dys = (AT(nxt, r + 1, c) - AT(nxt, r - 1, c) - AT(prv, r + 1, c) + AT(prv, r - 1, c)) * cross_scale;
        const float H[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        float X[3];
        solve3_cramer(H, dD, X);
        xc = -X[0]; xr = -X[1]; xi = -X[2];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        const float big = (float)(2147483647 / 3);
        if (fabsf(xi) > big || fabsf(xr) > big || fabsf(xc) > big) return 0;
        c += (int)lrintf(xc);
        r += (int)lrintf(xr);
        layer += (int)lrintf(xi);
        if (layer < 1 || layer > NL || c < IMG_BORDER || c >= w - IMG_BORDER || r < IMG_BORDER ||
            r >= h - IMG_BORDER)
            return 0;
    }
    if (it >= MAX_INTERP_STEPS) return 0;
    {
        const float *img = dogs[layer], *prv = dogs[layer - 1], *nxt = dogs[layer + 1];
        const float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale,
                             (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                             (AT(nxt, r, c) - AT(prv, r, c)) * deriv_scale};
        float t = 0.f;                                   /* Matx::dot */
        t += dD[0] * xc; t += dD[1] * xr; t += dD[2] * xi;
        const float contr = AT(img, r, c) * img_scale + t * 0.5f;
        if (fabsf(contr) * (float)NL < 0.04f) return 0;
        const float v2 = AT(img, r, c) * 2.f;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        const float e = 10.f;
        if (det <= 0 || tr * tr * e >= (e + 1) * (e + 1) * det) return 0;
        out->layer = layer; out->r = r; out->c = c;
        out->xi = xi; out->xr = xr; out->xc = xc; out->contr = contr;
    }
#undef AT
    return 1;
}

/* sift_oracle._orientation_hist + _orientation_peaks (calcOrientationHist and the peak search of
 * findScaleSpaceExtrema): float32 terms W * Mag, every bin the exact sum of its terms rounded to
 * float32 once (float64 accumulation), float32 smoothing / peak interpolation */
static int orientation_peaks(const float *img, int h, int w, int r, int c, int radius, float sigma,
                             float *angles /* [ORI_BINS] */)
{
    const int n = ORI_BINS;
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    double acc[ORI_BINS];
    float t[ORI_BINS], sm[ORI_BINS];
    for (int k = 0; k < n; ++k) acc[k] = 0.0;
    for (int i = -radius; i <= radius; ++i) {
        const int y = r + i;
        if (y <= 0 || y >= h - 1) continue;
        for (int j = -radius; j <= radius; ++j) {
            const int x = c + j;
            if (x <= 0 || x >= w - 1) continue;
            const float dx = img[(size_t)y * w + x + 1] - img[(size_t)y * w + x - 1];
            const float dy = img[(size_t)(y - 1) * w + x] - img[(size_t)(y + 1) * w + x];
            const float wgt = exp32((float)(i * i + j * j) * expf_scale);
            const float ori = fast_atan2f_cv(dy, dx);
            const float mag = sqrtf(dx * dx + dy * dy);
            int b = (int)lrintf((float)(n / 360.f) * ori);
            if (b >= n) b -= n;
            if (b < 0) b += n;
            acc[b] += (double)(wgt * mag);
        }
    }
    for (int k = 0; k < n; ++k) t[k] = (float)acc[k];
    float omax = 0.f;
    for (int k = 0; k < n; ++k) {
        const float m2 = t[(k + n - 2) % n], p2 = t[(k + 2) % n];
        const float m1 = t[(k + n - 1) % n], p1 = t[(k + 1) % n];
        sm[k] = (m2 + p2) * (1.f / 16.f) + (m1 + p1) * (4.f / 16.f) + t[k] * (6.f / 16.f);
        if (k == 0 || sm[k] > omax) omax = sm[k];
    }
    const float mag_thr = omax * 0.8f;
    int np = 0;
    for (int j = 0; j < n; ++j) {
        const float lft = sm[(j + n - 1) % n], rgt = sm[(j + 1) % n];
        if (sm[j] > lft && sm[j] > rgt && sm[j] >= mag_thr) {
            float bin = (float)j + 0.5f * (lft - rgt) / (lft - 2 * sm[j] + rgt);
            bin = bin < 0 ? (float)n + bin : (bin >= n ? bin - (float)n : bin);
            float angle = 360.f - (360.f / n) * bin;
            if (fabsf(angle - 360.f) < (float)FLT_EPS) angle = 0.f;
            angles[np++] = angle;
        }
    }
    return np;
}

/* detect() of sift_oracle.py for the candidates of one octave, in the order given (layer,
 * then row major): refinement, contrast / edge tests, orientation peaks.  dogs: NL+2 levels,
 * gauss: NL+3 levels of the octave, cand [n][3] = (layer, r, c).  kps [cap][6] = x, y, size,
 * angle, |contrast| (float32 values), packed octave in the coordinates of the DOUBLED image
 * (detect() halves them afterwards).  Returns the number of keypoints (may exceed cap: only cap
 * are stored). */
int oracle_sift_keypoints(const float *const *dogs, const float *const *gauss, int h, int w, int o,
                          const int32_t *cand, int n, double sigma0, double *kps, int cap, int nthreads)
{
    if (!dogs || !gauss || !cand || !kps) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    /* per candidate: up to ORI_BINS keypoints, written to a private slot, compacted in order */
    int *cnt = (int *)calloc((size_t)n + 1, sizeof(int));
    double *loc = (double *)malloc(sizeof(double) * (size_t)n * 6 * 4);     /* 4 peaks inline */
    double **more = (double **)calloc((size_t)n, sizeof(double *));
    if (!cnt || !loc || !more) return -2;
    const float sigma = (float)sigma0;                   /* adjustLocalExtrema takes float sigma */
#pragma omp parallel for schedule(dynamic, 64)
    for (int k = 0; k < n; ++k) {
        refined_t R;
        if (!adjust_local_extrema(dogs, h, w, cand[3 * k], cand[3 * k + 1], cand[3 * k + 2], &R)) continue;
        const float scale = (float)(1 << o);
        const float e = ((float)R.layer + R.xi) / (float)NL;
        const float size = sigma * (float)exp2((double)e) * scale * 2.f;     /* powf(2.f, e) */
        const float px = ((float)R.c + R.xc) * scale, py = ((float)R.r + R.xr) * scale;
        const int octave = o + (R.layer << 8) + ((int)lrint(((double)R.xi + 0.5) * 255) << 16);
        const float scl_octv = size * 0.5f / scale;
        float angles[ORI_BINS];
        const int np = orientation_peaks(gauss[R.layer], h, w, R.r, R.c, (int)lrintf(4.5f * scl_octv),
                                         1.5f * scl_octv, angles);
        double *dstp = loc + (size_t)k * 24;
        if (np > 4) {
            more[k] = (double *)malloc(sizeof(double) * 6 * (size_t)np);
            dstp = more[k];
        }
        for (int j = 0; j < np; ++j) {
            double *q = dstp + 6 * j;
            q[0] = px; q[1] = py; q[2] = size; q[3] = angles[j]; q[4] = fabsf(R.contr); q[5] = (double)octave;
        }
        cnt[k] = np;
    }
    int total = 0;
    for (int k = 0; k < n; ++k) {
        const double *srcp = more[k] ? more[k] : loc + (size_t)k * 24;
        for (int j = 0; j < cnt[k]; ++j, ++total)
            if (total < cap) memcpy(kps + (size_t)total * 6, srcp + 6 * j, 6 * sizeof(double));
        free(more[k]);
    }
    free(cnt); free(loc); free(more);
    return total;
}

/* sift_oracle.descriptor = calcSIFTDescriptor (float32 throughout; bins = exact sums rounded once) */
static void descriptor_one(const float *img, int h, int w, float ptx, float pty, float ori, float scl,
                           uint8_t *out)
{
    const int d = DESCR_W, n = DESCR_N;
    const int px = (int)lrintf(ptx), py = (int)lrintf(pty);
    const float ang = ori * (float)(3.141592653589793 / 180.0);
    float cos_t = (float)cos((double)ang), sin_t = (float)sin((double)ang);   /* cosf / sinf, correctly rounded */
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f);
    const float hist_width = 3.f * scl;
    int radius = (int)lrintf(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    const int diag = (int)sqrt((double)w * w + (double)h * h);
    radius = radius < diag ? radius : diag;
    cos_t /= hist_width;
    sin_t /= hist_width;
    double acc[(DESCR_W + 2) * (DESCR_W + 2) * (DESCR_N + 2)];
    float hist[(DESCR_W + 2) * (DESCR_W + 2) * (DESCR_N + 2)];
    memset(acc, 0, sizeof(acc));
    for (int i = -radius; i <= radius; ++i) {
        for (int j = -radius; j <= radius; ++j) {
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            const float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < h - 1 && c > 0 && c < w - 1))
                continue;
            const float dx = img[(size_t)r * w + c + 1] - img[(size_t)r * w + c - 1];
            const float dy = img[(size_t)(r - 1) * w + c] - img[(size_t)(r + 1) * w + c];
            const float wgt = exp32((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            const float og = fast_atan2f_cv(dy, dx);
            const float mag = sqrtf(dx * dx + dy * dy) * wgt;
            const float obin = (og - ori) * bins_per_rad;
            const float fr0 = floorf(rbin), fc0 = floorf(cbin), fo0 = floorf(obin);
            const int r0 = (int)fr0, c0 = (int)fc0;
            int o0 = (int)fo0;
            const float fr = rbin - fr0, fc = cbin - fc0, fo = obin - fo0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * fr, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * fc, v_rc10 = v_r1 - v_rc11;
            const float v_rc01 = v_r0 * fc, v_rc00 = v_r0 - v_rc01;
            const float vv[4] = {v_rc00, v_rc01, v_rc10, v_rc11};
            for (int q4 = 0; q4 < 4; ++q4) {
                const int rr = r0 + 1 + (q4 >> 1), cc = c0 + 1 + (q4 & 1);
                const int base = (rr * (d + 2) + cc) * (n + 2) + o0;
                const float v1 = vv[q4] * fo;
                acc[base] += (double)(vv[q4] - v1);
                acc[base + 1] += (double)v1;
            }
        }
    }
    for (int k = 0; k < (d + 2) * (d + 2) * (n + 2); ++k) hist[k] = (float)acc[k];
    float dst[DESCR_W * DESCR_W * DESCR_N];
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            float *cell = hist + ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            cell[0] += cell[n];
            cell[1] += cell[n + 1];
            for (int k = 0; k < n; ++k) dst[(i * d + j) * n + k] = cell[k];
        }
    float nrm2 = 0.f;
    for (int k = 0; k < d * d * n; ++k) nrm2 += dst[k] * dst[k];
    const float thr = sqrtf(nrm2) * 0.2f;
    nrm2 = 0.f;
    for (int k = 0; k < d * d * n; ++k) {
        const float val = dst[k] < thr ? dst[k] : thr;
        dst[k] = val;
        nrm2 += val * val;
    }
    const float s2 = sqrtf(nrm2);
    const float nrm = 512.f / (s2 > (float)FLT_EPS ? s2 : (float)FLT_EPS);
    for (int k = 0; k < d * d * n; ++k) {
        float x = rintf(dst[k] * nrm);
        x = x < 0 ? 0 : (x > 255 ? 255 : x);
        out[k] = (uint8_t)x;
    }
}

/* descriptors of n keypoints that live on one Gaussian level: par [n][4] = ptx, pty, ori, scl
 * (float32) in the coordinates of that level; desc [n][128] */
int oracle_sift_descriptors(const float *img, int h, int w, const float *par, int n, uint8_t *desc,
                            int nthreads)
{
    if (!img || !par || !desc) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 16)
    for (int k = 0; k < n; ++k)
        descriptor_one(img, h, w, par[4 * k], par[4 * k + 1], par[4 * k + 2], par[4 * k + 3],
                       desc + (size_t)k * 128);
    return 0;
}

/* KeyPoint12_LessThan of KeyPointsFilter::removeDuplicatedSorted on rows x, y, size, angle,
 * response, packed octave (before the first-octave adjustment; class_id is -1 everywhere) */
static int kp12_less(const void *pa, const void *pb)
{
    const double *a = (const double *)pa, *b = (const double *)pb;
    if (a[0] != b[0]) return a[0] < b[0] ? -1 : 1;
    if (a[1] != b[1]) return a[1] < b[1] ? -1 : 1;
    if (a[2] != b[2]) return a[2] > b[2] ? -1 : 1;
    if (a[3] != b[3]) return a[3] < b[3] ? -1 : 1;
    if (a[4] != b[4]) return a[4] > b[4] ? -1 : 1;
    if (a[5] != b[5]) return a[5] > b[5] ? -1 : 1;
    return 0;
}

/* sift_oracle.remove_duplicated_sorted: sorts kps [n][6] in place, returns the number kept */
int oracle_sift_remove_duplicated_sorted(double *kps, int n)
{
    if (!kps || n < 0) return -1;
    if (n < 2) return n;
    qsort(kps, (size_t)n, 6 * sizeof(double), kp12_less);
    int i = 0;
    for (int j = 1; j < n; ++j) {
        const double *k1 = kps + (size_t)i * 6, *k2 = kps + (size_t)j * 6;
        if (k1[0] != k2[0] || k1[1] != k2[1] || k1[2] != k2[2] || k1[3] != k2[3]) {
            ++i;
            if (i != j) memcpy(kps + (size_t)i * 6, k2, 6 * sizeof(double));
        }
    }
    return i + 1;
}

/* ---------------------------------------------------------------------------------------------
 * The whole detector + descriptor in C (sift_oracle.detect_and_compute without its numpy glue):
 * grey image -> x2 bilinear -> Gaussian / DoG pyramid -> 26-neighbour extrema -> refinement,
 * orientation peaks -> descriptors.  OpenMP over rows / candidates / keypoints.  This is the CPU
 * baseline of bench.py's SIFT section; tests/test_oracle.py checks it against the numpy oracle.
 * Output: duplicates removed and in OpenCV's order (KeyPointsFilter::removeDuplicatedSorted).
 * ------------------------------------------------------------------------------------------- */
static void gaussian_taps(double sigma, float *k, int *r_out)
{
    int ksize = (int)lrint(sigma * 8 + 1) | 1;
    int r = ksize / 2;
    double kd[129], sum = 0;
    if (r > 64) r = 64;
    for (int i = -r; i <= r; ++i) {
        kd[i + r] = exp(-(double)(i * i) / (2.0 * sigma * sigma));
        sum += kd[i + r];
    }
    for (int i = 0; i < 2 * r + 1; ++i) k[i] = (float)(kd[i] / sum);
    *r_out = r;
}

/* sift_oracle.resize_linear_2x (separately rounded float32 operations) */
static void resize2x(const uint8_t *gray, int h, int w, float *dst)
{
    const int H2 = 2 * h, W2 = 2 * w;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H2; ++y) {
        volatile float f = ((float)y + 0.5f) * 0.5f;
        float fy = f - 0.5f;
        int y0 = (int)floorf(fy);
        float ty = fy - (float)y0;
        if (y0 < 0) { y0 = 0; ty = 0.f; }
        if (y0 >= h - 1) { y0 = h - 1; ty = 0.f; }
        const int y1 = y0 + 1 < h ? y0 + 1 : h - 1;
        for (int x = 0; x < W2; ++x) {
            volatile float g = ((float)x + 0.5f) * 0.5f;
            float fx = g - 0.5f;
            int x0 = (int)floorf(fx);
            float tx = fx - (float)x0;
            if (x0 < 0) { x0 = 0; tx = 0.f; }
            if (x0 >= w - 1) { x0 = w - 1; tx = 0.f; }
            const int x1 = x0 + 1 < w ? x0 + 1 : w - 1;
            const float omtx = 1.f - tx, omty = 1.f - ty;
            volatile float a = (float)gray[(size_t)y0 * w + x0] * omtx, b = (float)gray[(size_t)y0 * w + x1] * tx;
            const float top = a + b;
            a = (float)gray[(size_t)y1 * w + x0] * omtx; b = (float)gray[(size_t)y1 * w + x1] * tx;
            const float bot = a + b;
            a = top * omty; b = bot * ty;
            dst[(size_t)y * W2 + x] = a + b;
        }
    }
}

int oracle_sift_detect(const uint8_t *gray, int h, int w, double *kps, uint8_t *desc, int cap,
                       int nthreads)
{
    if (!gray || !kps || !desc || h < 2 || w < 2) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    const double sigma0 = 1.6;
    int H = 2 * h, W = 2 * w;
    int n_oct = (int)lrint(log((double)(H < W ? H : W)) / log(2.0) - 2) + 1;
    if (n_oct < 1) n_oct = 1;
    if (n_oct > 24) n_oct = 24;
    double sig[NL + 3];
    sig[0] = sigma0;
    const double kf = pow(2.0, 1.0 / NL);
    for (int i = 1; i < NL + 3; ++i) {
        const double sp = pow(kf, (double)(i - 1)) * sigma0, st = sp * kf;
        sig[i] = sqrt(st * st - sp * sp);
    }
    float **gauss = (float **)calloc((size_t)n_oct * (NL + 3), sizeof(float *));
    float **dogs = (float **)calloc((size_t)n_oct * (NL + 2), sizeof(float *));
    int *oh = (int *)calloc((size_t)n_oct, sizeof(int)), *ow = (int *)calloc((size_t)n_oct, sizeof(int));
    float *tmp = (float *)malloc(sizeof(float) * (size_t)H * W);
    float taps[129];
    int r;
    int total = 0;
    for (int o = 0; o < n_oct; ++o) {
        oh[o] = H; ow[o] = W;
        const size_t npx = (size_t)H * W;
        for (int i = 0; i < NL + 3; ++i) gauss[o * (NL + 3) + i] = (float *)malloc(sizeof(float) * npx);
        for (int i = 0; i < NL + 2; ++i) dogs[o * (NL + 2) + i] = (float *)malloc(sizeof(float) * npx);
        float **g = gauss + o * (NL + 3), **dg = dogs + o * (NL + 2);
        if (o == 0) {
            float *up = (float *)malloc(sizeof(float) * npx);
            resize2x(gray, h, w, up);
            gaussian_taps(sqrt(fmax(sigma0 * sigma0 - 1.0, 0.01)), taps, &r);
            blur_rows(up, H, W, taps, r, tmp, g[0]);
            free(up);
        } else {
            const float *src = gauss[(o - 1) * (NL + 3) + NL];
            const int sw = ow[o - 1];
#pragma omp parallel for schedule(static)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) g[0][(size_t)y * W + x] = src[(size_t)(2 * y) * sw + 2 * x];
        }
        for (int i = 1; i < NL + 3; ++i) {
            gaussian_taps(sig[i], taps, &r);
            blur_rows(g[i - 1], H, W, taps, r, tmp, g[i]);
#pragma omp parallel for schedule(static)
            for (size_t q = 0; q < npx; ++q) dg[i - 1][q] = g[i][q] - g[i - 1][q];
        }
        /* extrema: candidates per (layer, row) counted, then written in order */
        if (H > 2 * IMG_BORDER && W > 2 * IMG_BORDER) {
            const float threshold = floorf(0.5f * 0.04f / NL * 255.f);
            const int rows = H - 2 * IMG_BORDER;
            int *cnt = (int *)calloc((size_t)NL * rows + 1, sizeof(int));
            for (int pass = 0; pass < 2; ++pass) {
                int32_t *cand = NULL;
                if (pass == 1) {
                    int acc = 0;
                    for (int q = 0; q < NL * rows; ++q) { const int c = cnt[q]; cnt[q] = acc; acc += c; }
                    cnt[NL * rows] = acc;
                    cand = (int32_t *)malloc(sizeof(int32_t) * 3 * (size_t)(acc + 1));
                }
#pragma omp parallel for schedule(dynamic, 8) collapse(2)
                for (int layer = 1; layer <= NL; ++layer)
                    for (int rr = IMG_BORDER; rr < H - IMG_BORDER; ++rr) {
                        const int slot = (layer - 1) * rows + (rr - IMG_BORDER);
                        int n = 0;
                        for (int cc = IMG_BORDER; cc < W - IMG_BORDER; ++cc) {
                            const float val = dg[layer][(size_t)rr * W + cc];
                            if (!(fabsf(val) > threshold)) continue;
                            int is_max = val > 0, is_min = val < 0;
                            for (int dl = -1; dl <= 1 && (is_max || is_min); ++dl)
                                for (int dr = -1; dr <= 1; ++dr)
                                    for (int dc = -1; dc <= 1; ++dc) {
                                        const float nb = dg[layer + dl][(size_t)(rr + dr) * W + cc + dc];
                                        is_max &= val >= nb;
                                        is_min &= val <= nb;
                                    }
                            if (is_max || is_min) {
                                if (pass == 1) {
                                    int32_t *q = cand + 3 * (size_t)(cnt[slot] + n);
                                    q[0] = layer; q[1] = rr; q[2] = cc;
                                }
                                ++n;
                            }
                        }
                        if (pass == 0) cnt[slot] = n;
                    }
                if (pass == 1) {
                    const int n_c = cnt[NL * rows];
                    const int room = cap - total > 0 ? cap - total : 0;
                    const int got = oracle_sift_keypoints((const float *const *)dg, (const float *const *)g, H, W,
                                                          o, cand, n_c, sigma0, kps + (size_t)total * 6, room, 0);
                    if (got > 0) total += got < room ? got : room;
                    free(cand);
                }
            }
            free(cnt);
        }
        H /= 2; W /= 2;
        if (H < 1 || W < 1) { n_oct = o + 1; break; }
    }
    /* detectAndCompute: removeDuplicatedSorted (also fixes the output order), then first octave
     * is -1 -> input-image coordinates, then the descriptors in that order */
    total = oracle_sift_remove_duplicated_sorted(kps, total);
#pragma omp parallel for schedule(dynamic, 16)
    for (int k = 0; k < total; ++k) {
        double *q = kps + (size_t)k * 6;
        int oc = (int)q[5];
        oc = (oc & ~255) | ((oc - 1) & 255);
        q[5] = (double)oc;
        q[0] = (double)((float)q[0] * 0.5f); q[1] = (double)((float)q[1] * 0.5f);
        q[2] = (double)((float)q[2] * 0.5f);
        int octave = oc & 255, layer = (oc >> 8) & 255;
        if (octave >= 128) octave |= -128;
        const float scale = octave >= 0 ? 1.f / (float)(1 << octave) : (float)(1 << -octave);
        float a = 360.f - (float)q[3];
        if (fabsf(a - 360.f) < (float)FLT_EPS) a = 0.f;
        const int oi = octave + 1;
        descriptor_one(gauss[oi * (NL + 3) + layer], oh[oi], ow[oi], (float)q[0] * scale, (float)q[1] * scale, a,
                       (float)q[2] * scale * 0.5f, desc + (size_t)k * 128);
    }
    for (int q = 0; q < n_oct * (NL + 3); ++q) free(gauss[q]);
    for (int q = 0; q < n_oct * (NL + 2); ++q) free(dogs[q]);
    free(gauss); free(dogs); free(oh); free(ow); free(tmp);
    return total;
}
