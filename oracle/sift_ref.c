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

/* Gaussian elimination with partial pivoting on a 3 x 3 system (what H.solve(dD, DECOMP_LU)
 * does; the same sequence as sift_oracle._solve3 and the device's solve3) */
static int solve3(double A[3][3], double b[3], double x[3])
{
    int p[3] = {0, 1, 2};
    for (int k = 0; k < 3; ++k) {
        int piv = k;
        double best = fabs(A[p[k]][k]);
        for (int i = k + 1; i < 3; ++i)
            if (fabs(A[p[i]][k]) > best) { best = fabs(A[p[i]][k]); piv = i; }
        if (best < 1e-300) return 0;
        const int t = p[k]; p[k] = p[piv]; p[piv] = t;
        for (int i = k + 1; i < 3; ++i) {
            const double f = A[p[i]][k] / A[p[k]][k];
            for (int j = k; j < 3; ++j) A[p[i]][j] -= f * A[p[k]][j];
            b[p[i]] -= f * b[p[k]];
        }
    }
    for (int k = 2; k >= 0; --k) {
        double s = b[p[k]];
        for (int j = k + 1; j < 3; ++j) s -= A[p[k]][j] * x[j];
        x[k] = s / A[p[k]][k];
    }
    return 1;
}

typedef struct {
    int layer, r, c;
    double xi, xr, xc, contr;
} refined_t;

/* sift_oracle._adjust_local_extrema (float32 derivative arithmetic with separately rounded
 * operations, float64 solve / contrast / edge tests) */
/* (compiled with -ffp-contract=off: oracle/Makefile) */
static int adjust_local_extrema(const float *const *dogs, int h, int w, int layer, int r, int c,
                                refined_t *out)
{
    const float img_scale = 1.f / 255.f;
    const float deriv_scale = img_scale * 0.5f, second_scale = img_scale, cross_scale = img_scale * 0.25f;
    double xi = 0, xr = 0, xc = 0;
    int it = 0;
#define AT(im, rr, cc) ((im)[(size_t)(rr) * w + (cc)])
    for (; it < MAX_INTERP_STEPS; ++it) {
        const float *img = dogs[layer], *prv = dogs[layer - 1], *nxt = dogs[layer + 1];
        volatile float t0;
        t0 = AT(img, r, c + 1) - AT(img, r, c - 1); const float dDx = t0 * deriv_scale;
        t0 = AT(img, r + 1, c) - AT(img, r - 1, c); const float dDy = t0 * deriv_scale;
        t0 = AT(nxt, r, c) - AT(prv, r, c);         const float dDs = t0 * deriv_scale;
        const float v2 = AT(img, r, c) * 2.f;
        t0 = AT(img, r, c + 1) + AT(img, r, c - 1); t0 = t0 - v2; const float dxx = t0 * second_scale;
        t0 = AT(img, r + 1, c) + AT(img, r - 1, c); t0 = t0 - v2; const float dyy = t0 * second_scale;
        t0 = AT(nxt, r, c) + AT(prv, r, c);         t0 = t0 - v2; const float dss = t0 * second_scale;
        t0 = AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1); t0 = t0 - AT(img, r - 1, c + 1);
        t0 = t0 + AT(img, r - 1, c - 1); const float dxy = t0 * cross_scale;
        t0 = AT(nxt, r, c + 1) - AT(nxt, r, c - 1); t0 = t0 - AT(prv, r, c + 1);
        t0 = t0 + AT(prv, r, c - 1); const float dxs = t0 * cross_scale;
        t0 = AT(nxt, r + 1, c) - AT(nxt, r - 1, c); t0 = t0 - AT(prv, r + 1, c);
        t0 = t0 + AT(prv, r - 1, c); const float dys = t0 * cross_scale;
        double A[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        double b[3] = {dDx, dDy, dDs}, X[3];
        if (!solve3(A, b, X)) return 0;
        xc = -X[0]; xr = -X[1]; xi = -X[2];
        if (fabs(xi) < 0.5 && fabs(xr) < 0.5 && fabs(xc) < 0.5) break;
        if (fabs(xi) > 2147483647.0 / 3 || fabs(xr) > 2147483647.0 / 3 || fabs(xc) > 2147483647.0 / 3)
            return 0;
        c += (int)rint(xc);
        r += (int)rint(xr);
        layer += (int)rint(xi);
        if (layer < 1 || layer > NL || c < IMG_BORDER || c >= w - IMG_BORDER || r < IMG_BORDER ||
            r >= h - IMG_BORDER)
            return 0;
    }
    if (it >= MAX_INTERP_STEPS) return 0;
    {
        const float *img = dogs[layer], *prv = dogs[layer - 1], *nxt = dogs[layer + 1];
        volatile float t0;
        t0 = AT(img, r, c + 1) - AT(img, r, c - 1); const double dDx = (double)(float)(t0 * deriv_scale);
        t0 = AT(img, r + 1, c) - AT(img, r - 1, c); const double dDy = (double)(float)(t0 * deriv_scale);
        t0 = AT(nxt, r, c) - AT(prv, r, c);         const double dDs = (double)(float)(t0 * deriv_scale);
        const double t = dDx * xc + dDy * xr + dDs * xi;
        const double contr = (double)AT(img, r, c) * (double)img_scale + t * 0.5;
        if (fabs(contr) * NL < 0.04) return 0;
        const double v2 = (double)AT(img, r, c) * 2.0;
        const double dxx = ((double)AT(img, r, c + 1) + (double)AT(img, r, c - 1) - v2) * (double)second_scale;
        const double dyy = ((double)AT(img, r + 1, c) + (double)AT(img, r - 1, c) - v2) * (double)second_scale;
        const double dxy = ((double)AT(img, r + 1, c + 1) - (double)AT(img, r + 1, c - 1)
                            - (double)AT(img, r - 1, c + 1) + (double)AT(img, r - 1, c - 1)) * (double)cross_scale;
        const double tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        const double e = 10.0;
        if (det <= 0 || tr * tr * e >= (e + 1) * (e + 1) * det) return 0;
        out->layer = layer; out->r = r; out->c = c;
        out->xi = xi; out->xr = xr; out->xc = xc; out->contr = contr;
    }
#undef AT
    return 1;
}

/* sift_oracle._orientation_hist (float64 throughout) + the peak search of detect() */
static int orientation_peaks(const float *img, int h, int w, int r, int c, int radius, double sigma,
                             double *angles /* [ORI_BINS] */)
{
    const int n = ORI_BINS;
    const double expf_scale = -1.0 / (2.0 * sigma * sigma);
    double hist[ORI_BINS], sm[ORI_BINS];
    for (int k = 0; k < n; ++k) hist[k] = 0.0;
    for (int i = -radius; i <= radius; ++i) {
        const int y = r + i;
        if (y <= 0 || y >= h - 1) continue;
        for (int j = -radius; j <= radius; ++j) {
            const int x = c + j;
            if (x <= 0 || x >= w - 1) continue;
            const double dx = (double)img[(size_t)y * w + x + 1] - (double)img[(size_t)y * w + x - 1];
            const double dy = (double)img[(size_t)(y - 1) * w + x] - (double)img[(size_t)(y + 1) * w + x];
            const double wgt = exp((double)(i * i + j * j) * expf_scale);
            double ori = fmod(atan2(dy, dx) * (180.0 / 3.141592653589793), 360.0);
            if (ori < 0) ori += 360.0;
            if (ori >= 360.0) ori -= 360.0;
            const double mag = sqrt(dx * dx + dy * dy);
            int b = (int)rint((n / 360.0) * ori);
            if (b >= n) b -= n;
            if (b < 0) b += n;
            hist[b] += wgt * mag;
        }
    }
    double omax = 0.0;
    for (int k = 0; k < n; ++k) {
        const double m2 = hist[(k + n - 2) % n], p2 = hist[(k + 2) % n];
        const double m1 = hist[(k + n - 1) % n], p1 = hist[(k + 1) % n];
        sm[k] = (m2 + p2) * (1.0 / 16) + (m1 + p1) * (4.0 / 16) + hist[k] * (6.0 / 16);
        if (k == 0 || sm[k] > omax) omax = sm[k];
    }
    const double mag_thr = omax * 0.8;
    int np = 0;
    for (int j = 0; j < n; ++j) {
        const double lft = sm[(j + n - 1) % n], rgt = sm[(j + 1) % n];
        if (sm[j] > lft && sm[j] > rgt && sm[j] >= mag_thr) {
            double bin = j + 0.5 * (lft - rgt) / (lft - 2 * sm[j] + rgt);
            bin = bin < 0 ? n + bin : (bin >= n ? bin - n : bin);
            double angle = 360.0 - (360.0 / n) * bin;
            if (fabs(angle - 360.0) < FLT_EPS) angle = 0.0;
            angles[np++] = angle;
        }
    }
    return np;
}

/* detect() of sift_oracle.py for the candidates of one octave, in the order given (layer,
 * then row major): refinement, contrast / edge tests, orientation peaks.  dogs: NL+2 levels,
 * gauss: NL+3 levels of the octave, cand [n][3] = (layer, r, c).  kps [cap][6] = x, y, size,
 * angle, |contrast|, packed octave in the coordinates of the DOUBLED image (detect() halves them
 * afterwards).  Returns the number of keypoints (may exceed cap: only cap are stored). */
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
#pragma omp parallel for schedule(dynamic, 64)
    for (int k = 0; k < n; ++k) {
        refined_t R;
        if (!adjust_local_extrema(dogs, h, w, cand[3 * k], cand[3 * k + 1], cand[3 * k + 2], &R)) continue;
        const double size = sigma0 * pow(2.0, (R.layer + R.xi) / NL) * (double)(1 << o) * 2.0;
        const double px = (R.c + R.xc) * (double)(1 << o), py = (R.r + R.xr) * (double)(1 << o);
        const int octave = o + (R.layer << 8) + ((int)rint((R.xi + 0.5) * 255) << 16);
        const double scl_octv = size * 0.5 / (double)(1 << o);
        double angles[ORI_BINS];
        const int np = orientation_peaks(gauss[R.layer], h, w, R.r, R.c, (int)rint(4.5 * scl_octv),
                                         1.5 * scl_octv, angles);
        double *dstp = loc + (size_t)k * 24;
        if (np > 4) {
            more[k] = (double *)malloc(sizeof(double) * 6 * (size_t)np);
            dstp = more[k];
        }
        for (int j = 0; j < np; ++j) {
            double *q = dstp + 6 * j;
            q[0] = px; q[1] = py; q[2] = size; q[3] = angles[j]; q[4] = fabs(R.contr); q[5] = (double)octave;
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

/* sift_oracle.descriptor (float64 throughout) */
static void descriptor_one(const float *img, int h, int w, double ptx, double pty, double ori, double scl,
                           uint8_t *out)
{
    const int d = DESCR_W, n = DESCR_N;
    const int px = (int)rint(ptx), py = (int)rint(pty);
    double cos_t = cos(ori * (3.141592653589793 / 180.0)), sin_t = sin(ori * (3.141592653589793 / 180.0));
    const double bins_per_rad = n / 360.0, exp_scale = -1.0 / (d * d * 0.5);
    const double hist_width = 3.0 * scl;
    int radius = (int)rint(hist_width * 1.4142135623730951 * (d + 1) * 0.5);
    const int diag = (int)sqrt((double)w * w + (double)h * h);
    radius = radius < diag ? radius : diag;
    cos_t /= hist_width;
    sin_t /= hist_width;
    double hist[(DESCR_W + 2) * (DESCR_W + 2) * (DESCR_N + 2)];
    memset(hist, 0, sizeof(hist));
    for (int i = -radius; i <= radius; ++i) {
        for (int j = -radius; j <= radius; ++j) {
            const double c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            const double rbin = r_rot + d / 2 - 0.5, cbin = c_rot + d / 2 - 0.5;
            const int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < h - 1 && c > 0 && c < w - 1))
                continue;
            const double dx = (double)img[(size_t)r * w + c + 1] - (double)img[(size_t)r * w + c - 1];
            const double dy = (double)img[(size_t)(r - 1) * w + c] - (double)img[(size_t)(r + 1) * w + c];
            const double wgt = exp((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            double og = fmod(atan2(dy, dx) * (180.0 / 3.141592653589793), 360.0);
            if (og < 0) og += 360.0;
            if (og >= 360.0) og -= 360.0;
            const double mag = sqrt(dx * dx + dy * dy) * wgt;
            const double obin = (og - ori) * bins_per_rad;
            const double fr0 = floor(rbin), fc0 = floor(cbin), fo0 = floor(obin);
            const int r0 = (int)fr0, c0 = (int)fc0;
            int o0 = (int)fo0;
            const double fr = rbin - fr0, fc = cbin - fc0, fo = obin - fo0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const double v_r1 = mag * fr, v_r0 = mag - v_r1;
            const double v_rc11 = v_r1 * fc, v_rc10 = v_r1 - v_rc11;
            const double v_rc01 = v_r0 * fc, v_rc00 = v_r0 - v_rc01;
            const double vv[4] = {v_rc00, v_rc01, v_rc10, v_rc11};
            for (int q4 = 0; q4 < 4; ++q4) {
                const int rr = r0 + 1 + (q4 >> 1), cc = c0 + 1 + (q4 & 1);
                const int base = (rr * (d + 2) + cc) * (n + 2) + o0;
                const double v1 = vv[q4] * fo;
                hist[base] += vv[q4] - v1;
                hist[base + 1] += v1;
            }
        }
    }
    double dst[DESCR_W * DESCR_W * DESCR_N], sq = 0.0;
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            double *cell = hist + ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            cell[0] += cell[n];
            cell[1] += cell[n + 1];
            for (int k = 0; k < n; ++k) {
                dst[(i * d + j) * n + k] = cell[k];
                sq += cell[k] * cell[k];
            }
        }
    const double thr = sqrt(sq) * 0.2;
    double sq2 = 0.0;
    for (int k = 0; k < d * d * n; ++k) {
        dst[k] = dst[k] < thr ? dst[k] : thr;
        sq2 += dst[k] * dst[k];
    }
    const double nrm = 512.0 / fmax(sqrt(sq2), FLT_EPS);
    for (int k = 0; k < d * d * n; ++k) {
        double x = rint(dst[k] * nrm);
        x = x < 0 ? 0 : (x > 255 ? 255 : x);
        out[k] = (uint8_t)x;
    }
}

/* descriptors of n keypoints that live on one Gaussian level: par [n][4] = ptx, pty, ori, scl in
 * the coordinates of that level; desc [n][128] */
int oracle_sift_descriptors(const float *img, int h, int w, const double *par, int n, uint8_t *desc,
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

/* ---------------------------------------------------------------------------------------------
 * The whole detector + descriptor in C (sift_oracle.detect_and_compute without its numpy glue):
 * grey image -> x2 bilinear -> Gaussian / DoG pyramid -> 26-neighbour extrema -> refinement,
 * orientation peaks -> descriptors.  OpenMP over rows / candidates / keypoints.  This is the CPU
 * baseline of bench.py's SIFT section; tests/test_oracle.py checks it against the numpy oracle.
 * Output order: octave, then layer, then row major, then peak -- the order detect() appends in.
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
    /* detectAndCompute: first octave is -1 -> input-image coordinates; KeyPoint fields are float32 */
#pragma omp parallel for schedule(dynamic, 16)
    for (int k = 0; k < total; ++k) {
        double *q = kps + (size_t)k * 6;
        int oc = (int)q[5];
        oc = (oc & ~255) | ((oc - 1) & 255);
        q[5] = (double)oc;
        q[0] = (double)(float)(q[0] * 0.5); q[1] = (double)(float)(q[1] * 0.5);
        q[2] = (double)(float)(q[2] * 0.5); q[3] = (double)(float)q[3]; q[4] = (double)(float)q[4];
        int octave = oc & 255, layer = (oc >> 8) & 255;
        if (octave >= 128) octave |= -128;
        const double scale = octave >= 0 ? 1.0 / (double)(1 << octave) : (double)(1 << -octave);
        double a = 360.0 - q[3];
        if (fabs(a - 360.0) < FLT_EPS) a = 0.0;
        const int oi = octave + 1;
        descriptor_one(gauss[oi * (NL + 3) + layer], oh[oi], ow[oi], q[0] * scale, q[1] * scale, a,
                       q[2] * scale * 0.5, desc + (size_t)k * 128);
    }
    for (int q = 0; q < n_oct * (NL + 3); ++q) free(gauss[q]);
    for (int q = 0; q < n_oct * (NL + 2); ++q) free(dogs[q]);
    free(gauss); free(dogs); free(oh); free(ow); free(tmp);
    return total;
}
