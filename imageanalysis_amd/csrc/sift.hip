// K1 -- SIFT detector + descriptor for gfx950 (MI355X).
//
// Replaces cv2.SIFT_create().detectAndCompute(scaled, None) on the reference's path
// (scripts/lib/image.py:235-237,324).  OpenCV itself is a third-party native that is not part
// of the reference tree; this is the published algorithm (Lowe 2004) with OpenCV 4.x defaults:
// 3 layers/octave, contrast 0.04, edge 10, sigma 1.6, image doubled (first octave -1), border 5,
// 36-bin orientation histogram (peak ratio 0.8), 4x4x8 descriptor, clip 0.2, x512 -> u8.
//
// Stages (all enqueued on one stream, no host round trip inside):
//   gray_up2x      BGR u8 -> gray u8 (fixed point) -> x2 bilinear -> f32                HBM
//   blur_h/blur_v  separable Gaussian, BORDER_REFLECT_101, f32, taps in fixed order     HBM
//   downsample, dog                                                                     HBM
//   extrema        26-neighbour test on the DoG stack -> candidate list (atomic append)
//   refine_orient  one thread per candidate: 3-D quadratic fit (<=5 steps), contrast/edge
//                  tests, orientation histogram + peaks -> keypoints (atomic append)
//   descriptor     one wave per keypoint: rotated 4x4x8 trilinear histogram in LDS (f64
//                  atomics), clip/normalise/quantise
// The per-pixel arithmetic of the pyramid uses explicit round-to-nearest mul/add (no FMA
// contraction) in the same tap order as the CPU oracle so the pyramids are bit-identical and
// the keypoint sets can be compared one to one; keypoints are appended in nondeterministic
// order and put into the canonical (octave, layer, y, x, angle) order by the host wrapper.
// Pyramid traffic: 6 Gaussian + 5 DoG f32 levels per octave written once, read once
// (~469 B per detect-resolution pixel).
#include "iamx_common.h"

namespace {

constexpr int NL = 3;                 // nOctaveLayers
constexpr int BORDER = 5;
constexpr int MAX_STEPS = 5;
constexpr int ORI_BINS = 36;
constexpr int MAX_TAPS = 33;

struct Taps {
    int r;
    float k[MAX_TAPS];
};

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    p = p < 0 ? -p : p;
    p %= period;
    return p >= n ? period - p : p;
}

__global__ __launch_bounds__(256) void gray_up2x_kernel(const uint8_t *__restrict__ src, int h,
                                                        int w, int ch, float *__restrict__ dst)
{
    const int W2 = 2 * w, H2 = 2 * h;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)W2 * H2) return;
    const int x = (int)(i % W2), y = (int)(i / W2);
    auto tap = [](int d, int n, int &i0, int &i1, float &t) {
        const float f = __fsub_rn(__fmul_rn(__fadd_rn((float)d, 0.5f), 0.5f), 0.5f);
        i0 = (int)floorf(f);
        t = __fsub_rn(f, (float)i0);
        if (i0 < 0) { i0 = 0; t = 0.f; }
        if (i0 >= n - 1) { i0 = n - 1; t = 0.f; }
        i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    };
    int x0, x1, y0, y1;
    float tx, ty;
    tap(x, w, x0, x1, tx);
    tap(y, h, y0, y1, ty);
    auto gray = [&](int yy, int xx) -> float {
        const uint8_t *p = src + ((int64_t)yy * w + xx) * ch;
        if (ch == 1) return (float)p[0];
        const int v = ((int)p[0] * 1868 + (int)p[1] * 9617 + (int)p[2] * 4899 + 8192) >> 14;
        return (float)v;
    };
    const float omtx = __fsub_rn(1.f, tx), omty = __fsub_rn(1.f, ty);
    const float top = __fadd_rn(__fmul_rn(gray(y0, x0), omtx), __fmul_rn(gray(y0, x1), tx));
    const float bot = __fadd_rn(__fmul_rn(gray(y1, x0), omtx), __fmul_rn(gray(y1, x1), tx));
    dst[i] = __fadd_rn(__fmul_rn(top, omty), __fmul_rn(bot, ty));
}

template <bool VERTICAL>
__global__ __launch_bounds__(256) void blur_kernel(const float *__restrict__ src, int h, int w,
                                                   Taps T, float *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)w * h) return;
    const int x = (int)(i % w), y = (int)(i / w);
    float acc = 0.f;
    for (int t = -T.r; t <= T.r; ++t) {
        float v;
        if (VERTICAL) v = src[(int64_t)reflect101(y + t, h) * w + x];
        else          v = src[(int64_t)y * w + reflect101(x + t, w)];
        acc = __fadd_rn(acc, __fmul_rn(v, T.k[t + T.r]));
    }
    dst[i] = acc;
}

__global__ __launch_bounds__(256) void downsample_kernel(const float *__restrict__ src, int sw,
                                                         int dh, int dw, float *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)dw * dh) return;
    const int x = (int)(i % dw), y = (int)(i / dw);
    dst[i] = src[(int64_t)(2 * y) * sw + 2 * x];
}

__global__ __launch_bounds__(256) void dog_kernel(const float *__restrict__ a,
                                                  const float *__restrict__ b, int64_t n,
                                                  float *__restrict__ d)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = __fsub_rn(b[i], a[i]);
}

struct Cand {
    int o, layer, r, c;
};

// one launch per (octave, layer): prv/cur/nxt DoG images
__global__ __launch_bounds__(256) void extrema_kernel(const float *__restrict__ prv,
                                                      const float *__restrict__ cur,
                                                      const float *__restrict__ nxt, int h, int w,
                                                      int o, int layer, float threshold,
                                                      Cand *__restrict__ cand, int cap,
                                                      int *__restrict__ count)
{
    const int iw = w - 2 * BORDER, ih = h - 2 * BORDER;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)iw * ih) return;
    const int c = (int)(i % iw) + BORDER, r = (int)(i / iw) + BORDER;
    const float val = cur[(int64_t)r * w + c];
    if (!(fabsf(val) > threshold)) return;
    bool is_max = val > 0.f, is_min = val < 0.f;
    if (!is_max && !is_min) return;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
            const int64_t q = (int64_t)(r + dr) * w + (c + dc);
            const float a = prv[q], b = nxt[q], m = cur[q];
            is_max = is_max && val >= a && val >= b && val >= m;
            is_min = is_min && val <= a && val <= b && val <= m;
        }
    }
    if (is_max || is_min) {
        const int k = atomicAdd(count, 1);
        if (k < cap) cand[k] = Cand{o, layer, r, c};
    }
}

struct Pyr {
    // per octave: pointers of the 6 Gaussian and 5 DoG levels, dims
    float *g[6];
    float *d[5];
    int h, w;
};
constexpr int MAX_OCT = 16;
struct PyrTable {
    Pyr oct[MAX_OCT];
    int n_oct;
};

__device__ bool solve3(double A[3][3], double b[3], double x[3])
{
    // Gaussian elimination with partial pivoting (H.solve(dD, DECOMP_LU))
    int p[3] = {0, 1, 2};
    for (int k = 0; k < 3; ++k) {
        int piv = k;
        double best = fabs(A[p[k]][k]);
        for (int i = k + 1; i < 3; ++i)
            if (fabs(A[p[i]][k]) > best) { best = fabs(A[p[i]][k]); piv = i; }
        if (best < 1e-300) return false;
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
    return true;
}

__device__ __forceinline__ int round_half_even(double v)
{
    return (int)rint(v);
}

__global__ __launch_bounds__(64) void refine_orient_kernel(PyrTable T, const Cand *__restrict__ cand,
                                                           const int *__restrict__ n_cand, int cap_c,
                                                           float contrast_threshold,
                                                           float edge_threshold, float sigma,
                                                           float *__restrict__ kp, int cap_k,
                                                           int *__restrict__ n_kp)
{
    const int idx = blockIdx.x * 64 + threadIdx.x;
    const int total = *n_cand < cap_c ? *n_cand : cap_c;
    if (idx >= total) return;
    const Cand cd = cand[idx];
    const Pyr &P = T.oct[cd.o];
    const int h = P.h, w = P.w;
    int layer = cd.layer, r = cd.r, c = cd.c;
    const float img_scale = 1.f / 255.f;
    const float deriv_scale = img_scale * 0.5f, second_scale = img_scale, cross_scale = img_scale * 0.25f;
    double xi = 0, xr = 0, xc = 0;
    int it = 0;
    for (; it < MAX_STEPS; ++it) {
        const float *img = P.d[layer], *prv = P.d[layer - 1], *nxt = P.d[layer + 1];
        auto at = [&](const float *im, int rr, int cc) { return im[(int64_t)rr * w + cc]; };
        const float dDx = __fmul_rn(__fsub_rn(at(img, r, c + 1), at(img, r, c - 1)), deriv_scale);
        const float dDy = __fmul_rn(__fsub_rn(at(img, r + 1, c), at(img, r - 1, c)), deriv_scale);
        const float dDs = __fmul_rn(__fsub_rn(at(nxt, r, c), at(prv, r, c)), deriv_scale);
        const float v2 = __fmul_rn(at(img, r, c), 2.f);
        const float dxx = __fmul_rn(__fsub_rn(__fadd_rn(at(img, r, c + 1), at(img, r, c - 1)), v2), second_scale);
        const float dyy = __fmul_rn(__fsub_rn(__fadd_rn(at(img, r + 1, c), at(img, r - 1, c)), v2), second_scale);
        const float dss = __fmul_rn(__fsub_rn(__fadd_rn(at(nxt, r, c), at(prv, r, c)), v2), second_scale);
        const float dxy = __fmul_rn(__fadd_rn(__fsub_rn(__fsub_rn(at(img, r + 1, c + 1), at(img, r + 1, c - 1)), at(img, r - 1, c + 1)), at(img, r - 1, c - 1)), cross_scale);
        const float dxs = __fmul_rn(__fadd_rn(__fsub_rn(__fsub_rn(at(nxt, r, c + 1), at(nxt, r, c - 1)), at(prv, r, c + 1)), at(prv, r, c - 1)), cross_scale);
        const float dys = __fmul_rn(__fadd_rn(__fsub_rn(__fsub_rn(at(nxt, r + 1, c), at(nxt, r - 1, c)), at(prv, r + 1, c)), at(prv, r - 1, c)), cross_scale);
        double A[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        double b[3] = {dDx, dDy, dDs}, X[3];
        if (!solve3(A, b, X)) return;
        xc = -X[0]; xr = -X[1]; xi = -X[2];
        if (fabs(xi) < 0.5 && fabs(xr) < 0.5 && fabs(xc) < 0.5) break;
        if (fabs(xi) > 2147483647.0 / 3 || fabs(xr) > 2147483647.0 / 3 || fabs(xc) > 2147483647.0 / 3)
            return;
        c += round_half_even(xc);
        r += round_half_even(xr);
        layer += round_half_even(xi);
        if (layer < 1 || layer > NL || c < BORDER || c >= w - BORDER || r < BORDER || r >= h - BORDER)
            return;
    }
    if (it >= MAX_STEPS) return;
    double contr;
    {
        const float *img = P.d[layer], *prv = P.d[layer - 1], *nxt = P.d[layer + 1];
        auto at = [&](const float *im, int rr, int cc) { return im[(int64_t)rr * w + cc]; };
        const double dDx = (double)__fmul_rn(__fsub_rn(at(img, r, c + 1), at(img, r, c - 1)), deriv_scale);
        const double dDy = (double)__fmul_rn(__fsub_rn(at(img, r + 1, c), at(img, r - 1, c)), deriv_scale);
        const double dDs = (double)__fmul_rn(__fsub_rn(at(nxt, r, c), at(prv, r, c)), deriv_scale);
        const double t = dDx * xc + dDy * xr + dDs * xi;
        contr = (double)at(img, r, c) * (double)img_scale + t * 0.5;
        if (fabs(contr) * NL < (double)contrast_threshold) return;
        const double v2 = (double)at(img, r, c) * 2.0;
        const double dxx = ((double)at(img, r, c + 1) + (double)at(img, r, c - 1) - v2) * (double)second_scale;
        const double dyy = ((double)at(img, r + 1, c) + (double)at(img, r - 1, c) - v2) * (double)second_scale;
        const double dxy = ((double)at(img, r + 1, c + 1) - (double)at(img, r + 1, c - 1)
                            - (double)at(img, r - 1, c + 1) + (double)at(img, r - 1, c - 1)) * (double)cross_scale;
        const double tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        const double e = edge_threshold;
        if (det <= 0 || tr * tr * e >= (e + 1) * (e + 1) * det) return;
    }
    const int o = cd.o;
    const double size = (double)sigma * exp2((layer + xi) / NL) * (double)(1 << o) * 2.0;
    const double px = (c + xc) * (double)(1 << o), py = (r + xr) * (double)(1 << o);
    const int octave = o + (layer << 8) + (round_half_even((xi + 0.5) * 255) << 16);
    const double scl_octv = size * 0.5 / (double)(1 << o);

    // ---- orientation histogram on the Gaussian level
    const float *g = P.g[layer];
    const int radius = round_half_even(4.5 * scl_octv);
    const double osig = 1.5 * scl_octv;
    const double expf_scale = -1.0 / (2.0 * osig * osig);
    double hist[ORI_BINS];
    for (int k = 0; k < ORI_BINS; ++k) hist[k] = 0.0;
    for (int i = -radius; i <= radius; ++i) {
        const int y = r + i;
        if (y <= 0 || y >= h - 1) continue;
        for (int j = -radius; j <= radius; ++j) {
            const int x = c + j;
            if (x <= 0 || x >= w - 1) continue;
            const double dx = (double)g[(int64_t)y * w + x + 1] - (double)g[(int64_t)y * w + x - 1];
            const double dy = (double)g[(int64_t)(y - 1) * w + x] - (double)g[(int64_t)(y + 1) * w + x];
            const double wgt = exp((double)(i * i + j * j) * expf_scale);
            double ori = atan2(dy, dx) * (180.0 / 3.141592653589793);
            if (ori < 0) ori += 360.0;
            if (ori >= 360.0) ori -= 360.0;
            const double mag = sqrt(dx * dx + dy * dy);
            int b = round_half_even((ORI_BINS / 360.0) * ori);
            if (b >= ORI_BINS) b -= ORI_BINS;
            if (b < 0) b += ORI_BINS;
            hist[b] += wgt * mag;
        }
    }
    double sm[ORI_BINS];
    double omax = 0.0;
    for (int k = 0; k < ORI_BINS; ++k) {
        const double m2 = hist[(k + ORI_BINS - 2) % ORI_BINS], p2 = hist[(k + 2) % ORI_BINS];
        const double m1 = hist[(k + ORI_BINS - 1) % ORI_BINS], p1 = hist[(k + 1) % ORI_BINS];
        sm[k] = (m2 + p2) * (1.0 / 16) + (m1 + p1) * (4.0 / 16) + hist[k] * (6.0 / 16);
        omax = sm[k] > omax ? sm[k] : omax;
    }
    const double mag_thr = omax * 0.8;
    for (int j = 0; j < ORI_BINS; ++j) {
        const double lft = sm[(j + ORI_BINS - 1) % ORI_BINS], rgt = sm[(j + 1) % ORI_BINS];
        if (sm[j] > lft && sm[j] > rgt && sm[j] >= mag_thr) {
            double bin = j + 0.5 * (lft - rgt) / (lft - 2 * sm[j] + rgt);
            bin = bin < 0 ? ORI_BINS + bin : (bin >= ORI_BINS ? bin - ORI_BINS : bin);
            double angle = 360.0 - (360.0 / ORI_BINS) * bin;
            if (fabs(angle - 360.0) < 1.1920929e-07) angle = 0.0;
            const int k = atomicAdd(n_kp, 1);
            if (k < cap_k) {
                float *q = kp + (int64_t)k * 8;
                // first octave is -1: report in input-image pixels (detectAndCompute)
                q[0] = (float)(px * 0.5);
                q[1] = (float)(py * 0.5);
                q[2] = (float)(size * 0.5);
                q[3] = (float)angle;
                q[4] = (float)fabs(contr);
                const int oct_out = (octave & ~255) | ((octave - 1) & 255);
                q[5] = __int_as_float(oct_out);
                q[6] = __int_as_float(o * 256 + layer);        // pyramid address for the descriptor
                q[7] = 0.f;
            }
        }
    }
}

// one wave per keypoint
__global__ __launch_bounds__(256) void descriptor_kernel(PyrTable T, const float *__restrict__ kp,
                                                         const int *__restrict__ n_kp, int cap_k,
                                                         uint8_t *__restrict__ desc)
{
    constexpr int d = 4, n = 8;
    constexpr int HB = (d + 2) * (d + 2) * (n + 2);       // 360
    __shared__ double hist_s[4][HB];
    __shared__ double red_s[4][2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + wave;
    const int total = *n_kp < cap_k ? *n_kp : cap_k;
    double *hist = hist_s[wave];
    for (int i = lane; i < HB; i += 64) hist[i] = 0.0;
    __syncthreads();
    if (k < total) {
        const float *q = kp + (int64_t)k * 8;
        const int addr = __float_as_int(q[6]);
        const int o = addr >> 8, layer = addr & 255;
        const Pyr &P = T.oct[o];
        const float *img = P.g[layer];
        const int h = P.h, w = P.w;
        // keypoint in the coordinates of its octave: x * scale with scale = 2^-(o-1)
        const double scale = o >= 1 ? 1.0 / (double)(1 << (o - 1)) : 2.0;
        const double ptx = (double)q[0] * scale, pty = (double)q[1] * scale;
        double ori = 360.0 - (double)q[3];
        if (fabs(ori - 360.0) < 1.1920929e-07) ori = 0.0;
        const double scl = (double)q[2] * scale * 0.5;
        const int px = round_half_even(ptx), py = round_half_even(pty);
        double cos_t = cos(ori * (3.141592653589793 / 180.0)), sin_t = sin(ori * (3.141592653589793 / 180.0));
        const double bins_per_rad = n / 360.0, exp_scale = -1.0 / (d * d * 0.5);
        const double hist_width = 3.0 * scl;
        int radius = round_half_even(hist_width * 1.4142135623730951 * (d + 1) * 0.5);
        const int diag = (int)sqrt((double)w * w + (double)h * h);
        radius = radius < diag ? radius : diag;
        cos_t /= hist_width;
        sin_t /= hist_width;
        const int side = 2 * radius + 1;
        const int64_t nsamp = (int64_t)side * side;
        for (int64_t s = lane; s < nsamp; s += 64) {
            const int i = (int)(s / side) - radius, j = (int)(s % side) - radius;
            const double c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            const double rbin = r_rot + d / 2 - 0.5, cbin = c_rot + d / 2 - 0.5;
            const int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < h - 1 && c > 0 && c < w - 1))
                continue;
            const double dx = (double)img[(int64_t)r * w + c + 1] - (double)img[(int64_t)r * w + c - 1];
            const double dy = (double)img[(int64_t)(r - 1) * w + c] - (double)img[(int64_t)(r + 1) * w + c];
            const double wgt = exp((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            double og = atan2(dy, dx) * (180.0 / 3.141592653589793);
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
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int rr = r0 + 1 + (q4 >> 1), cc = c0 + 1 + (q4 & 1);
                const int base = (rr * (d + 2) + cc) * (n + 2) + o0;
                const double v1 = vv[q4] * fo;
                atomicAdd(&hist[base], vv[q4] - v1);
                atomicAdd(&hist[base + 1], v1);
            }
        }
    }
    __syncthreads();
    if (k < total) {
        // circular orientation bins, then the 128 values: lanes 0..63 hold 2 each
        double v[2];
        double sq = 0.0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = lane * 2 + e;             // (i*d + j)*n + kk
            const int cell = t / n, kk = t % n;
            const int i = cell / d, j = cell % d;
            const int base = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            double x = hist[base + kk];
            if (kk == 0) x += hist[base + n];
            if (kk == 1) x += hist[base + n + 1];
            v[e] = x;
            sq += x * x;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m);
        const double thr = sqrt(sq) * 0.2;
        double sq2 = 0.0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            v[e] = v[e] < thr ? v[e] : thr;
            sq2 += v[e] * v[e];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sq2 += __shfl_xor(sq2, m);
        const double nrm = 512.0 / fmax(sqrt(sq2), 1.1920929e-07);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            double x = rint(v[e] * nrm);
            x = x < 0 ? 0 : (x > 255 ? 255 : x);
            desc[(int64_t)k * 128 + lane * 2 + e] = (uint8_t)x;
        }
    }
    (void)red_s;
}

inline unsigned blocks(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

void gaussian_taps(double sigma, Taps &T)
{
    int ksize = (int)lrint(sigma * 8 + 1) | 1;
    int r = ksize / 2;
    if (r > (MAX_TAPS - 1) / 2) r = (MAX_TAPS - 1) / 2;
    double k[MAX_TAPS], sum = 0;
    for (int i = -r; i <= r; ++i) {
        k[i + r] = exp(-(double)(i * i) / (2.0 * sigma * sigma));
        sum += k[i + r];
    }
    T.r = r;
    for (int i = 0; i < 2 * r + 1; ++i) T.k[i] = (float)(k[i] / sum);
}

struct Layout {
    int n_oct;
    int h[MAX_OCT], w[MAX_OCT];
    int64_t g_off[MAX_OCT][6], d_off[MAX_OCT][5];
    int64_t tmp_off, cand_off, count_off, total;
};

Layout make_layout(int height, int width, int cap_c)
{
    Layout L;
    int H = 2 * height, W = 2 * width;
    int n_oct = (int)lrint(log((double)(H < W ? H : W)) / log(2.0) - 2) + 1;
    if (n_oct < 1) n_oct = 1;
    if (n_oct > MAX_OCT) n_oct = MAX_OCT;
    L.n_oct = n_oct;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    for (int o = 0; o < n_oct; ++o) {
        L.h[o] = H; L.w[o] = W;
        for (int i = 0; i < 6; ++i) L.g_off[o][i] = take((int64_t)H * W * 4);
        for (int i = 0; i < 5; ++i) L.d_off[o][i] = take((int64_t)H * W * 4);
        H /= 2; W /= 2;
        if (H < 1 || W < 1) { L.n_oct = o + 1; break; }
    }
    L.tmp_off = take((int64_t)L.h[0] * L.w[0] * 4);
    L.cand_off = take((int64_t)cap_c * sizeof(Cand));
    L.count_off = take(256);
    L.total = off;
    return L;
}

constexpr int CAP_CAND = 1 << 21;

}  // namespace

extern "C" int64_t iamx_sift_workspace_bytes(int height, int width)
{
    if (height < 1 || width < 1) return 0;
    return make_layout(height, width, CAP_CAND).total;
}

extern "C" int iamx_sift_detect(const uint8_t *image, int height, int width, int channels,
                                float contrast_threshold, float edge_threshold, float sigma,
                                void *workspace, int64_t workspace_bytes, float *kp, uint8_t *desc,
                                int cap, int32_t *n_out, void *stream)
{
    IAMX_REQUIRE(image && workspace && kp && desc && n_out, "null pointer");
    IAMX_REQUIRE(height >= 2 && width >= 2 && (channels == 1 || channels == 3) && cap > 0,
                 "bad image size / channels / capacity");
    const Layout L = make_layout(height, width, CAP_CAND);
    IAMX_REQUIRE(workspace_bytes >= L.total, "workspace too small (iamx_sift_workspace_bytes)");
    hipStream_t st = iamx::as_stream(stream);
    char *ws = static_cast<char *>(workspace);
    PyrTable T;
    T.n_oct = L.n_oct;
    for (int o = 0; o < L.n_oct; ++o) {
        T.oct[o].h = L.h[o];
        T.oct[o].w = L.w[o];
        for (int i = 0; i < 6; ++i) T.oct[o].g[i] = reinterpret_cast<float *>(ws + L.g_off[o][i]);
        for (int i = 0; i < 5; ++i) T.oct[o].d[i] = reinterpret_cast<float *>(ws + L.d_off[o][i]);
    }
    float *tmp = reinterpret_cast<float *>(ws + L.tmp_off);
    Cand *cand = reinterpret_cast<Cand *>(ws + L.cand_off);
    int *n_cand = reinterpret_cast<int *>(ws + L.count_off);
    (void)hipMemsetAsync(n_cand, 0, 4, st);
    (void)hipMemsetAsync(n_out, 0, 4, st);

    // layer sigmas (Lowe / OpenCV buildGaussianPyramid)
    double sig[6];
    sig[0] = sigma;
    const double kf = pow(2.0, 1.0 / NL);
    for (int i = 1; i < NL + 3; ++i) {
        const double sp = pow(kf, (double)(i - 1)) * sigma, stt = sp * kf;
        sig[i] = sqrt(stt * stt - sp * sp);
    }
    auto blur = [&](const float *src, float *dst, int h, int w, double s) {
        Taps tp;
        gaussian_taps(s, tp);
        const unsigned g = blocks((int64_t)h * w, 256);
        hipLaunchKernelGGL(blur_kernel<false>, dim3(g), dim3(256), 0, st, src, h, w, tp, tmp);
        hipLaunchKernelGGL(blur_kernel<true>, dim3(g), dim3(256), 0, st, tmp, h, w, tp, dst);
    };
    // base image: gray -> x2 -> blur(sqrt(sigma^2 - 1))
    {
        const int H = L.h[0], W = L.w[0];
        float *up = T.oct[0].d[0];             // scratch (overwritten by the DoG later)
        hipLaunchKernelGGL(gray_up2x_kernel, dim3(blocks((int64_t)H * W, 256)), dim3(256), 0, st,
                           image, height, width, channels, up);
        const double sd = sqrt(fmax((double)sigma * sigma - 1.0, 0.01));
        blur(up, T.oct[0].g[0], H, W, sd);
    }
    const float threshold = floorf(0.5f * contrast_threshold / NL * 255.f);
    for (int o = 0; o < L.n_oct; ++o) {
        const int H = L.h[o], W = L.w[o];
        const int64_t npx = (int64_t)H * W;
        if (o > 0)
            hipLaunchKernelGGL(downsample_kernel, dim3(blocks(npx, 256)), dim3(256), 0, st,
                               T.oct[o - 1].g[NL], L.w[o - 1], H, W, T.oct[o].g[0]);
        for (int i = 1; i < NL + 3; ++i) blur(T.oct[o].g[i - 1], T.oct[o].g[i], H, W, sig[i]);
        for (int i = 0; i < NL + 2; ++i)
            hipLaunchKernelGGL(dog_kernel, dim3(blocks(npx, 256)), dim3(256), 0, st, T.oct[o].g[i],
                               T.oct[o].g[i + 1], npx, T.oct[o].d[i]);
        if (H > 2 * BORDER && W > 2 * BORDER) {
            const int64_t inner = (int64_t)(H - 2 * BORDER) * (W - 2 * BORDER);
            for (int layer = 1; layer <= NL; ++layer)
                hipLaunchKernelGGL(extrema_kernel, dim3(blocks(inner, 256)), dim3(256), 0, st,
                                   T.oct[o].d[layer - 1], T.oct[o].d[layer], T.oct[o].d[layer + 1], H,
                                   W, o, layer, threshold, cand, CAP_CAND, n_cand);
        }
    }
    // the number of candidates is only known on the device: launch for the capacity in slabs
    // sized by the largest plausible count (threads beyond *n_cand exit immediately)
    hipLaunchKernelGGL(refine_orient_kernel, dim3(blocks(CAP_CAND, 64)), dim3(64), 0, st, T, cand,
                       n_cand, CAP_CAND, contrast_threshold, edge_threshold, sigma, kp, cap, n_out);
    hipLaunchKernelGGL(descriptor_kernel, dim3(blocks(cap, 4)), dim3(256), 0, st, T, kp, n_out, cap,
                       desc);
    return iamx::check_launch("iamx_sift_detect");
}
