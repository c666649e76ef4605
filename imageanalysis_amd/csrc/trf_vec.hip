// n-vector kernels of the trust-region-reflective outer loop (gfx950), float64.
//
// scipy.optimize.least_squares(method='trf') -- what the reference's Optimizer.run() calls
// (scripts/lib/optimizer.py:352-399) -- works on n-vectors between two LSMR solves: the
// Coleman-Li scaling (scipy/optimize/_lsq/common.py CL_scaling_vector), step_size_to_bound,
// make_strictly_feasible, find_active_constraints, and the sums / scaled copies of trf.py's
// trf_bounds().  With x, g and the candidate steps resident in HBM (ba_solver._trf_device) these
// are O(n) streams, a few dozen per outer iteration; they live here so that the solve contains
// no framework compute kernels.  Reductions use a fixed grid and a fixed tree => deterministic.
//
// Every function mirrors the numpy expression of the SciPy helper it restates, including the
// corner cases the helpers define (a zero step component never limits the step, a point on both
// bounds of a degenerate box moves to the midpoint, ...); tests/test_trf_helpers_gpu.py checks
// them against scipy.optimize._lsq.common itself.
#include "iamx_common.h"
#include <math.h>

namespace {

constexpr int TRF_BLOCKS = 256;       // partial results per reduction
constexpr int MAX_DOTS = 8;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double o = __shfl_xor(v, m);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double o = __shfl_xor(v, m);
        v = o > v ? o : v;
    }
    return v;
}

inline int grid_for(int64_t n)
{
    const int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

#define GRID_STRIDE(i, n) \
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (int64_t)gridDim.x * 256)

// ---- elementwise -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lincomb_kernel(int64_t n, double a, const double *__restrict__ x,
                                                      double b, const double *__restrict__ y, double c,
                                                      const double *__restrict__ z, double *__restrict__ out)
{
    GRID_STRIDE(i, n) {
        double v = a * x[i];
        if (y) v += b * y[i];
        if (z) v += c * z[i];
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void mul_kernel(int64_t n, double s, const double *__restrict__ x,
                                                  const double *__restrict__ y, double *__restrict__ out)
{
    GRID_STRIDE(i, n) out[i] = y ? s * x[i] * y[i] : s * x[i];
}

__global__ __launch_bounds__(256) void sqrt_shift_kernel(int64_t n, const double *__restrict__ x, double shift,
                                                         double *__restrict__ out)
{
    GRID_STRIDE(i, n) out[i] = sqrt(x[i] + shift);
}

__global__ __launch_bounds__(256) void gather_kernel(int64_t n, const double *__restrict__ x,
                                                     const int64_t *__restrict__ idx,
                                                     double *__restrict__ out)
{
    GRID_STRIDE(i, n) out[i] = x[idx[i]];
}

// CL_scaling_vector(x, g, lb, ub) -> v, dv
__global__ __launch_bounds__(256) void cl_scaling_kernel(int64_t n, const double *__restrict__ x,
                                                         const double *__restrict__ g,
                                                         const double *__restrict__ lb,
                                                         const double *__restrict__ ub,
                                                         double *__restrict__ v, double *__restrict__ dv)
{
    GRID_STRIDE(i, n) {
        double vi = 1.0, di = 0.0;
        if (g[i] < 0 && isfinite(ub[i])) { vi = ub[i] - x[i]; di = -1.0; }
        if (g[i] > 0 && isfinite(lb[i])) { vi = x[i] - lb[i]; di = 1.0; }
        v[i] = vi;
        dv[i] = di;
    }
}

// trf.py trf_bounds, top of the outer iteration:
//   v[dv != 0] *= scale_inv[dv != 0];  d = v**0.5 * scale;  diag_h = g * dv * scale;  g_h = d * g
// (scale = 1 / scale_inv)
__global__ __launch_bounds__(256) void trf_scale_kernel(int64_t n, const double *__restrict__ v,
                                                        const double *__restrict__ dv,
                                                        const double *__restrict__ g,
                                                        const double *__restrict__ scale_inv,
                                                        double *__restrict__ v_out, double *__restrict__ d,
                                                        double *__restrict__ diag_h, double *__restrict__ g_h)
{
    GRID_STRIDE(i, n) {
        const double si = scale_inv[i], scale = 1.0 / si;
        double vi = v[i];
        if (dv[i] != 0) vi *= si;
        const double di = sqrt(vi) * scale;
        if (v_out) v_out[i] = vi;
        d[i] = di;
        diag_h[i] = g[i] * dv[i] * scale;
        g_h[i] = di * g[i];
    }
}

// compute_jac_scale: scale_inv = sum(J**2, axis=0)**0.5; first call: zeros become 1,
// later calls: the maximum with the previous scale_inv
__global__ __launch_bounds__(256) void jac_scale_kernel(int64_t n, const double *__restrict__ colsq,
                                                        double *__restrict__ scale_inv, int first)
{
    GRID_STRIDE(i, n) {
        const double c = sqrt(colsq[i]);
        if (first) scale_inv[i] = c == 0 ? 1.0 : c;
        else scale_inv[i] = c > scale_inv[i] ? c : scale_inv[i];
    }
}

__device__ __forceinline__ double bound_step(double x, double s, double lb, double ub)
{
    if (s == 0) return INFINITY;
    const double a = (lb - x) / s, b = (ub - x) / s;
    return a > b ? a : b;                                    // np.maximum (no NaN can occur: s != 0)
}

// step_size_to_bound(x, s, lb, ub): per-block minima
__global__ __launch_bounds__(256) void step_to_bound_kernel(int64_t n, const double *__restrict__ x,
                                                            const double *__restrict__ s,
                                                            const double *__restrict__ lb,
                                                            const double *__restrict__ ub,
                                                            double *__restrict__ partial)
{
    __shared__ double sh[4];
    double m = INFINITY;
    GRID_STRIDE(i, n) {
        const double t = bound_step(x[i], s[i], lb[i], ub[i]);
        m = t < m ? t : m;
    }
    m = wave_min(m);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
        for (int k = 1; k < 4; ++k) r = sh[k] < r ? sh[k] : r;
        partial[blockIdx.x] = r;
    }
}

template <int OP>       // 0: min, 1: max, 2: sum of `width` columns
__global__ __launch_bounds__(256) void final_kernel(const double *__restrict__ partial, int n_partial,
                                                    int width, double *__restrict__ out)
{
    __shared__ double sh[4];
    for (int k = 0; k < width; ++k) {
        double v = OP == 0 ? INFINITY : (OP == 1 ? -INFINITY : 0.0);
        if ((int)threadIdx.x < n_partial) v = partial[(int64_t)k * n_partial + threadIdx.x];
        v = OP == 0 ? wave_min(v) : (OP == 1 ? wave_max(v) : wave_sum(v));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double r = sh[0];
            for (int j = 1; j < 4; ++j)
                r = OP == 0 ? (sh[j] < r ? sh[j] : r) : (OP == 1 ? (sh[j] > r ? sh[j] : r) : r + sh[j]);
            out[k] = r;
        }
    }
}

// trf.py select_step: hits = equal(steps, min_step) * sign(s);  r_h = p_h, r_h[hits != 0] *= -1
__global__ __launch_bounds__(256) void reflect_kernel(int64_t n, const double *__restrict__ x,
                                                      const double *__restrict__ s,
                                                      const double *__restrict__ lb,
                                                      const double *__restrict__ ub, double min_step,
                                                      const double *__restrict__ p_h,
                                                      double *__restrict__ r_h, double *__restrict__ hits)
{
    GRID_STRIDE(i, n) {
        const double t = bound_step(x[i], s[i], lb[i], ub[i]);
        const double sg = s[i] > 0 ? 1.0 : (s[i] < 0 ? -1.0 : 0.0);
        const double h = t == min_step ? sg : 0.0;
        if (hits) hits[i] = h;
        if (r_h) r_h[i] = h != 0 ? -p_h[i] : p_h[i];
    }
}

// in_bounds(x + p, lb, ub): number of components outside, per block
__global__ __launch_bounds__(256) void outside_kernel(int64_t n, const double *__restrict__ x,
                                                      const double *__restrict__ p,
                                                      const double *__restrict__ lb,
                                                      const double *__restrict__ ub,
                                                      double *__restrict__ partial)
{
    __shared__ double sh[4];
    double c = 0.0;
    GRID_STRIDE(i, n) {
        const double t = p ? x[i] + p[i] : x[i];
        c += (t >= lb[i] && t <= ub[i]) ? 0.0 : 1.0;         // (a NaN is outside, like np.all(...))
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// make_strictly_feasible(x + step, lb, ub, rstep=0)
__global__ __launch_bounds__(256) void strictly_feasible_kernel(int64_t n, const double *__restrict__ x,
                                                                const double *__restrict__ step,
                                                                const double *__restrict__ lb,
                                                                const double *__restrict__ ub,
                                                                double *__restrict__ out)
{
    GRID_STRIDE(i, n) {
        const double t = step ? x[i] + step[i] : x[i];
        double r = t;
        if (t <= lb[i]) r = nextafter(lb[i], ub[i]);
        if (t >= ub[i]) r = nextafter(ub[i], lb[i]);
        if (r < lb[i] || r > ub[i]) r = 0.5 * (lb[i] + ub[i]);
        out[i] = r;
    }
}

// make_strictly_feasible(x, lb, ub, rstep) with rstep > 0 (trf.py: the start point): a component
// within rstep * max(1, |bound|) of a finite bound moves that far inside it (the upper bound wins
// where both tests hold, as find_active_constraints writes +1 last); a box too tight for that
// takes its midpoint
__global__ __launch_bounds__(256) void feasible_start_kernel(int64_t n, const double *__restrict__ x,
                                                             const double *__restrict__ lb,
                                                             const double *__restrict__ ub, double rstep,
                                                             double *__restrict__ out)
{
    GRID_STRIDE(i, n) {
        const double t = x[i];
        const double ld = t - lb[i], ud = ub[i] - t;
        const double lt = rstep * fmax(1.0, fabs(lb[i])), ut = rstep * fmax(1.0, fabs(ub[i]));
        double r = t;
        if (isfinite(lb[i]) && ld <= (ud < lt ? ud : lt)) r = lb[i] + lt;
        if (isfinite(ub[i]) && ud <= (ld < ut ? ld : ut)) r = ub[i] - ut;
        if (r < lb[i] || r > ub[i]) r = 0.5 * (lb[i] + ub[i]);
        out[i] = r;
    }
}

// trf.py, before the loop: x * scale_inv / v**0.5 with v[dv != 0] *= scale_inv -- the vector whose
// norm is the first trust-region radius
__global__ __launch_bounds__(256) void scaled_start_kernel(int64_t n, const double *__restrict__ x,
                                                           const double *__restrict__ scale_inv,
                                                           const double *__restrict__ v,
                                                           const double *__restrict__ dv,
                                                           double *__restrict__ out)
{
    GRID_STRIDE(i, n) {
        const double vv = dv[i] != 0.0 ? v[i] * scale_inv[i] : v[i];
        out[i] = x[i] * scale_inv[i] / sqrt(vv);
    }
}

// find_active_constraints(x, lb, ub, rtol) with rtol > 0
__global__ __launch_bounds__(256) void active_kernel(int64_t n, const double *__restrict__ x,
                                                     const double *__restrict__ lb,
                                                     const double *__restrict__ ub, double rtol,
                                                     double *__restrict__ active)
{
    GRID_STRIDE(i, n) {
        const double ld = x[i] - lb[i], ud = ub[i] - x[i];
        const double lt = rtol * fmax(1.0, fabs(lb[i])), ut = rtol * fmax(1.0, fabs(ub[i]));
        double a = 0.0;
        if (isfinite(lb[i]) && ld <= (ud < lt ? ud : lt)) a = -1.0;
        if (isfinite(ub[i]) && ud <= (ld < ut ? ld : ut)) a = 1.0;
        active[i] = a;
    }
}

// ---- several weighted inner products in one pass --------------------------------------------
struct DotArgs {
    const double *a[MAX_DOTS], *b[MAX_DOTS], *w[MAX_DOTS];
    int k;
};

__global__ __launch_bounds__(256) void dots_kernel(int64_t n, DotArgs A, double *__restrict__ partial)
{
    __shared__ double sh[4];
    for (int k = 0; k < A.k; ++k) {
        const double *a = A.a[k], *b = A.b[k], *w = A.w[k];
        double acc = 0.0;
        GRID_STRIDE(i, n) acc += w ? a[i] * w[i] * b[i] : a[i] * b[i];
        acc = wave_sum(acc);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) partial[(int64_t)k * TRF_BLOCKS + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    }
}

__global__ __launch_bounds__(256) void absmax_prod_kernel(int64_t n, const double *__restrict__ x,
                                                          const double *__restrict__ y,
                                                          double *__restrict__ partial)
{
    __shared__ double sh[4];
    double m = 0.0;
    bool bad = false;
    GRID_STRIDE(i, n) {
        const double t = fabs(y ? x[i] * y[i] : x[i]);
        bad |= !(t == t);
        m = t > m ? t : m;
    }
    if (bad) m = NAN;                                        // np.max propagates a NaN
    // NaN-propagating wave maximum
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const double o = __shfl_xor(m, s);
        m = (o != o || m != m) ? NAN : (o > m ? o : m);
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
        for (int k = 1; k < 4; ++k) r = (sh[k] != sh[k] || r != r) ? NAN : (sh[k] > r ? sh[k] : r);
        partial[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(256) void final_nanmax_kernel(const double *__restrict__ partial, int n_partial,
                                                           double *__restrict__ out)
{
    __shared__ double sh[4];
    double m = (int)threadIdx.x < n_partial ? partial[threadIdx.x] : 0.0;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const double o = __shfl_xor(m, s);
        m = (o != o || m != m) ? NAN : (o > m ? o : m);
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
        for (int k = 1; k < 4; ++k) r = (sh[k] != sh[k] || r != r) ? NAN : (sh[k] > r ? sh[k] : r);
        out[0] = r;
    }
}

}  // namespace

#define LAUNCH(kernel, n, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid_for(n)), dim3(256), 0, iamx::as_stream(stream), __VA_ARGS__)

extern "C" int iamx_vec_lincomb(int64_t n, double a, const double *x, double b, const double *y,
                                double c, const double *z, double *out, void *stream)
{
    IAMX_REQUIRE(x && out, "null pointer");
    if (n <= 0) return IAMX_OK;
    LAUNCH(lincomb_kernel, n, n, a, x, b, y, c, z, out);
    return iamx::check_launch("iamx_vec_lincomb");
}

extern "C" int iamx_vec_mul(int64_t n, double s, const double *x, const double *y, double *out,
                            void *stream)
{
    IAMX_REQUIRE(x && out, "null pointer");
    if (n <= 0) return IAMX_OK;
    LAUNCH(mul_kernel, n, n, s, x, y, out);
    return iamx::check_launch("iamx_vec_mul");
}

extern "C" int iamx_vec_gather(int64_t n, const double *x, const int64_t *idx, double *out,
                               void *stream)
{
    IAMX_REQUIRE(x && idx && out, "null pointer");
    IAMX_REQUIRE(x != out, "out must not alias x");
    if (n <= 0) return IAMX_OK;
    LAUNCH(gather_kernel, n, n, x, idx, out);
    return iamx::check_launch("iamx_vec_gather");
}

extern "C" int iamx_vec_sqrt_shift(int64_t n, const double *x, double shift, double *out, void *stream)
{
    IAMX_REQUIRE(x && out, "null pointer");
    if (n <= 0) return IAMX_OK;
    LAUNCH(sqrt_shift_kernel, n, n, x, shift, out);
    return iamx::check_launch("iamx_vec_sqrt_shift");
}

extern "C" int iamx_vec_scratch_doubles(void) { return MAX_DOTS * TRF_BLOCKS; }

extern "C" int iamx_vec_dots(int64_t n, int k, const double *const *a, const double *const *b,
                             const double *const *w, double *out, double *scratch, void *stream)
{
    IAMX_REQUIRE(a && b && out && scratch, "null pointer");
    IAMX_REQUIRE(k >= 1 && k <= MAX_DOTS, "1 <= k <= 8 inner products per call");
    DotArgs A;
    A.k = k;
    for (int i = 0; i < k; ++i) {
        IAMX_REQUIRE(a[i] && b[i], "null vector");
        A.a[i] = a[i];
        A.b[i] = b[i];
        A.w[i] = w ? w[i] : nullptr;
    }
    hipStream_t st = iamx::as_stream(stream);
    hipLaunchKernelGGL(dots_kernel, dim3(TRF_BLOCKS), dim3(256), 0, st, n, A, scratch);
    hipLaunchKernelGGL(final_kernel<2>, dim3(1), dim3(256), 0, st, scratch, TRF_BLOCKS, k, out);
    return iamx::check_launch("iamx_vec_dots");
}

extern "C" int iamx_vec_absmax_prod(int64_t n, const double *x, const double *y, double *out,
                                    double *scratch, void *stream)
{
    IAMX_REQUIRE(x && out && scratch, "null pointer");
    hipStream_t st = iamx::as_stream(stream);
    hipLaunchKernelGGL(absmax_prod_kernel, dim3(TRF_BLOCKS), dim3(256), 0, st, n, x, y, scratch);
    hipLaunchKernelGGL(final_nanmax_kernel, dim3(1), dim3(256), 0, st, scratch, TRF_BLOCKS, out);
    return iamx::check_launch("iamx_vec_absmax_prod");
}

extern "C" int iamx_trf_cl_scaling(int64_t n, const double *x, const double *g, const double *lb,
                                   const double *ub, double *v, double *dv, void *stream)
{
    IAMX_REQUIRE(x && g && lb && ub && v && dv, "null pointer");
    if (n <= 0) return IAMX_OK;
    LAUNCH(cl_scaling_kernel, n, n, x, g, lb, ub, v, dv);
    return iamx::check_launch("iamx_trf_cl_scaling");
}

extern "C" int iamx_trf_scale(int64_t n, const double *v, const double *dv, const double *g,
                              const double *scale_inv, double *v_out, double *d, double *diag_h,
                              double *g_h, void *stream)
{
    IAMX_REQUIRE(v && dv && g && scale_inv && d && diag_h && g_h, "null pointer");
    if (n <= 0) return IAMX_OK;
    LAUNCH(trf_scale_kernel, n, n, v, dv, g, scale_inv, v_out, d, diag_h, g_h);
    return iamx::check_launch("iamx_trf_scale");
}

extern "C" int iamx_trf_jac_scale(int64_t n, const double *colsq, double *scale_inv, int first,
                                  void *stream)
{
    IAMX_REQUIRE(colsq && scale_inv, "null pointer");
    if (n <= 0) return IAMX_OK;
    LAUNCH(jac_scale_kernel, n, n, colsq, scale_inv, first);
    return iamx::check_launch("iamx_trf_jac_scale");
}

extern "C" int iamx_trf_step_to_bound(int64_t n, const double *x, const double *s, const double *lb,
                                      const double *ub, double *out, double *scratch, void *stream)
{
    IAMX_REQUIRE(x && s && lb && ub && out && scratch, "null pointer");
    hipStream_t st = iamx::as_stream(stream);
    hipLaunchKernelGGL(step_to_bound_kernel, dim3(TRF_BLOCKS), dim3(256), 0, st, n, x, s, lb, ub, scratch);
    hipLaunchKernelGGL(final_kernel<0>, dim3(1), dim3(256), 0, st, scratch, TRF_BLOCKS, 1, out);
    return iamx::check_launch("iamx_trf_step_to_bound");
}

extern "C" int iamx_trf_reflect(int64_t n, const double *x, const double *s, const double *lb,
                                const double *ub, double min_step, const double *p_h, double *r_h,
                                double *hits, void *stream)
{
    IAMX_REQUIRE(x && s && lb && ub && (hits || (p_h && r_h)), "null pointer");
    IAMX_REQUIRE(!r_h || p_h, "r_h needs p_h");
    if (n <= 0) return IAMX_OK;
    LAUNCH(reflect_kernel, n, n, x, s, lb, ub, min_step, p_h, r_h, hits);
    return iamx::check_launch("iamx_trf_reflect");
}

extern "C" int iamx_trf_count_outside(int64_t n, const double *x, const double *p, const double *lb,
                                      const double *ub, double *out, double *scratch, void *stream)
{
    IAMX_REQUIRE(x && lb && ub && out && scratch, "null pointer");
    hipStream_t st = iamx::as_stream(stream);
    hipLaunchKernelGGL(outside_kernel, dim3(TRF_BLOCKS), dim3(256), 0, st, n, x, p, lb, ub, scratch);
    hipLaunchKernelGGL(final_kernel<2>, dim3(1), dim3(256), 0, st, scratch, TRF_BLOCKS, 1, out);
    return iamx::check_launch("iamx_trf_count_outside");
}

extern "C" int iamx_trf_strictly_feasible(int64_t n, const double *x, const double *step,
                                          const double *lb, const double *ub, double *out, void *stream)
{
    IAMX_REQUIRE(x && lb && ub && out, "null pointer");
    if (n <= 0) return IAMX_OK;
    LAUNCH(strictly_feasible_kernel, n, n, x, step, lb, ub, out);
    return iamx::check_launch("iamx_trf_strictly_feasible");
}

extern "C" int iamx_trf_feasible_start(int64_t n, const double *x, const double *lb, const double *ub,
                                       double rstep, double *out, void *stream)
{
    IAMX_REQUIRE(x && lb && ub && out, "null pointer");
    IAMX_REQUIRE(rstep > 0, "rstep > 0 (rstep = 0 is iamx_trf_strictly_feasible)");
    if (n <= 0) return IAMX_OK;
    LAUNCH(feasible_start_kernel, n, n, x, lb, ub, rstep, out);
    return iamx::check_launch("iamx_trf_feasible_start");
}

extern "C" int iamx_trf_scaled_start(int64_t n, const double *x, const double *scale_inv,
                                     const double *v, const double *dv, double *out, void *stream)
{
    IAMX_REQUIRE(x && scale_inv && v && dv && out, "null pointer");
    if (n <= 0) return IAMX_OK;
    LAUNCH(scaled_start_kernel, n, n, x, scale_inv, v, dv, out);
    return iamx::check_launch("iamx_trf_scaled_start");
}

extern "C" int iamx_trf_active(int64_t n, const double *x, const double *lb, const double *ub,
                               double rtol, double *active, void *stream)
{
    IAMX_REQUIRE(x && lb && ub && active, "null pointer");
    if (n <= 0) return IAMX_OK;
    LAUNCH(active_kernel, n, n, x, lb, ub, rtol, active);
    return iamx::check_launch("iamx_trf_active");
}
