// K4 -- linear algebra on the device-resident BA Jacobian (gfx950), float64.
//
// The reference hands SciPy a sparsity mask and lets it finite-difference fun()
// (scripts/lib/optimizer.py:142-169,491-501); SciPy's TRF then works on the sparse matrix with
// LSMR (scipy/optimize/_lsq/trf.py:205-400).  Here the Jacobian never leaves HBM: it stays in
// the block form ba_residual_jac_kernel writes (Jc[O][2][7], Jp[O][2][3], Jk[O][2][8]) and the
// operator applications LSMR / the trust-region step need are kernels over those blocks:
//   J v     one thread per observation (gathers the camera's 7 and the point's 3 entries of v)
//   J^T u   camera part: one wave per camera over its contiguous (camera-major) observations,
//           point part: one thread per point over its observation list (CSR built once),
//           calibration part: two-stage block reduction -- no atomics => deterministic
//   column sums of J.^2 (x_scale='jac', scipy compute_jac_scale) -- same traversal.
// All are HBM-bound streams of the Jacobian blocks: 160 B/obs (+128 B with calibration)
// + 16 B/obs of u/y  (one SpMV over 40 M nnz ~ 0.5 GB in CSR; 0.35 GB here).
#include "iamx_common.h"

namespace {

__global__ __launch_bounds__(256) void jv_kernel(const double *__restrict__ Jc,
                                                 const double *__restrict__ Jp,
                                                 const double *__restrict__ Jk,
                                                 const int32_t *__restrict__ cam_idx,
                                                 const int32_t *__restrict__ pt_idx, int64_t n_obs,
                                                 int n_cams, int n_pts,
                                                 const double *__restrict__ x,
                                                 double *__restrict__ y)
{
    const double *xp = x + (int64_t)n_cams * 7;
    const double *xk = xp + (int64_t)n_pts * 3;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n_obs;
         o += (int64_t)gridDim.x * 256) {
        const double *xc = x + (int64_t)cam_idx[o] * 7;
        const double *xq = xp + (int64_t)pt_idx[o] * 3;
        const double *jc = Jc + o * 14;
        const double *jp = Jp + o * 6;
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const double v = xc[k];
            a += jc[k] * v;
            b += jc[7 + k] * v;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = xq[k];
            a += jp[k] * v;
            b += jp[3 + k] * v;
        }
        if (Jk) {
            const double *jk = Jk + o * 16;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double v = xk[k];
                a += jk[k] * v;
                b += jk[8 + k] * v;
            }
        }
        *reinterpret_cast<double2 *>(y + 2 * o) = make_double2(a, b);
    }
}

// camera part of J^T u (SQUARE: column sums of squares): one wave per camera
template <bool SQUARE>
__global__ __launch_bounds__(256) void jt_cam_kernel(const double *__restrict__ Jc,
                                                     const int32_t *__restrict__ cam_ptr,
                                                     int n_cams, const double *__restrict__ u,
                                                     double *__restrict__ out)
{
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_cams) return;
    const int lane = threadIdx.x & 63;
    double acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int o = cam_ptr[c] + lane; o < cam_ptr[c + 1]; o += 64) {
        const double *jc = Jc + (int64_t)o * 14;
        if (SQUARE) {
#pragma unroll
            for (int k = 0; k < 7; ++k) acc[k] += jc[k] * jc[k] + jc[7 + k] * jc[7 + k];
        } else {
            const double2 uu = *reinterpret_cast<const double2 *>(u + 2 * (int64_t)o);
#pragma unroll
            for (int k = 0; k < 7; ++k) acc[k] += jc[k] * uu.x + jc[7 + k] * uu.y;
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) acc[k] += __shfl_xor(acc[k], m);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) out[(int64_t)c * 7 + k] = acc[k];
    }
}

// point part: one thread per point over its observation list
template <bool SQUARE>
__global__ __launch_bounds__(256) void jt_pt_kernel(const double *__restrict__ Jp,
                                                    const int32_t *__restrict__ pt_ptr,
                                                    const int32_t *__restrict__ pt_obs, int n_pts,
                                                    const double *__restrict__ u,
                                                    double *__restrict__ out /* at the point block */)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pts) return;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int e = pt_ptr[p]; e < pt_ptr[p + 1]; ++e) {
        const int o = pt_obs[e];
        const double *jp = Jp + (int64_t)o * 6;
        if (SQUARE) {
            a0 += jp[0] * jp[0] + jp[3] * jp[3];
            a1 += jp[1] * jp[1] + jp[4] * jp[4];
            a2 += jp[2] * jp[2] + jp[5] * jp[5];
        } else {
            const double2 uu = *reinterpret_cast<const double2 *>(u + 2 * (int64_t)o);
            a0 += jp[0] * uu.x + jp[3] * uu.y;
            a1 += jp[1] * uu.x + jp[4] * uu.y;
            a2 += jp[2] * uu.x + jp[5] * uu.y;
        }
    }
    out[(int64_t)p * 3 + 0] = a0;
    out[(int64_t)p * 3 + 1] = a1;
    out[(int64_t)p * 3 + 2] = a2;
}

constexpr int RED_BLOCKS = 256;

__device__ __forceinline__ double block_sum_256(double v, double *sh)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// calibration part, stage 1: per-block partial sums of the 8 dense columns
template <bool SQUARE>
__global__ __launch_bounds__(256) void jt_cal_kernel(const double *__restrict__ Jk, int64_t n_obs,
                                                     const double *__restrict__ u,
                                                     double *__restrict__ partial /*[RED_BLOCKS][8]*/)
{
    __shared__ double sh[4];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n_obs;
         o += (int64_t)RED_BLOCKS * 256) {
        const double *jk = Jk + o * 16;
        if (SQUARE) {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += jk[k] * jk[k] + jk[8 + k] * jk[8 + k];
        } else {
            const double2 uu = *reinterpret_cast<const double2 *>(u + 2 * o);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += jk[k] * uu.x + jk[8 + k] * uu.y;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double s = block_sum_256(acc[k], sh);
        if (threadIdx.x == 0) partial[blockIdx.x * 8 + k] = s;
    }
}

__global__ __launch_bounds__(256) void final_sum_kernel(const double *__restrict__ partial,
                                                        int n_partial, int width,
                                                        double *__restrict__ out)
{
    // `width` independent columns of n_partial (<= 256) partials each; fixed reduction tree
    __shared__ double sh[4];
    for (int k = 0; k < width; ++k) {
        const double v = (int)threadIdx.x < n_partial ? partial[threadIdx.x * width + k] : 0.0;
        const double s = block_sum_256(v, sh);
        if (threadIdx.x == 0) out[k] = s;
    }
}

// ---------------------------------------------------------------------------------
// generic float64 vector kernels
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void axpby_kernel(int64_t n, double a, const double *__restrict__ x,
                                                    double b, double *__restrict__ y)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = (b == 0.0 ? 0.0 : b * y[i]) + a * x[i];
}

__global__ __launch_bounds__(256) void mul2_kernel(int64_t n, const double *__restrict__ a,
                                                   const double *__restrict__ b,
                                                   const double *__restrict__ c,
                                                   const double *__restrict__ d,
                                                   double *__restrict__ out)
{
    // out = a.*b (+ c.*d when c is not null)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double v = a[i] * b[i];
        if (c) v += c[i] * d[i];
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void dot_kernel(int64_t n, const double *__restrict__ x,
                                                  const double *__restrict__ y,
                                                  double *__restrict__ partial)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)RED_BLOCKS * 256)
        acc += x[i] * y[i];
    const double s = block_sum_256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void lsmr_update_kernel(int64_t n, double *__restrict__ h,
                                                          double *__restrict__ hbar,
                                                          double *__restrict__ x,
                                                          const double *__restrict__ v,
                                                          double c_hbar, double c_x, double c_h)
{
    // scipy lsmr.py: hbar = h + c_hbar*hbar ; x += c_x*hbar ; h = v + c_h*h
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double hb = h[i] + c_hbar * hbar[i];
        hbar[i] = hb;
        x[i] += c_x * hb;
        h[i] = v[i] + c_h * h[i];
    }
}

inline unsigned grid_for(int64_t n)
{
    int64_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int iamx_ba_jv(const double *Jc, const double *Jp, const double *Jk,
                          const int32_t *cam_idx, const int32_t *pt_idx, int64_t n_obs,
                          int n_cams, int n_pts, const double *x, double *y, void *stream)
{
    IAMX_REQUIRE(Jc && Jp && cam_idx && pt_idx && x && y, "null pointer");
    if (n_obs <= 0) return IAMX_OK;
    hipLaunchKernelGGL(jv_kernel, dim3(grid_for(n_obs)), dim3(256), 0, iamx::as_stream(stream), Jc,
                       Jp, Jk, cam_idx, pt_idx, n_obs, n_cams, n_pts, x, y);
    return iamx::check_launch("iamx_ba_jv");
}

extern "C" int iamx_ba_jtv(const double *Jc, const double *Jp, const double *Jk,
                           const int32_t *cam_ptr, const int32_t *pt_ptr, const int32_t *pt_obs,
                           int64_t n_obs, int n_cams, int n_pts, const double *u, int square,
                           double *out, double *scratch, void *stream)
{
    IAMX_REQUIRE(Jc && Jp && cam_ptr && pt_ptr && pt_obs && out, "null pointer");
    IAMX_REQUIRE(square || u, "u is required unless square != 0");
    IAMX_REQUIRE(!Jk || scratch, "scratch (2048 doubles) is required with calibration columns");
    hipStream_t st = iamx::as_stream(stream);
    double *out_p = out + (int64_t)n_cams * 7;
    if (square) {
        hipLaunchKernelGGL(jt_cam_kernel<true>, dim3((n_cams + 3) / 4), dim3(256), 0, st, Jc,
                           cam_ptr, n_cams, u, out);
        hipLaunchKernelGGL(jt_pt_kernel<true>, dim3((n_pts + 255) / 256), dim3(256), 0, st, Jp,
                           pt_ptr, pt_obs, n_pts, u, out_p);
    } else {
        hipLaunchKernelGGL(jt_cam_kernel<false>, dim3((n_cams + 3) / 4), dim3(256), 0, st, Jc,
                           cam_ptr, n_cams, u, out);
        hipLaunchKernelGGL(jt_pt_kernel<false>, dim3((n_pts + 255) / 256), dim3(256), 0, st, Jp,
                           pt_ptr, pt_obs, n_pts, u, out_p);
    }
    if (Jk) {
        if (square)
            hipLaunchKernelGGL(jt_cal_kernel<true>, dim3(RED_BLOCKS), dim3(256), 0, st, Jk, n_obs, u,
                               scratch);
        else
            hipLaunchKernelGGL(jt_cal_kernel<false>, dim3(RED_BLOCKS), dim3(256), 0, st, Jk, n_obs,
                               u, scratch);
        hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, scratch, RED_BLOCKS, 8,
                           out_p + (int64_t)n_pts * 3);
    }
    return iamx::check_launch("iamx_ba_jtv");
}

extern "C" int iamx_vec_axpby(int64_t n, double a, const double *x, double b, double *y,
                              void *stream)
{
    IAMX_REQUIRE(x && y, "null pointer");
    if (n <= 0) return IAMX_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n)), dim3(256), 0, iamx::as_stream(stream), n, a,
                       x, b, y);
    return iamx::check_launch("iamx_vec_axpby");
}

extern "C" int iamx_vec_mul2(int64_t n, const double *a, const double *b, const double *c,
                             const double *d, double *out, void *stream)
{
    IAMX_REQUIRE(a && b && out && (!c || d), "null pointer");
    if (n <= 0) return IAMX_OK;
    hipLaunchKernelGGL(mul2_kernel, dim3(grid_for(n)), dim3(256), 0, iamx::as_stream(stream), n, a,
                       b, c, d, out);
    return iamx::check_launch("iamx_vec_mul2");
}

extern "C" int iamx_vec_dot(int64_t n, const double *x, const double *y, double *out,
                            double *scratch, void *stream)
{
    IAMX_REQUIRE(x && y && out && scratch, "null pointer");
    hipStream_t st = iamx::as_stream(stream);
    hipLaunchKernelGGL(dot_kernel, dim3(RED_BLOCKS), dim3(256), 0, st, n, x, y, scratch);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, scratch, RED_BLOCKS, 1, out);
    return iamx::check_launch("iamx_vec_dot");
}

extern "C" int iamx_vec_lsmr_update(int64_t n, double *h, double *hbar, double *x, const double *v,
                                    double c_hbar, double c_x, double c_h, void *stream)
{
    IAMX_REQUIRE(h && hbar && x && v, "null pointer");
    if (n <= 0) return IAMX_OK;
    hipLaunchKernelGGL(lsmr_update_kernel, dim3(grid_for(n)), dim3(256), 0, iamx::as_stream(stream),
                       n, h, hbar, x, v, c_hbar, c_x, c_h);
    return iamx::check_launch("iamx_vec_lsmr_update");
}
