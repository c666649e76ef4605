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

// =====================================================================================
// Fused, host-free LSMR iterations (scipy/sparse/linalg/_isolve/lsmr.py) on the operator
//     A = [ J diag(d) ; diag(dreg) ],   b = [ r ; 0 ]
//
// * iamx_ba_lsmr_prepare folds the column scaling into J once per solve and lays the blocks
//   out structure-of-arrays so that every load of the iteration kernels is coalesced:
//       Jc_s [14][O] observation order,  Jp_s [6][O] observation order (forward product),
//       Jp_p [6][O]  point-sorted order (adjoint product, one thread per point).
// * The Golub-Kahan vectors are kept UNnormalised (ut = beta*u, vt = alpha*v), so one
//   bidiagonalisation step is a forward and an adjoint kernel whose epilogues emit the
//   squared-norm partials:
//       ut' = (1/alpha) A vt - (alpha/beta) ut          beta'  = |ut'|
//       vt' = (1/beta') A^T ut' - (beta'/alpha) vt      alpha' = |vt'|
// * All scalars of the recurrence live in a double-buffered device state block.  Every
//   workgroup re-derives the scalars it needs in its prologue from the partial sums of the
//   previous kernel (same instruction sequence => bit-identical in all workgroups); workgroup 0
//   writes the next state buffer, which nobody reads in the same kernel.  An iteration is
//   four launches and no host synchronisation; the stopping tests of iteration k run in the
//   prologue of iteration k+1's forward kernel and latch R_ISTOP, which turns everything
//   enqueued behind it into no-ops.
// * Each camera block of J is read ONCE per iteration: the forward kernel runs one workgroup per
//   camera (observations are camera-major), forms ut' for its observations and, with the
//   blocks still in registers, the camera part of J^T ut' (raw, unscaled: beta' is not known
//   yet).  The point part of J^T ut' is a second kernel over the point-sorted copy of Jp.
//       forward+camera adjoint   O*(160 J + 8 idx + 32 ut r/w)
//       point adjoint            O*(48 Jp + 16 ut gather + 4 idx)      + n-vectors
// =====================================================================================
namespace {

// carried recurrences, double-buffered by iteration parity
enum {
    S_ALPHA, S_BETA, S_ZETABAR, S_ALPHABAR, S_RHO, S_RHOBAR, S_CBAR, S_SBAR, S_BETADD, S_BETAD,
    S_RHODOLD, S_TAUTILDEOLD, S_THETATILDE, S_ZETA, S_D, S_NORMA2, S_MAXRBAR, S_MINRBAR, S_ITN,
    S_NORMR, S_NORMAR, S_NORMA, S_CONDA, S_NBUF
};
// constants + latched results, after the two buffers
enum {
    R_ATOL = 2 * S_NBUF, R_BTOL, R_CTOL, R_MAXITER, R_NORMB, R_ISTOP, R_ITN, R_NORMR, R_NORMAR,
    R_NORMA, R_CONDA, R_NORMX, R_COUNT
};

struct LsmrArgs {
    const double *Jc_s, *Jp_s, *Jp_p;
    const int32_t *cam_idx, *pt_idx, *cam_ptr, *pt_ptr, *pt_obs;
    int64_t n_obs;
    int n_cams, n_pts;
    const double *dreg;
    double *u1, *u2, *vt, *h, *hbar, *x;
    double *S;                   // state block [R_COUNT]
    double *partU, *partV, *partX;
    int n_partV;
    double *partU2;
    // xr[0] = |ut1'|^2 (rank local), tbuf = raw J^T ut1' (camera part always; with multi != 0
    // also the point part).  Multi-rank form (observations sharded by point, n-vectors
    // replicated): the caller all-reduces xr[0] and tbuf[0..n) between the phases.
    double *xr, *tbuf;
    int multi;
};


constexpr int LS_U2_BLOCKS = 256;     // fixed grids => fixed reduction trees
constexpr int LS_UPD_BLOCKS = 1024;

struct Givens { double c, s, r; };

__device__ __forceinline__ double sign_of(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); }

// stable plane rotation (scipy/sparse/linalg/_isolve/lsqr.py:_sym_ortho)
__device__ __forceinline__ Givens sym_ortho(double a, double b)
{
    Givens g;
    if (b == 0) { g.c = sign_of(a); g.s = 0; g.r = fabs(a); return g; }
    if (a == 0) { g.c = 0; g.s = sign_of(b); g.r = fabs(b); return g; }
    if (fabs(b) > fabs(a)) {
        const double tau = a / b;
        g.s = sign_of(b) / sqrt(1 + tau * tau);
        g.c = g.s * tau;
        g.r = b / g.s;
    } else {
        const double tau = b / a;
        g.c = sign_of(a) / sqrt(1 + tau * tau);
        g.s = g.c * tau;
        g.r = a / g.c;
    }
    return g;
}

__device__ __forceinline__ double sum_partials(const double *__restrict__ part, int n, double *sh)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
    return block_sum_256(acc, sh);
}

// beta' = |ut'|: the observation part xr[0] (summed by lsmr_sumU_kernel, all-reduced by the
// caller on several ranks) plus the replicated n-vector part
__device__ __forceinline__ double beta_new(const LsmrArgs &A, double *sh)
{
    return sqrt(A.xr[0] + sum_partials(A.partU2, LS_U2_BLOCKS, sh));
}

// one thread per observation: scaled SoA copies in observation order
__global__ __launch_bounds__(256) void lsmr_prepare_obs_kernel(
    const double *__restrict__ Jc, const double *__restrict__ Jp,
    const int32_t *__restrict__ cam_idx, const int32_t *__restrict__ pt_idx, int64_t n_obs,
    int n_cams, const double *__restrict__ d, double *__restrict__ Jc_s, double *__restrict__ Jp_s)
{
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_obs) return;
    const double *dc = d + (int64_t)cam_idx[o] * 7;
    const double *dp = d + (int64_t)n_cams * 7 + (int64_t)pt_idx[o] * 3;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        Jc_s[(int64_t)k * n_obs + o] = Jc[o * 14 + k] * dc[k];
        Jc_s[(int64_t)(7 + k) * n_obs + o] = Jc[o * 14 + 7 + k] * dc[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Jp_s[(int64_t)k * n_obs + o] = Jp[o * 6 + k] * dp[k];
        Jp_s[(int64_t)(3 + k) * n_obs + o] = Jp[o * 6 + 3 + k] * dp[k];
    }
}

// one thread per point-sorted slot e: Jp_p[.][e] = scaled Jp of observation pt_obs[e]
__global__ __launch_bounds__(256) void lsmr_prepare_pt_kernel(
    const double *__restrict__ Jp, const int32_t *__restrict__ pt_idx,
    const int32_t *__restrict__ pt_obs, int64_t n_obs, int n_cams, const double *__restrict__ d,
    double *__restrict__ Jp_p)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_obs) return;
    const int64_t o = pt_obs[e];
    const double *dp = d + (int64_t)n_cams * 7 + (int64_t)pt_idx[o] * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Jp_p[(int64_t)k * n_obs + e] = Jp[o * 6 + k] * dp[k];
        Jp_p[(int64_t)(3 + k) * n_obs + e] = Jp[o * 6 + 3 + k] * dp[k];
    }
}

// ---- kernel A: stopping tests of the previous iteration, then ut' ------------------------
__global__ __launch_bounds__(256) void lsmr_fwd_kernel(LsmrArgs A, int parity)
{
    __shared__ double sh[4];
    double *S = A.S;
    if (S[R_ISTOP] != 0.0) return;
    const double *in = S + parity * S_NBUF;
    const double itn = in[S_ITN];
    if (itn > 0.0) {             // lsmr.py: "Test for convergence" of iteration itn
        const double normx = sqrt(sum_partials(A.partX, LS_UPD_BLOCKS, sh));
        const double normb = S[R_NORMB], normA = in[S_NORMA], normr = in[S_NORMR];
        const double normar = in[S_NORMAR], condA = in[S_CONDA];
        const double test1 = normr / normb;
        const double test2 = (normA * normr) != 0 ? normar / (normA * normr) : INFINITY;
        const double test3 = 1.0 / condA;
        const double t1 = test1 / (1 + normA * normx / normb);
        const double rtol = S[R_BTOL] + S[R_ATOL] * normA * normx / normb;
        double istop = 0;
        if (itn >= S[R_MAXITER]) istop = 7;
        if (1 + test3 <= 1) istop = 6;
        if (1 + test2 <= 1) istop = 5;
        if (1 + t1 <= 1) istop = 4;
        if (test3 <= S[R_CTOL]) istop = 3;
        if (test2 <= S[R_ATOL]) istop = 2;
        if (test1 <= rtol) istop = 1;
        if (!(test1 == test1) || !(normx == normx)) istop = 8;     // breakdown (NaN)
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            S[R_ITN] = itn; S[R_NORMR] = normr; S[R_NORMAR] = normar; S[R_NORMA] = normA;
            S[R_CONDA] = condA; S[R_NORMX] = normx;
            if (istop != 0) S[R_ISTOP] = istop;
        }
        if (istop != 0) return;
    }
    const double ia = 1.0 / in[S_ALPHA], ab = in[S_ALPHA] / in[S_BETA];
    const int64_t O = A.n_obs;
    if ((int)blockIdx.x >= A.n_cams) {       // the replicated n-vector part of ut'
        const int64_t n = (int64_t)A.n_cams * 7 + (int64_t)A.n_pts * 3;
        double acc2 = 0.0;
        for (int64_t i = (int64_t)(blockIdx.x - A.n_cams) * 256 + threadIdx.x; i < n; i += (int64_t)LS_U2_BLOCKS * 256) {
            const double v = ia * A.dreg[i] * A.vt[i] - ab * A.u2[i];
            A.u2[i] = v;
            acc2 += v * v;
        }
        const double s2 = block_sum_256(acc2, sh);
        if (threadIdx.x == 0) A.partU2[blockIdx.x - A.n_cams] = s2;
        return;
    }
    // one camera: its 7 entries of vt are wave-uniform (scalar loads)
    const int c = blockIdx.x;
    const double *vc = A.vt + (int64_t)c * 7;
    const double *xp = A.vt + (int64_t)A.n_cams * 7;
    double acc = 0.0;
    double t7[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int o = A.cam_ptr[c] + threadIdx.x; o < A.cam_ptr[c + 1]; o += 256) {
        const double *vp = xp + (int64_t)A.pt_idx[o] * 3;
        double ju[7], jv[7];
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            ju[k] = A.Jc_s[(int64_t)k * O + o];
            jv[k] = A.Jc_s[(int64_t)(7 + k) * O + o];
            a += ju[k] * vc[k];
            b += jv[k] * vc[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double v = vp[k];
            a += A.Jp_s[(int64_t)k * O + o] * v;
            b += A.Jp_s[(int64_t)(3 + k) * O + o] * v;
        }
        double2 u = *reinterpret_cast<double2 *>(A.u1 + 2 * (int64_t)o);
        u.x = a * ia - ab * u.x;
        u.y = b * ia - ab * u.y;
        *reinterpret_cast<double2 *>(A.u1 + 2 * (int64_t)o) = u;
        acc += u.x * u.x + u.y * u.y;
#pragma unroll
        for (int k = 0; k < 7; ++k) t7[k] += ju[k] * u.x + jv[k] * u.y;      // camera part of J^T ut'
    }
    __shared__ double red[4][8];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) t7[k] += __shfl_xor(t7[k], m);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) red[threadIdx.x >> 6][k] = t7[k];
        red[threadIdx.x >> 6][7] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (threadIdx.x < 7) A.tbuf[(int64_t)c * 7 + threadIdx.x] = v;
        else A.partU[c] = v;
    }
}

// xr[0] = this rank's |ut1'|^2 (all-reduced by the caller on several ranks)
__global__ __launch_bounds__(256) void lsmr_sumU_kernel(LsmrArgs A)
{
    __shared__ double sh[4];
    if (A.S[R_ISTOP] != 0.0) { if (threadIdx.x == 0) A.xr[0] = 0.0; return; }
    const double s = sum_partials(A.partU, A.n_cams, sh);
    if (threadIdx.x == 0) A.xr[0] = s;
}

// ---- kernel B: beta', point part of J^T ut', vt' --------------------------------------------
// Point workgroups: 256 consecutive points each, whose point-sorted slots form one contiguous
// range.  The range is streamed in rounds of ADJ_CH slots: every thread forms the 3 products of
// 4 slots with fully coalesced loads (only the 16-byte ut gather is indirect) into LDS, then
// thread t adds up the slots of point t in ascending order.  The workgroups behind them turn
// the raw camera sums of the forward kernel into vt' (single rank only; on several ranks both
// parts stay raw in tbuf for the all-reduce and lsmr_vt_kernel finishes).
constexpr int ADJ_CH = 1024;

__global__ __launch_bounds__(256) void lsmr_adj_kernel(LsmrArgs A, int parity)
{
    __shared__ double sh[4];
    __shared__ double prod[3][ADJ_CH];
    const bool raw = A.multi != 0;
    if (A.S[R_ISTOP] != 0.0) return;             // (tbuf keeps stale values: nobody reads them)
    const double *in = A.S + parity * S_NBUF;
    double ib = 0.0, ba = 0.0;
    if (!raw) {
        const double bn = beta_new(A, sh);
        if (!(bn > 0)) {         // exact solution reached: v keeps its value (lsmr.py "if beta > 0")
            if (threadIdx.x == 0) A.partV[blockIdx.x] = 0.0;
            return;
        }
        ib = 1.0 / bn;
        ba = bn / in[S_ALPHA];
    }
    const int64_t O = A.n_obs;
    const int n_pt_blocks = (A.n_pts + 255) / 256;
    double sq = 0.0;
    if ((int)blockIdx.x < n_pt_blocks) {
        const int p0 = blockIdx.x * 256;
        const int p1 = min(p0 + 256, A.n_pts);
        const int p = p0 + threadIdx.x;
        const int e0 = A.pt_ptr[p0], e1 = A.pt_ptr[p1];
        const int my_lo = p < p1 ? A.pt_ptr[p] : e1, my_hi = p < p1 ? A.pt_ptr[p + 1] : e1;
        double a0 = 0, a1 = 0, a2 = 0;
        for (int base = e0; base < e1; base += ADJ_CH) {
            const int cnt = min(ADJ_CH, e1 - base);
#pragma unroll
            for (int i = 0; i < ADJ_CH / 256; ++i) {
                const int jx = threadIdx.x + 256 * i;
                if (jx < cnt) {
                    const int64_t e = base + jx;
                    const double2 uu = *reinterpret_cast<const double2 *>(A.u1 + 2 * (int64_t)A.pt_obs[e]);
                    prod[0][jx] = A.Jp_p[e] * uu.x + A.Jp_p[3 * O + e] * uu.y;
                    prod[1][jx] = A.Jp_p[O + e] * uu.x + A.Jp_p[4 * O + e] * uu.y;
                    prod[2][jx] = A.Jp_p[2 * O + e] * uu.x + A.Jp_p[5 * O + e] * uu.y;
                }
            }
            __syncthreads();
            const int lo = max(my_lo, base) - base, hi = min(my_hi, base + cnt) - base;
            for (int jx = lo; jx < hi; ++jx) {
                a0 += prod[0][jx];
                a1 += prod[1][jx];
                a2 += prod[2][jx];
            }
            __syncthreads();
        }
        if (p < p1) {
            const double acc[3] = {a0, a1, a2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int64_t i = (int64_t)A.n_cams * 7 + (int64_t)p * 3 + k;
                if (raw) { A.tbuf[i] = acc[k]; continue; }
                const double v = (acc[k] + A.dreg[i] * A.u2[i]) * ib - ba * A.vt[i];
                A.vt[i] = v;
                sq += v * v;
            }
        }
    } else {                                     // camera entries (launched on a single rank only)
        const int i = ((int)blockIdx.x - n_pt_blocks) * 256 + threadIdx.x;
        if (i < A.n_cams * 7) {
            const double v = (A.tbuf[i] + A.dreg[i] * A.u2[i]) * ib - ba * A.vt[i];
            A.vt[i] = v;
            sq = v * v;
        }
    }
    if (raw) return;
    const double s = block_sum_256(sq, sh);
    if (threadIdx.x == 0) A.partV[blockIdx.x] = s;
}

// multi-rank: vt' from the all-reduced J^T ut1 (tbuf), squared-norm partials
__global__ __launch_bounds__(256) void lsmr_vt_kernel(LsmrArgs A, int parity)
{
    __shared__ double sh[4];
    if (A.S[R_ISTOP] != 0.0) return;
    const double *in = A.S + parity * S_NBUF;
    const double bn = beta_new(A, sh);
    if (!(bn > 0)) {
        if (threadIdx.x == 0) A.partV[blockIdx.x] = 0.0;
        return;
    }
    const double ib = 1.0 / bn, ba = bn / in[S_ALPHA];
    const int64_t n = (int64_t)A.n_cams * 7 + (int64_t)A.n_pts * 3;
    double sq = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)LS_UPD_BLOCKS * 256) {
        const double v = (A.tbuf[i] + A.dreg[i] * A.u2[i]) * ib - ba * A.vt[i];
        A.vt[i] = v;
        sq += v * v;
    }
    const double s = block_sum_256(sq, sh);
    if (threadIdx.x == 0) A.partV[blockIdx.x] = s;
}

// ---- kernel C: alpha', plane rotations (lsmr.py main loop), then h / hbar / x -------------
__global__ __launch_bounds__(256) void lsmr_update3_kernel(LsmrArgs A, int parity)
{
    __shared__ double sh[4];
    double *S = A.S;
    if (S[R_ISTOP] != 0.0) return;
    const double *in = S + parity * S_NBUF;
    double *out = S + (1 - parity) * S_NBUF;
    const double beta = beta_new(A, sh);
    const double s2 = sum_partials(A.partV, A.n_partV, sh);
    const double alpha = beta > 0 ? sqrt(s2) : in[S_ALPHA];

    const Givens g1 = sym_ortho(in[S_ALPHABAR], 0.0);
    const double chat = g1.c, shat = g1.s, alphahat = g1.r;
    const double rhoold = in[S_RHO];
    const Givens g2 = sym_ortho(alphahat, beta);
    const double c = g2.c, s = g2.s, rho = g2.r;
    const double thetanew = s * alpha;
    const double alphabar = c * alpha;
    const double rhobarold = in[S_RHOBAR], zetaold = in[S_ZETA];
    const double thetabar = in[S_SBAR] * rho;
    const double rhotemp = in[S_CBAR] * rho;
    const Givens g3 = sym_ortho(in[S_CBAR] * rho, thetanew);
    const double cbar = g3.c, sbar = g3.s, rhobar = g3.r;
    const double zeta = cbar * in[S_ZETABAR];
    const double zetabar = -sbar * in[S_ZETABAR];
    const double chb = -(thetabar * rho / (rhoold * rhobarold));
    const double cx = zeta / (rho * rhobar);
    const double ch = -(thetanew / rho);

    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // estimate of ||r||, ||A||, cond(A)
        const double betaacute = chat * in[S_BETADD], betacheck = -shat * in[S_BETADD];
        const double betahat = c * betaacute;
        const double betadd = -s * betaacute;
        const double thetatildeold = in[S_THETATILDE];
        const Givens g4 = sym_ortho(in[S_RHODOLD], thetabar);
        const double ct = g4.c, st = g4.s, rhotildeold = g4.r;
        const double thetatilde = st * rhobar;
        const double rhodold = ct * rhobar;
        const double betad = -st * in[S_BETAD] + ct * betahat;
        const double tautildeold = (zetaold - thetatildeold * in[S_TAUTILDEOLD]) / rhotildeold;
        const double taud = (zeta - thetatilde * tautildeold) / rhodold;
        const double dsum = in[S_D] + betacheck * betacheck;
        const double itn = in[S_ITN] + 1.0;
        double normA2 = in[S_NORMA2] + beta * beta;
        out[S_NORMA] = sqrt(normA2);
        normA2 += alpha * alpha;
        const double maxrbar = fmax(in[S_MAXRBAR], rhobarold);
        const double minrbar = itn > 1.0 ? fmin(in[S_MINRBAR], rhobarold) : in[S_MINRBAR];
        out[S_ALPHA] = alpha; out[S_BETA] = beta; out[S_ZETABAR] = zetabar;
        out[S_ALPHABAR] = alphabar; out[S_RHO] = rho; out[S_RHOBAR] = rhobar; out[S_CBAR] = cbar;
        out[S_SBAR] = sbar; out[S_BETADD] = betadd; out[S_BETAD] = betad; out[S_RHODOLD] = rhodold;
        out[S_TAUTILDEOLD] = tautildeold; out[S_THETATILDE] = thetatilde; out[S_ZETA] = zeta;
        out[S_D] = dsum; out[S_NORMA2] = normA2; out[S_MAXRBAR] = maxrbar; out[S_MINRBAR] = minrbar;
        out[S_ITN] = itn;
        out[S_NORMR] = sqrt(dsum + (betad - taud) * (betad - taud) + betadd * betadd);
        out[S_NORMAR] = fabs(zetabar);
        out[S_CONDA] = fmax(maxrbar, rhotemp) / fmin(minrbar, rhotemp);
    }

    const double ia = alpha > 0 ? 1.0 / alpha : 0.0;
    const int64_t n = (int64_t)A.n_cams * 7 + (int64_t)A.n_pts * 3;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)LS_UPD_BLOCKS * 256) {
        const double hb = A.h[i] + chb * A.hbar[i];
        A.hbar[i] = hb;
        const double xv = A.x[i] + cx * hb;
        A.x[i] = xv;
        A.h[i] = A.vt[i] * ia + ch * A.h[i];
        acc += xv * xv;
    }
    const double t = block_sum_256(acc, sh);
    if (threadIdx.x == 0) A.partX[blockIdx.x] = t;
}

}  // namespace

extern "C" int iamx_ba_lsmr_state_size(void) { return R_COUNT; }

namespace {

struct PartLayout {
    int n_adj;                    // single-rank adjoint workgroups (points + camera entries)
    int64_t off_U2, off_X, off_V, total;
};

PartLayout part_layout(int n_cams, int n_pts)
{
    PartLayout L;
    L.n_adj = (n_pts + 255) / 256 + (n_cams * 7 + 255) / 256;
    L.off_U2 = n_cams;
    L.off_X = L.off_U2 + LS_U2_BLOCKS;
    L.off_V = L.off_X + LS_UPD_BLOCKS;
    L.total = L.off_V + (L.n_adj > LS_UPD_BLOCKS ? L.n_adj : LS_UPD_BLOCKS);
    return L;
}

}  // namespace

extern "C" int64_t iamx_ba_lsmr_partials_size(int n_cams, int n_pts)
{
    return part_layout(n_cams, n_pts).total;
}

extern "C" int iamx_ba_lsmr_prepare(const double *Jc, const double *Jp, const int32_t *cam_idx,
                                    const int32_t *pt_idx, const int32_t *pt_obs, int64_t n_obs,
                                    int n_cams, int n_pts, const double *d, double *Jc_s,
                                    double *Jp_s, double *Jp_p, void *stream)
{
    IAMX_REQUIRE(Jc && Jp && cam_idx && pt_idx && pt_obs && d && Jc_s && Jp_s && Jp_p, "null pointer");
    IAMX_REQUIRE(n_obs > 0 && n_cams > 0 && n_pts > 0, "bad size");
    hipStream_t st = iamx::as_stream(stream);
    const unsigned g = (unsigned)((n_obs + 255) / 256);
    hipLaunchKernelGGL(lsmr_prepare_obs_kernel, dim3(g), dim3(256), 0, st, Jc, Jp, cam_idx, pt_idx,
                       n_obs, n_cams, d, Jc_s, Jp_s);
    hipLaunchKernelGGL(lsmr_prepare_pt_kernel, dim3(g), dim3(256), 0, st, Jp, pt_idx, pt_obs, n_obs,
                       n_cams, d, Jp_p);
    return iamx::check_launch("iamx_ba_lsmr_prepare");
}

extern "C" int iamx_ba_lsmr_iterate(const double *Jc_s, const double *Jp_s, const double *Jp_p,
                                    const int32_t *cam_idx, const int32_t *pt_idx,
                                    const int32_t *cam_ptr, const int32_t *pt_ptr,
                                    const int32_t *pt_obs, int64_t n_obs, int n_cams, int n_pts,
                                    const double *dreg, double *u1, double *u2, double *vt,
                                    double *h, double *hbar, double *x, double *state,
                                    double *partials, double *xr, double *tbuf, int n_iter,
                                    void *stream)
{
    IAMX_REQUIRE(Jc_s && Jp_s && Jp_p && cam_idx && pt_idx && cam_ptr && pt_ptr && pt_obs && dreg &&
                     u1 && u2 && vt && h && hbar && x && state && partials && xr && tbuf,
                 "null pointer");
    IAMX_REQUIRE(n_obs > 0 && n_cams > 0 && n_pts > 0 && n_iter >= 0 && (n_iter & 1) == 0,
                 "bad size (n_iter must be even: the state block is double-buffered)");
    const PartLayout L = part_layout(n_cams, n_pts);
    LsmrArgs A{Jc_s, Jp_s, Jp_p, cam_idx, pt_idx, cam_ptr, pt_ptr, pt_obs, n_obs, n_cams, n_pts,
               dreg, u1, u2, vt, h, hbar, x, state,
               partials, partials + L.off_V, partials + L.off_X, L.n_adj, partials + L.off_U2,
               xr, tbuf, 0};
    hipStream_t st = iamx::as_stream(stream);
    for (int it = 0; it < n_iter; ++it) {
        const int parity = it & 1;
        hipLaunchKernelGGL(lsmr_fwd_kernel, dim3(n_cams + LS_U2_BLOCKS), dim3(256), 0, st, A, parity);
        hipLaunchKernelGGL(lsmr_sumU_kernel, dim3(1), dim3(256), 0, st, A);
        hipLaunchKernelGGL(lsmr_adj_kernel, dim3(L.n_adj), dim3(256), 0, st, A, parity);
        hipLaunchKernelGGL(lsmr_update3_kernel, dim3(LS_UPD_BLOCKS), dim3(256), 0, st, A, parity);
    }
    return iamx::check_launch("iamx_ba_lsmr_iterate");
}

// One phase of one iteration of the multi-rank form (the caller all-reduces xr[0] after phase 0
// and tbuf[0..n) after phase 1, on the same stream):
//   phase 0: stopping tests of the previous iteration, ut', raw camera part of J^T ut1' -> tbuf,
//            xr[0] = local |ut1'|^2
//   phase 1: raw point part of J^T ut1' -> tbuf
//   phase 2: vt' from the reduced tbuf, alpha', plane rotations, h / hbar / x
extern "C" int iamx_ba_lsmr_phase(const double *Jc_s, const double *Jp_s, const double *Jp_p,
                                  const int32_t *cam_idx, const int32_t *pt_idx,
                                  const int32_t *cam_ptr, const int32_t *pt_ptr,
                                  const int32_t *pt_obs, int64_t n_obs, int n_cams, int n_pts,
                                  const double *dreg, double *u1, double *u2, double *vt, double *h,
                                  double *hbar, double *x, double *state, double *partials,
                                  double *xr, double *tbuf, int phase, int parity, void *stream)
{
    IAMX_REQUIRE(Jc_s && Jp_s && Jp_p && cam_idx && pt_idx && cam_ptr && pt_ptr && pt_obs && dreg &&
                     u1 && u2 && vt && h && hbar && x && state && partials && xr && tbuf,
                 "null pointer");
    IAMX_REQUIRE(n_obs >= 0 && n_cams > 0 && n_pts > 0 && phase >= 0 && phase <= 2 &&
                     (parity == 0 || parity == 1),
                 "bad size / phase / parity");
    const PartLayout L = part_layout(n_cams, n_pts);
    LsmrArgs A{Jc_s, Jp_s, Jp_p, cam_idx, pt_idx, cam_ptr, pt_ptr, pt_obs, n_obs, n_cams, n_pts,
               dreg, u1, u2, vt, h, hbar, x, state,
               partials, partials + L.off_V, partials + L.off_X, LS_UPD_BLOCKS, partials + L.off_U2,
               xr, tbuf, 1};
    hipStream_t st = iamx::as_stream(stream);
    if (phase == 0) {
        hipLaunchKernelGGL(lsmr_fwd_kernel, dim3(n_cams + LS_U2_BLOCKS), dim3(256), 0, st, A, parity);
        hipLaunchKernelGGL(lsmr_sumU_kernel, dim3(1), dim3(256), 0, st, A);
    } else if (phase == 1) {
        hipLaunchKernelGGL(lsmr_adj_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, st, A, parity);
    } else {
        hipLaunchKernelGGL(lsmr_vt_kernel, dim3(LS_UPD_BLOCKS), dim3(256), 0, st, A, parity);
        hipLaunchKernelGGL(lsmr_update3_kernel, dim3(LS_UPD_BLOCKS), dim3(256), 0, st, A, parity);
    }
    return iamx::check_launch("iamx_ba_lsmr_phase");
}
