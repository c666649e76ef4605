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
// * Matrix free: the iteration kernels re-derive the 2x10 block of an observation from the
//   camera and the point instead of reading it (the analytic Jacobian of ba_kernels.hip,
//   factored so that the per-observation work is ~200 flops).  A materialised, scaled J is
//   408 MB per pass at BASELINE config 4; the matrix-free working set (ut 31 MB, the point /
//   camera tables 13 MB, the n-vectors) stays in the 256 MB Infinity Cache.
//   iamx_ba_lsmr_prepare builds the tables once per solve:
//       ctab [C][32]: B^T = M(q)^T/|q|^2 (9), ned (3), q (4), 1/|q|^2, d of the 7 columns
//       ptab [P][6] : X (3), d of the 3 columns
//   With  w = du*ut_x + dv*ut_y  (du, dv = d(u,v)/d(body point)):
//       J_point^T ut = -B w,   J_ned^T ut = +B w,   J_q[k]^T ut = -w.e_k,
//       e_k = 2 (Q_k dX - q_k yb)/|q|^2,  Q_k the bilinear forms of ba_residual_jac_kernel
//   and J v = du.g, dv.g with g = B^T (v_ned - v_point) - sum_k v_q[k] e_k.
// * The Golub-Kahan vectors are kept UNnormalised (ut = beta*u, vt = alpha*v), so one
//   bidiagonalisation step is a forward and an adjoint kernel whose epilogues emit the
//   squared-norm partials:
//       ut' = (1/alpha) A vt - (alpha/beta) ut          beta'  = |ut'|
//       vt' = (1/beta') A^T ut' - (beta'/alpha) vt      alpha' = |vt'|
// * All scalars of the recurrence live in a double-buffered device state block.  Every
//   workgroup re-derives the scalars it needs in its prologue from the partial sums of the
//   previous kernel (same instruction sequence => bit-identical in all workgroups); workgroup 0
//   writes the next state buffer, which nobody reads in the same kernel.  An iteration is
//   three launches (forward, adjoint, update) and no host synchronisation; the stopping tests
//   of iteration k run in the prologue of iteration k+1's adjoint kernel (several ranks: in
//   the one-workgroup kernel in front of the all-reduce) and latch R_ISTOP, which turns
//   everything enqueued behind it into no-ops.
// * The forward kernel runs one workgroup per camera (observations are camera-major, the
//   camera's table row comes through the scalar cache), forms ut' for its observations and,
//   with the geometry still in registers, the camera part of J^T ut' (raw: beta' is not known
//   yet; accumulated as sum w and sum w (x) dX, 12 numbers, and expanded once per camera).
//   The point part of J^T ut' is a second kernel over the point-sorted slots.
//       forward+camera adjoint   O*(4 idx + 32 ut r/w) + cached point rows and vt gathers
//       point adjoint            O*(12 idx + 16 ut gather) + cached point / camera rows
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

constexpr int CT = 32;            // doubles per camera table row
enum { CT_BT = 0, CT_NED = 9, CT_Q = 12, CT_INVN = 16, CT_D = 17 };

struct LsmrArgs {
    const double *ctab, *ptab, *calib;
    const int32_t *pt_idx, *cam_ptr, *pt_ptr, *pt_obs;
    const int2 *slot_cp;         // per point-sorted slot: (camera, point) of its observation
    int64_t n_obs;
    int n_cams, n_pts;
    const double *dreg;
    double *u1, *u2, *vt, *h, *hbar, *x;
    double *S;                   // state block [R_COUNT]
    double *partU, *partV, *partX;
    int n_partV;
    double *partU2;
    // Sums that cross kernels: xr[0] + xr[2] = |ut'|^2, xr[1] + xr[3] = |x|^2 (several ranks
    // only).  tbuf[0 .. 7C) = raw camera part of J^T ut1'.
    // Multi-rank form: observations AND the point part of every n-vector are sharded by point
    // (this rank owns points [pt_lo, pt_hi)); only the camera part (7C entries) is replicated.
    // xr[0] / xr[1] hold the rank-local parts (observations + owned points), xr[2] / xr[3] the
    // replicated camera parts; tbuf[7C] carries the rank's sum of squares of the point part of
    // vt'.  The caller all-reduces xr[0..2) after phase 0 and tbuf[0 .. 7C] after phase 1.
    double *xr, *tbuf;
    int multi;
    int pt_lo, pt_hi;
    // eprod [n_obs][3]: the point part of J^T ut' per observation (camera-major order), written by
    // the forward kernel -- which has the observation's geometry and the new ut' in registers
    // anyway -- and summed per point by the adjoint kernel
    double *eprod;
};


constexpr int LS_U2_BLOCKS = 256;     // fixed grids => fixed reduction trees
constexpr int LS_UPD_BLOCKS = 1024;
// the first blocks of the n-vector grids take the (replicated) camera entries, the others the
// point entries of the points this rank owns: the two kinds of partial sums stay apart
constexpr int LS_U2_CAM = 16;
constexpr int LS_UPD_CAM = 32;

// n-vector entries of block b (of nb, the first ncam on camera entries): i = first; i < end; i += step
struct NvecRange { int64_t first, end, step; };
__device__ __forceinline__ NvecRange nvec_range(const LsmrArgs &A, int b, int nb, int ncam)
{
    NvecRange r;
    const int64_t ncam_e = (int64_t)A.n_cams * 7;
    if (b < ncam) {
        r.first = (int64_t)b * 256 + threadIdx.x; r.end = ncam_e; r.step = (int64_t)ncam * 256;
    } else {
        r.first = ncam_e + 3 * (int64_t)A.pt_lo + (int64_t)(b - ncam) * 256 + threadIdx.x;
        r.end = ncam_e + 3 * (int64_t)A.pt_hi;
        r.step = (int64_t)(nb - ncam) * 256;
    }
    return r;
}

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
    (void)sh;
    return sqrt(A.xr[0] + A.xr[2]);
}

// a wave-uniform double that the compiler computed with vector instructions: move it to scalar
// registers (two v_readfirstlane) so that it stops occupying a VGPR pair in every lane
__device__ __forceinline__ double uniform(double x)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

// d(u,v)/d(body point) of one observation from its camera table row and point -- the
// arithmetic of ba_residual_jac_kernel (ba_kernels.hip) up to `du`, `dv`
struct ObsGeom { double du[3], dv[3], dX[3], yb[3]; };

__device__ __forceinline__ void obs_geom(const double *__restrict__ ct, const double *__restrict__ X,
                                         const double (&cal)[9], ObsGeom &G)
{
    const double a = X[0] - ct[CT_NED], b = X[1] - ct[CT_NED + 1], c = X[2] - ct[CT_NED + 2];
    G.dX[0] = a; G.dX[1] = b; G.dX[2] = c;
    const double y0 = ct[0] * a + ct[1] * b + ct[2] * c;
    const double y1 = ct[3] * a + ct[4] * b + ct[5] * c;
    const double y2 = ct[6] * a + ct[7] * b + ct[8] * c;
    G.yb[0] = y0; G.yb[1] = y1; G.yb[2] = y2;
    const double iz = 1.0 / y0;            // camera z = body x
    const double x = y1 * iz, y = y2 * iz;
    const double r2 = x * x + y * y;
    const double k1 = cal[4], k2 = cal[5], p1 = cal[6], p2 = cal[7], k3 = cal[8];
    const double rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
    const double drad = k1 + r2 * (2.0 * k2 + 3.0 * k3 * r2);
    const double xdx = rad + 2.0 * x * x * drad + 2.0 * p1 * y + 6.0 * p2 * x;
    const double xdy = 2.0 * x * y * drad + 2.0 * p1 * x + 2.0 * p2 * y;
    const double ydy = rad + 2.0 * y * y * drad + 6.0 * p1 * y + 2.0 * p2 * x;
    const double ux = cal[0] * xdx, uy = cal[0] * xdy, vx = cal[1] * xdy, vy = cal[1] * ydy;
    G.du[0] = -(ux * x + uy * y) * iz;   G.dv[0] = -(vx * x + vy * y) * iz;
    G.du[1] = ux * iz;                   G.dv[1] = vx * iz;
    G.du[2] = uy * iz;                   G.dv[2] = vy * iz;
}

// one thread per camera / per point: the tables of a solve
__global__ __launch_bounds__(256) void lsmr_prepare_mf_kernel(
    const double *__restrict__ cams, const double *__restrict__ pts, const double *__restrict__ d,
    int n_cams, int n_pts, double *__restrict__ ctab, double *__restrict__ ptab)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_cams) {
        const double *cam = cams + (int64_t)i * 7;
        double *o = ctab + (int64_t)i * CT;
        const double w = cam[3], x = cam[4], y = cam[5], z = cam[6];
        const double n = w * w + x * x + y * y + z * z;
        if (n < 2.220446049250313e-16 * 4.0) {       // transformations._EPS: identity, no q columns
            o[0] = 1; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 1; o[5] = 0; o[6] = 0; o[7] = 0; o[8] = 1;
            o[CT_INVN] = 0.0;
        } else {
            const double inv_n = 1.0 / n;
            const double ww = w * w, xx = x * x, yy = y * y, zz = z * z;
            const double xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
            o[0] = (ww + xx - yy - zz) * inv_n; o[1] = 2.0 * (xy + wz) * inv_n; o[2] = 2.0 * (xz - wy) * inv_n;
            o[3] = 2.0 * (xy - wz) * inv_n; o[4] = (ww - xx + yy - zz) * inv_n; o[5] = 2.0 * (yz + wx) * inv_n;
            o[6] = 2.0 * (xz + wy) * inv_n; o[7] = 2.0 * (yz - wx) * inv_n; o[8] = (ww - xx - yy + zz) * inv_n;
            o[CT_INVN] = inv_n;
        }
        o[CT_NED] = cam[0]; o[CT_NED + 1] = cam[1]; o[CT_NED + 2] = cam[2];
        o[CT_Q] = w; o[CT_Q + 1] = x; o[CT_Q + 2] = y; o[CT_Q + 3] = z;
#pragma unroll
        for (int k = 0; k < 7; ++k) o[CT_D + k] = d[(int64_t)i * 7 + k];
#pragma unroll
        for (int k = CT_D + 7; k < CT; ++k) o[k] = 0.0;
    }
    const int p = i - n_cams;
    if (p >= 0 && p < n_pts) {
        double *o = ptab + (int64_t)p * 6;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = pts[(int64_t)p * 3 + k];
            o[3 + k] = d[(int64_t)n_cams * 7 + (int64_t)p * 3 + k];
        }
    }
}

// ---- kernel A: ut' ------------------------------------------------------------------------
// (the stopping tests of the previous iteration run behind this kernel -- adjoint prologue on a
// single rank, lsmr_sumU_kernel on several: if they latch a stop, this launch has only touched
// ut / tbuf, which nobody reads any more, and x is final)
// Three workgroups per CU on purpose (amdgpu_waves_per_eu): every observation gathers its
// point's 72 bytes, and with four resident workgroups per CU those gathers evict each other --
// measured 70 us at four (the register count would allow it), 57 at three, 66 at two.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void lsmr_fwd_kernel(LsmrArgs A, int parity)
{
    __shared__ double sh[4];
    double *S = A.S;
    if (S[R_ISTOP] != 0.0) return;
    const double *in = S + parity * S_NBUF;
    const double ia = uniform(1.0 / in[S_ALPHA]), ab = uniform(in[S_ALPHA] / in[S_BETA]);
    if ((int)blockIdx.x >= A.n_cams) {       // the n-vector part of ut' (camera + owned points)
        const NvecRange R = nvec_range(A, blockIdx.x - A.n_cams, LS_U2_BLOCKS, LS_U2_CAM);
        double acc2 = 0.0;
        for (int64_t i = R.first; i < R.end; i += R.step) {
            const double v = ia * A.dreg[i] * A.vt[i] - ab * A.u2[i];
            A.u2[i] = v;
            acc2 += v * v;
        }
        const double s2 = block_sum_256(acc2, sh);
        if (threadIdx.x == 0) A.partU2[blockIdx.x - A.n_cams] = s2;
        return;
    }
    // one camera: its table row and its 7 entries of vt are wave-uniform (scalar loads)
    // workgroups are dealt to the 8 XCDs round robin: give every XCD a CONTIGUOUS run of cameras
    // (neighbouring cameras of a survey see the same points, and each XCD has its own L2)
    int c;
    {
        const int bid = blockIdx.x, xcd = bid & 7, k = bid >> 3, q = A.n_cams >> 3, r = A.n_cams & 7;
        c = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const double *ct = A.ctab + (int64_t)c * CT;
    const double *vc = A.vt + (int64_t)c * 7;
    const double *xp = A.vt + (int64_t)A.n_cams * 7;
    double cal[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cal[i] = A.calib[i];
    const double qw = ct[CT_Q], qx = ct[CT_Q + 1], qy = ct[CT_Q + 2], qz = ct[CT_Q + 3];
    const double inv_n2 = uniform(2.0 * ct[CT_INVN]);
    // rows of the bilinear forms s, t, p, d of (q, dX) (ba_residual_jac_kernel)
    const double Sr[3] = {qw, qz, -qy}, Tr[3] = {-qz, qw, qx}, Pr[3] = {qy, -qx, qw}, Dr[3] = {qx, qy, qz};
    const double wn0 = uniform(ct[CT_D] * vc[0]), wn1 = uniform(ct[CT_D + 1] * vc[1]);
    const double wn2 = uniform(ct[CT_D + 2] * vc[2]);
    const double wq0 = ct[CT_D + 3] * vc[3], wq1 = ct[CT_D + 4] * vc[4];
    const double wq2 = ct[CT_D + 5] * vc[5], wq3 = ct[CT_D + 6] * vc[6];
    // sum_k wq[k] e_k = inv_n2 * (N dX - qs yb); N, qs are the same in every lane: keep them
    // in scalar registers (142 -> fewer VGPRs = one more wave per SIMD)
    double N[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        N[0][j] = uniform(wq0 * Sr[j] + wq1 * Dr[j] - wq2 * Pr[j] + wq3 * Tr[j]);
        N[1][j] = uniform(wq0 * Tr[j] + wq1 * Pr[j] + wq2 * Dr[j] - wq3 * Sr[j]);
        N[2][j] = uniform(wq0 * Pr[j] - wq1 * Tr[j] + wq2 * Sr[j] + wq3 * Dr[j]);
    }
    const double qs = uniform(wq0 * qw + wq1 * qx + wq2 * qy + wq3 * qz);
    double acc = 0.0;
    double sw[3] = {0, 0, 0};
    double Mo[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    // K observations per thread and pass, their loads issued before the arithmetic.  K = 3
    // (a whole camera in one pass) was measured SLOWER than K = 1 (74 vs 61 us): 206 instead of
    // 142 VGPRs drop the occupancy to 2 waves per SIMD, which costs more than the overlap gains.
    constexpr int K = 1;
    const int o_end = A.cam_ptr[c + 1];
    for (int base = A.cam_ptr[c] + threadIdx.x; base < o_end; base += 256 * K) {
        int pi[K];
        double2 uo[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int o = base + 256 * k;
            pi[k] = o < o_end ? A.pt_idx[o] : -1;
            uo[k] = o < o_end ? *reinterpret_cast<const double2 *>(A.u1 + 2 * (int64_t)o) : make_double2(0, 0);
        }
        double pr[K][6], vp[K][3];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (pi[k] < 0) continue;
            const double2 *src = reinterpret_cast<const double2 *>(A.ptab + (int64_t)pi[k] * 6);
            const double2 p0 = src[0], p1 = src[1], p2 = src[2];
            pr[k][0] = p0.x; pr[k][1] = p0.y; pr[k][2] = p1.x; pr[k][3] = p1.y; pr[k][4] = p2.x; pr[k][5] = p2.y;
            const double *v = xp + (int64_t)pi[k] * 3;
            vp[k][0] = v[0]; vp[k][1] = v[1]; vp[k][2] = v[2];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (pi[k] < 0) continue;
            const int o = base + 256 * k;
            ObsGeom G;
            obs_geom(ct, pr[k], cal, G);
            // z = (d v)_ned - (d v)_point  (J_ned = +B, J_point = -B in the body frame)
            const double z0 = wn0 - pr[k][3] * vp[k][0], z1 = wn1 - pr[k][4] * vp[k][1];
            const double z2 = wn2 - pr[k][5] * vp[k][2];
            double g[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double bz = ct[3 * i] * z0 + ct[3 * i + 1] * z1 + ct[3 * i + 2] * z2;
                const double nd = N[i][0] * G.dX[0] + N[i][1] * G.dX[1] + N[i][2] * G.dX[2];
                g[i] = bz - inv_n2 * (nd - qs * G.yb[i]);
            }
            const double a = G.du[0] * g[0] + G.du[1] * g[1] + G.du[2] * g[2];
            const double b = G.dv[0] * g[0] + G.dv[1] * g[1] + G.dv[2] * g[2];
            double2 u = uo[k];
            u.x = a * ia - ab * u.x;
            u.y = b * ia - ab * u.y;
            *reinterpret_cast<double2 *>(A.u1 + 2 * (int64_t)o) = u;
            acc += u.x * u.x + u.y * u.y;
            double e0 = 0.0, e1 = 0.0, e2 = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double w = G.du[i] * u.x + G.dv[i] * u.y;
                sw[i] += w;
#pragma unroll
                for (int j = 0; j < 3; ++j) Mo[i][j] += w * G.dX[j];
                e0 += ct[3 * i] * w;
                e1 += ct[3 * i + 1] * w;
                e2 += ct[3 * i + 2] * w;
            }
            // point part of J^T ut' of this observation (J_point = -B in the body frame): the
            // adjoint kernel only gathers and sums these (a 24-byte record per observation: one
            // gather per slot there; three planes cost it 8 us more)
            double *ep = A.eprod + 3 * (int64_t)o;
            ep[0] = -e0; ep[1] = -e1; ep[2] = -e2;
        }
    }
    __shared__ double red[4][13];
    double vals[13];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        vals[i] = sw[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) vals[3 + 3 * i + j] = Mo[i][j];
    }
    vals[12] = acc;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) vals[k] += __shfl_xor(vals[k], m);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 13; ++k) red[threadIdx.x >> 6][k] = vals[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[13];
#pragma unroll
        for (int k = 0; k < 13; ++k) t[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        // camera part of J^T ut' from sum w (t[0..2]) and sum w (x) dX (t[3..11])
        double out7[7];
#pragma unroll
        for (int j = 0; j < 3; ++j) out7[j] = ct[j] * t[0] + ct[3 + j] * t[1] + ct[6 + j] * t[2];
        // w.yb = sum_ij B^T[i][j] Mo[i][j];  w.(Q_k dX) = sum_ij Q_k[i][j] Mo[i][j]
        double wy = 0.0, f[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double m0 = t[3 + j], m1 = t[6 + j], m2 = t[9 + j];
            wy += ct[j] * m0 + ct[3 + j] * m1 + ct[6 + j] * m2;
            f[0] += Sr[j] * m0 + Tr[j] * m1 + Pr[j] * m2;
            f[1] += Dr[j] * m0 + Pr[j] * m1 - Tr[j] * m2;
            f[2] += -Pr[j] * m0 + Dr[j] * m1 + Sr[j] * m2;
            f[3] += Tr[j] * m0 - Sr[j] * m1 + Dr[j] * m2;
        }
        const double qq[4] = {qw, qx, qy, qz};
#pragma unroll
        for (int k = 0; k < 4; ++k) out7[3 + k] = -inv_n2 * (f[k] - qq[k] * wy);
#pragma unroll
        for (int k = 0; k < 7; ++k) A.tbuf[(int64_t)c * 7 + k] = out7[k] * ct[CT_D + k];
        A.partU[c] = t[12];
    }
}

// Several ranks only -- one workgroup between the forward kernel and the all-reduce of xr[0..2):
//   xr[0] = this rank's part of |ut'|^2 (its observations + the owned point entries of ut2'),
//   xr[1] = its part of |x|^2 (owned point entries),  xr[2] / xr[3] = the camera parts of the two
//   (replicated: every rank computes the same numbers, they are NOT summed over ranks).
// The stopping tests of the previous iteration need the reduced |x|^2 and run in the prologue of
// the adjoint kernel, like on a single rank.
// A separate launch on purpose: letting the last forward workgroup do it ("threadfence
// reduction") needs device-scope fences, and on this 8-XCD part every such fence writes the
// XCD's L2 back -- measured 48 -> 297 us for the forward kernel.
__global__ __launch_bounds__(256) void lsmr_sumU_kernel(LsmrArgs A, int parity)
{
    (void)parity;
    if (A.S[R_ISTOP] != 0.0) return;
    double v[5] = {0, 0, 0, 0, 0};                 // U obs, U2 cam, U2 pts, X cam, X pts
    for (int i = threadIdx.x; i < A.n_cams; i += 256) v[0] += A.partU[i];
    for (int i = threadIdx.x; i < LS_U2_BLOCKS; i += 256) v[i < LS_U2_CAM ? 1 : 2] += A.partU2[i];
    for (int i = threadIdx.x; i < LS_UPD_BLOCKS; i += 256) v[i < LS_UPD_CAM ? 3 : 4] += A.partX[i];
    __shared__ double sh5[4][5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v[k] += __shfl_xor(v[k], m);
        if ((threadIdx.x & 63) == 0) sh5[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) t[k] = sh5[0][k] + sh5[1][k] + sh5[2][k] + sh5[3][k];
        A.xr[0] = t[0] + t[2];
        A.xr[1] = t[4];
        A.xr[2] = t[1];
        A.xr[3] = t[3];
    }
}

// Several ranks only -- behind the adjoint kernel: this rank's sum of squares of the point part
// of vt' goes to tbuf[7C], right behind the raw camera part, so that ONE all-reduce of
// tbuf[0 .. 7C] delivers both
__global__ __launch_bounds__(256) void lsmr_sumV_kernel(LsmrArgs A)
{
    __shared__ double sh[4];
    if (A.S[R_ISTOP] != 0.0) return;
    const int npb = (A.pt_hi - A.pt_lo + 255) / 256;
    const double s = sum_partials(A.partV, npb, sh);
    if (threadIdx.x == 0) A.tbuf[(int64_t)A.n_cams * 7] = s;
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
    if (A.S[R_ISTOP] != 0.0) return;             // (tbuf keeps stale values: nobody reads them)
    const double *in = A.S + parity * S_NBUF;
    double ib = 0.0, ba = 0.0;
    {
        // No launch between the forward kernel (+ the all-reduce on several ranks) and this one:
        // every workgroup derives |x|^2 and |ut'|^2 itself -- from the partial sums on a single
        // rank, from the reduced xr on several (same code, same order => bit-identical decisions
        // everywhere) --, runs the stopping tests of the previous iteration and derives beta';
        // workgroup 0 records them.
        double *S = A.S;
        double sumX, sUU;
        if (A.multi) {
            sumX = A.xr[1] + A.xr[3];
            sUU = A.xr[0] + A.xr[2];
        } else {
            double aX = 0.0, aU = 0.0, aU2 = 0.0;
            for (int i = threadIdx.x; i < LS_UPD_BLOCKS; i += 256) aX += A.partX[i];
            for (int i = threadIdx.x; i < A.n_cams; i += 256) aU += A.partU[i];
            for (int i = threadIdx.x; i < LS_U2_BLOCKS; i += 256) aU2 += A.partU2[i];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                aX += __shfl_xor(aX, m);
                aU += __shfl_xor(aU, m);
                aU2 += __shfl_xor(aU2, m);
            }
            __shared__ double sh3[4][3];
            if ((threadIdx.x & 63) == 0) {
                sh3[threadIdx.x >> 6][0] = aX; sh3[threadIdx.x >> 6][1] = aU; sh3[threadIdx.x >> 6][2] = aU2;
            }
            __syncthreads();
            sumX = sh3[0][0] + sh3[1][0] + sh3[2][0] + sh3[3][0];
            const double sU = sh3[0][1] + sh3[1][1] + sh3[2][1] + sh3[3][1];
            const double sU2 = sh3[0][2] + sh3[1][2] + sh3[2][2] + sh3[3][2];
            sUU = sU + sU2;
            if (blockIdx.x == 0 && threadIdx.x == 0) { A.xr[0] = sUU; A.xr[2] = 0.0; }
        }
        const double itn = in[S_ITN];
        if (itn > 0.0) {             // lsmr.py: "Test for convergence" of iteration itn
            const double normx = sqrt(sumX);
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
        const double bn = sqrt(sUU);
        if (!(bn > 0)) {         // exact solution reached: v keeps its value (lsmr.py "if beta > 0")
            if (threadIdx.x == 0) A.partV[blockIdx.x] = 0.0;
            return;
        }
        ib = 1.0 / bn;
        ba = bn / in[S_ALPHA];
    }
    const int n_pt_blocks = (A.pt_hi - A.pt_lo + 255) / 256;
    double sq = 0.0;
    if ((int)blockIdx.x < n_pt_blocks) {
        const int p0 = A.pt_lo + blockIdx.x * 256;
        const int p1 = min(p0 + 256, A.pt_hi);
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
                    const int64_t oe = A.pt_obs[e];
                    prod[0][jx] = A.eprod[3 * oe];
                    prod[1][jx] = A.eprod[3 * oe + 1];
                    prod[2][jx] = A.eprod[3 * oe + 2];
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
            const double *dp = A.ptab + (int64_t)p * 6 + 3;
            const double acc[3] = {a0 * dp[0], a1 * dp[1], a2 * dp[2]};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int64_t i = (int64_t)A.n_cams * 7 + (int64_t)p * 3 + k;
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
    const double s = block_sum_256(sq, sh);
    if (threadIdx.x == 0) A.partV[blockIdx.x] = s;
}

// multi-rank: the camera part of vt' from the all-reduced raw camera part of J^T ut1' (tbuf);
// LS_UPD_CAM workgroups, their squared-norm partials behind those of the point workgroups
__global__ __launch_bounds__(256) void lsmr_vt_kernel(LsmrArgs A, int parity)
{
    __shared__ double sh[4];
    if (A.S[R_ISTOP] != 0.0) return;
    const double *in = A.S + parity * S_NBUF;
    const int npb = (A.pt_hi - A.pt_lo + 255) / 256;
    const double bn = beta_new(A, sh);
    if (!(bn > 0)) {
        if (threadIdx.x == 0) A.partV[npb + blockIdx.x] = 0.0;
        return;
    }
    const double ib = 1.0 / bn, ba = bn / in[S_ALPHA];
    const int64_t n = (int64_t)A.n_cams * 7;
    double sq = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)LS_UPD_CAM * 256) {
        const double v = (A.tbuf[i] + A.dreg[i] * A.u2[i]) * ib - ba * A.vt[i];
        A.vt[i] = v;
        sq += v * v;
    }
    const double s = block_sum_256(sq, sh);
    if (threadIdx.x == 0) A.partV[npb + blockIdx.x] = s;
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
    double s2;
    if (A.multi) {       // camera part (replicated partials) + the all-reduced point part
        const int npb = (A.pt_hi - A.pt_lo + 255) / 256;
        s2 = sum_partials(A.partV + npb, LS_UPD_CAM, sh) + A.tbuf[(int64_t)A.n_cams * 7];
    } else {
        s2 = sum_partials(A.partV, A.n_partV, sh);
    }
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
    const NvecRange R = nvec_range(A, blockIdx.x, LS_UPD_BLOCKS, LS_UPD_CAM);
    double acc = 0.0;
    for (int64_t i = R.first; i < R.end; i += R.step) {
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

extern "C" int iamx_ba_lsmr_prepare(const double *cams, const double *pts, const double *d,
                                    int n_cams, int n_pts, double *ctab, double *ptab,
                                    void *stream)
{
    IAMX_REQUIRE(cams && pts && d && ctab && ptab, "null pointer");
    IAMX_REQUIRE(n_cams > 0 && n_pts > 0, "bad size");
    const unsigned g = (unsigned)(((int64_t)n_cams + n_pts + 255) / 256);
    hipLaunchKernelGGL(lsmr_prepare_mf_kernel, dim3(g), dim3(256), 0, iamx::as_stream(stream), cams,
                       pts, d, n_cams, n_pts, ctab, ptab);
    return iamx::check_launch("iamx_ba_lsmr_prepare");
}

extern "C" int iamx_ba_lsmr_iterate(const double *ctab, const double *ptab, const double *calib,
                                    const int32_t *pt_idx, const int32_t *cam_ptr,
                                    const int32_t *pt_ptr, const int32_t *pt_obs,
                                    const int32_t *slot_cp, int64_t n_obs, int n_cams, int n_pts,
                                    const double *dreg, double *u1, double *u2, double *vt,
                                    double *h, double *hbar, double *x, double *state,
                                    double *partials, double *xr, double *tbuf, double *eprod,
                                    int n_iter, void *stream)
{
    IAMX_REQUIRE(ctab && ptab && calib && pt_idx && cam_ptr && pt_ptr && pt_obs && slot_cp && dreg &&
                     u1 && u2 && vt && h && hbar && x && state && partials && xr && tbuf,
                 "null pointer");
    IAMX_REQUIRE(n_obs > 0 && n_cams > 0 && n_pts > 0 && n_iter >= 0 && (n_iter & 1) == 0,
                 "bad size (n_iter must be even: the state block is double-buffered)");
    const PartLayout L = part_layout(n_cams, n_pts);
    LsmrArgs A{ctab, ptab, calib, pt_idx, cam_ptr, pt_ptr, pt_obs,
               reinterpret_cast<const int2 *>(slot_cp), n_obs, n_cams, n_pts,
               dreg, u1, u2, vt, h, hbar, x, state,
               partials, partials + L.off_V, partials + L.off_X, L.n_adj, partials + L.off_U2,
               xr, tbuf, 0, 0, n_pts, eprod};
    hipStream_t st = iamx::as_stream(stream);
    for (int it = 0; it < n_iter; ++it) {
        const int parity = it & 1;
        hipLaunchKernelGGL(lsmr_fwd_kernel, dim3(n_cams + LS_U2_BLOCKS), dim3(256), 0, st, A, parity);
        hipLaunchKernelGGL(lsmr_adj_kernel, dim3(L.n_adj), dim3(256), 0, st, A, parity);
        hipLaunchKernelGGL(lsmr_update3_kernel, dim3(LS_UPD_BLOCKS), dim3(256), 0, st, A, parity);
    }
    return iamx::check_launch("iamx_ba_lsmr_iterate");
}

// One phase of one iteration of the multi-rank form: observations and the point part of every
// n-vector sharded by point (this rank owns the points [pt_lo, pt_hi) of the internal order), the
// camera part (7 n_cams entries) replicated.  The caller all-reduces (sum) xr[0..2) after phase
// 0 and tbuf[0 .. 7 n_cams] (7 n_cams + 1 doubles) after phase 1, on the same stream:
//   phase 0: ut', raw camera part of J^T ut1' -> tbuf, local / replicated sums -> xr[0..4)
//   phase 1: stopping tests of the previous iteration (reduced |x|^2), beta', point part of vt'
//            (rank local), its sum of squares -> tbuf[7 n_cams]
//   phase 2: camera part of vt' from the reduced tbuf, alpha', plane rotations, h / hbar / x
// Afterwards the point part of x is complete on its owner only (the entries of the other ranks
// are untouched: all-reduce x[7 n_cams ..) once per solve if x started as zero).
extern "C" int iamx_ba_lsmr_phase(const double *ctab, const double *ptab, const double *calib,
                                  const int32_t *pt_idx, const int32_t *cam_ptr,
                                  const int32_t *pt_ptr, const int32_t *pt_obs,
                                  const int32_t *slot_cp, int64_t n_obs, int n_cams, int n_pts,
                                  int pt_lo, int pt_hi, const double *dreg, double *u1, double *u2,
                                  double *vt, double *h, double *hbar, double *x, double *state,
                                  double *partials, double *xr, double *tbuf, double *eprod, int phase,
                                  int parity, void *stream)
{
    IAMX_REQUIRE(ctab && ptab && calib && pt_idx && cam_ptr && pt_ptr && pt_obs && slot_cp && dreg &&
                     u1 && u2 && vt && h && hbar && x && state && partials && xr && tbuf,
                 "null pointer");
    IAMX_REQUIRE(n_obs >= 0 && n_cams > 0 && n_pts > 0 && phase >= 0 && phase <= 2 &&
                     (parity == 0 || parity == 1) && pt_lo >= 0 && pt_lo <= pt_hi && pt_hi <= n_pts,
                 "bad size / phase / parity / point range");
    const PartLayout L = part_layout(n_cams, n_pts);
    LsmrArgs A{ctab, ptab, calib, pt_idx, cam_ptr, pt_ptr, pt_obs,
               reinterpret_cast<const int2 *>(slot_cp), n_obs, n_cams, n_pts,
               dreg, u1, u2, vt, h, hbar, x, state,
               partials, partials + L.off_V, partials + L.off_X, 0, partials + L.off_U2,
               xr, tbuf, 1, pt_lo, pt_hi, eprod};
    hipStream_t st = iamx::as_stream(stream);
    const int npb = (pt_hi - pt_lo + 255) / 256;
    if (phase == 0) {
        hipLaunchKernelGGL(lsmr_fwd_kernel, dim3(n_cams + LS_U2_BLOCKS), dim3(256), 0, st, A, parity);
        hipLaunchKernelGGL(lsmr_sumU_kernel, dim3(1), dim3(256), 0, st, A, parity);
    } else if (phase == 1) {
        if (npb > 0) hipLaunchKernelGGL(lsmr_adj_kernel, dim3(npb), dim3(256), 0, st, A, parity);
        hipLaunchKernelGGL(lsmr_sumV_kernel, dim3(1), dim3(256), 0, st, A);
    } else {
        hipLaunchKernelGGL(lsmr_vt_kernel, dim3(LS_UPD_CAM), dim3(256), 0, st, A, parity);
        hipLaunchKernelGGL(lsmr_update3_kernel, dim3(LS_UPD_BLOCKS), dim3(256), 0, st, A, parity);
    }
    return iamx::check_launch("iamx_ba_lsmr_phase");
}
