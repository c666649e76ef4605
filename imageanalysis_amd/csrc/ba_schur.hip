// K4b -- normal-equation blocks and the Schur-complement Gauss-Newton step of the bundle
// adjustment (gfx950), float64.
//
// The reference hands scipy.optimize.least_squares a sparsity mask (scripts/lib/optimizer.py:142-169,
// 491-501) and SciPy solves every trust-region subproblem
//     min || [J D; Dreg] p - [r; 0] ||          (scipy/optimize/_lsq/trf.py:303-314)
// with LSMR on the whole (cameras + points) system: ~400 iterations per outer iteration at
// BASELINE configs[3], because column scaling is the only preconditioning (csrc/ba_linalg.hip
// keeps that path).  Here the same subproblem is solved through its normal equations in the
// block form SURVEY.md 8b names (iamx_ba_accumulate: U = C x 7x7, V = P x 3x3, g_c, g_p):
//
//     [ U'   W ] [p_c]   [g'_c]        U' = D_c U D_c + Dreg_c^2,  V' = D_p V D_p + Dreg_p^2
//     [ W^T  V'] [p_p] = [g'_p]        W  = D_c Jc^T Jp D_p  (never formed)
//
// Points are eliminated exactly (V' is block diagonal, 3x3 per point, rank local when the
// observations are sharded by point), the reduced camera system
//     S p_c = g'_c - W V'^-1 g'_p ,    S = U' - W V'^-1 W^T
// is solved by conjugate gradients preconditioned with the inverse of its 7x7 diagonal blocks,
// and p_p = V'^-1 (g'_p - W^T p_c).  S is applied matrix free from the stored Jacobian blocks:
//     t  = Jc (D_c y)                       one thread per observation            (schur_fwd)
//     z' = D_p V'^-1 D_p  sum_o Jp_o^T t_o  one thread per point                  (schur_pt)
//     q  = D_c sum_o Jc_o^T (t_o - Jp_o z') one workgroup per camera              (schur_adj)
// + Dreg_c^2 y in the update kernel.  All three are HBM-bound streams of the blocks
// (Jc 112 B, Jp 48 B, t 16 B per observation).  Several ranks: every sum over observations
// (U, g_c, the diagonal blocks of S, its right-hand side, q) is a partial sum the caller
// all-reduces -- C x (49 + 7) and C x 35 doubles once per outer iteration, C x 7 doubles per CG
// iteration; the scalars of the recurrence are replicated.  No atomics: bitwise reproducible.
//
// optimize_calib='global' (scripts/lib/optimizer.py:142-169,181-189: 8 calibration columns shared
// by every observation) is the BORDERED form of the same system: the calibration block k joins the
// camera side, q = [p_c; p_k], S q = [Jc Jk]^T (I - Jp V'^-1 Jp^T) [Jc Jk] q + Dreg^2 q.  The
// passes above carry the extra term (t += Jk (D_k y_k) in schur_fwd, q_k = D_k sum_o Jk_o^T e_o as
// per-camera partials of schur_adj + one small reduction), the preconditioner gets an 8x8 block
// (D_k (sum Jk^T Jk) D_k + Dreg_k^2)^-1 -- the calibration block of the normal equations, without
// its Schur correction, which couples all observations of a point --, and the recurrence treats
// the block as one more "camera" of 8 parameters stored behind the 7 C camera entries.
#include "iamx_common.h"

namespace {

constexpr int ST_RZ = 0, ST_RZ0 = 1, ST_ITER = 2, ST_STOP = 3, ST_ETA = 4, ST_MAXIT = 5,
              ST_ALPHA = 6, ST_BETA = 7, ST_PQ = 8, ST_Q = 9, ST_QTOL = 10, ST_COUNT = 16;

// index of (i, j), i <= j, in the packed upper triangle of a 7x7 symmetric matrix
__host__ __device__ constexpr int tri7(int i, int j) { return i * 7 - i * (i - 1) / 2 + (j - i); }
// ... of an 8x8 one (the calibration block): 36 entries
__host__ __device__ constexpr int tri8(int i, int j) { return i * 8 - i * (i - 1) / 2 + (j - i); }
constexpr int NK = 8, NK_TRI = 36;

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// sum over the 256 threads of a workgroup of K values per thread; result valid in thread 0
template <int K>
__device__ __forceinline__ void block_sum_k(double (&acc)[K], double *sh /* [4][K] */)
{
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = wave_sum(acc[k]);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sh[w * K + k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = sh[k] + sh[K + k] + sh[2 * K + k] + sh[3 * K + k];
    }
}

// ---- iamx_ba_accumulate -----------------------------------------------------------------------
// camera side: U_c = sum Jc^T Jc (28 unique entries, written as the full 7x7), g_c = sum Jc^T r
__global__ __launch_bounds__(256) void acc_cam_kernel(const double *__restrict__ Jc,
                                                      const double *__restrict__ r,
                                                      const int32_t *__restrict__ cam_ptr,
                                                      double *__restrict__ U, double *__restrict__ gc)
{
    __shared__ double sh[4 * 35];
    const int c = blockIdx.x;
    double acc[35];
#pragma unroll
    for (int k = 0; k < 35; ++k) acc[k] = 0.0;
    for (int o = cam_ptr[c] + threadIdx.x; o < cam_ptr[c + 1]; o += 256) {
        double j0[7], j1[7];
        const double *jc = Jc + (int64_t)o * 14;
#pragma unroll
        for (int k = 0; k < 7; ++k) { j0[k] = jc[k]; j1[k] = jc[7 + k]; }
        const double2 rr = *reinterpret_cast<const double2 *>(r + 2 * (int64_t)o);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
#pragma unroll
            for (int j = i; j < 7; ++j) acc[tri7(i, j)] += j0[i] * j0[j] + j1[i] * j1[j];
            acc[28 + i] += j0[i] * rr.x + j1[i] * rr.y;
        }
    }
    block_sum_k<35>(acc, sh);
    if (threadIdx.x == 0) {
        double *u = U + (int64_t)c * 49;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
#pragma unroll
            for (int j = i; j < 7; ++j) u[i * 7 + j] = u[j * 7 + i] = acc[tri7(i, j)];
            gc[(int64_t)c * 7 + i] = acc[28 + i];
        }
    }
}

// point side: V_p = sum Jp^T Jp (full 3x3), g_p = sum Jp^T r; one thread per point
__global__ __launch_bounds__(256) void acc_pt_kernel(const double *__restrict__ Jp,
                                                     const double *__restrict__ r,
                                                     const int32_t *__restrict__ pt_ptr,
                                                     const int32_t *__restrict__ pt_obs, int n_pts,
                                                     double *__restrict__ V, double *__restrict__ gp)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pts) return;
    double v00 = 0, v01 = 0, v02 = 0, v11 = 0, v12 = 0, v22 = 0, g0 = 0, g1 = 0, g2 = 0;
    for (int e = pt_ptr[p]; e < pt_ptr[p + 1]; ++e) {
        const int o = pt_obs[e];
        const double *jp = Jp + (int64_t)o * 6;
        const double a0 = jp[0], a1 = jp[1], a2 = jp[2], b0 = jp[3], b1 = jp[4], b2 = jp[5];
        const double2 rr = *reinterpret_cast<const double2 *>(r + 2 * (int64_t)o);
        v00 += a0 * a0 + b0 * b0; v01 += a0 * a1 + b0 * b1; v02 += a0 * a2 + b0 * b2;
        v11 += a1 * a1 + b1 * b1; v12 += a1 * a2 + b1 * b2; v22 += a2 * a2 + b2 * b2;
        g0 += a0 * rr.x + b0 * rr.y; g1 += a1 * rr.x + b1 * rr.y; g2 += a2 * rr.x + b2 * rr.y;
    }
    double *v = V + (int64_t)p * 9;
    v[0] = v00; v[1] = v01; v[2] = v02; v[3] = v01; v[4] = v11; v[5] = v12;
    v[6] = v02; v[7] = v12; v[8] = v22;
    gp[(int64_t)p * 3 + 0] = g0; gp[(int64_t)p * 3 + 1] = g1; gp[(int64_t)p * 3 + 2] = g2;
}

// ---- point elimination ------------------------------------------------------------------------
// Y = (D_p V D_p + Dreg_p^2)^-1 (6 unique), yg = Y (d_p .* g_p), zp = d_p .* yg
__global__ __launch_bounds__(256) void schur_points_kernel(const double *__restrict__ V,
                                                           const double *__restrict__ gp,
                                                           const double *__restrict__ d_p,
                                                           const double *__restrict__ dreg_p,
                                                           int n_pts, double *__restrict__ Y,
                                                           double *__restrict__ yg,
                                                           double *__restrict__ zp)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pts) return;
    const double *v = V + (int64_t)p * 9;
    const double d0 = d_p[(int64_t)p * 3], d1 = d_p[(int64_t)p * 3 + 1], d2 = d_p[(int64_t)p * 3 + 2];
    const double l0 = dreg_p[(int64_t)p * 3], l1 = dreg_p[(int64_t)p * 3 + 1],
                 l2 = dreg_p[(int64_t)p * 3 + 2];
    const double a00 = d0 * d0 * v[0] + l0 * l0, a01 = d0 * d1 * v[1], a02 = d0 * d2 * v[2];
    const double a11 = d1 * d1 * v[4] + l1 * l1, a12 = d1 * d2 * v[5];
    const double a22 = d2 * d2 * v[8] + l2 * l2;
    // symmetric 3x3 inverse through the cofactors
    const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    const double det = a00 * c00 + a01 * c01 + a02 * c02;
    double y00, y01, y02, y11, y12, y22;
    // V' is positive definite whenever dreg > 0; a block that is numerically not (det below
    // 1e-14 of the product of its diagonal) falls back to its diagonal
    if (det > 1e-14 * a00 * a11 * a22 && a00 > 0 && a11 > 0 && a22 > 0) {
        const double id = 1.0 / det;
        y00 = c00 * id; y01 = c01 * id; y02 = c02 * id; y11 = c11 * id; y12 = c12 * id; y22 = c22 * id;
    } else {
        y00 = a00 > 0 ? 1.0 / a00 : 0.0; y11 = a11 > 0 ? 1.0 / a11 : 0.0; y22 = a22 > 0 ? 1.0 / a22 : 0.0;
        y01 = y02 = y12 = 0.0;
    }
    double *y = Y + (int64_t)p * 6;
    y[0] = y00; y[1] = y01; y[2] = y02; y[3] = y11; y[4] = y12; y[5] = y22;
    const double g0 = d0 * gp[(int64_t)p * 3], g1 = d1 * gp[(int64_t)p * 3 + 1],
                 g2 = d2 * gp[(int64_t)p * 3 + 2];
    const double h0 = y00 * g0 + y01 * g1 + y02 * g2, h1 = y01 * g0 + y11 * g1 + y12 * g2,
                 h2 = y02 * g0 + y12 * g1 + y22 * g2;
    yg[(int64_t)p * 3] = h0; yg[(int64_t)p * 3 + 1] = h1; yg[(int64_t)p * 3 + 2] = h2;
    zp[(int64_t)p * 3] = d0 * h0; zp[(int64_t)p * 3 + 1] = d1 * h1; zp[(int64_t)p * 3 + 2] = d2 * h2;
}

// ---- the three passes of  q = S y ----------------------------------------------------------------
// Jk / yk: the calibration columns [O][2][8] and the 8 scaled calibration entries of y (nullptr
// without them)
__global__ __launch_bounds__(256) void schur_fwd_kernel(const double *__restrict__ Jc,
                                                        const double *__restrict__ Jk,
                                                        const int32_t *__restrict__ cam_idx,
                                                        int64_t n_obs, const double *__restrict__ yv,
                                                        const double *__restrict__ yk,
                                                        const double *__restrict__ state,
                                                        double *__restrict__ t)
{
    if (state && state[ST_STOP] != 0.0) return;
    double ykr[NK];
    if (Jk) {
#pragma unroll
        for (int k = 0; k < NK; ++k) ykr[k] = yk[k];
    }
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n_obs;
         o += (int64_t)gridDim.x * 256) {
        const double *y = yv + (int64_t)cam_idx[o] * 7;
        const double *jc = Jc + o * 14;
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const double v = y[k];
            a += jc[k] * v;
            b += jc[7 + k] * v;
        }
        if (Jk) {
            const double *jk = Jk + o * 16;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                a += jk[k] * ykr[k];
                b += jk[NK + k] * ykr[k];
            }
        }
        *reinterpret_cast<double2 *>(t + 2 * o) = make_double2(a, b);
    }
}

// FINAL = false: zp = d_p .* (Y (d_p .* sum Jp^T t))
// FINAL = true : out_p = yg - Y (d_p .* sum Jp^T t)   (the point part of the step; 0 for the
//                points another rank owns)
template <bool FINAL>
__global__ __launch_bounds__(256) void schur_pt_kernel(const double *__restrict__ Jp,
                                                       const int32_t *__restrict__ pt_ptr,
                                                       const int32_t *__restrict__ pt_obs, int n_pts,
                                                       int pt_lo, int pt_hi,
                                                       const double *__restrict__ t,
                                                       const double *__restrict__ d_p,
                                                       const double *__restrict__ Y,
                                                       const double *__restrict__ yg,
                                                       const double *__restrict__ state,
                                                       double *__restrict__ out)
{
    if (!FINAL && state && state[ST_STOP] != 0.0) return;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pts) return;
    if (FINAL && (p < pt_lo || p >= pt_hi)) {
        out[(int64_t)p * 3] = out[(int64_t)p * 3 + 1] = out[(int64_t)p * 3 + 2] = 0.0;
        return;
    }
    double s0 = 0, s1 = 0, s2 = 0;
    for (int e = pt_ptr[p]; e < pt_ptr[p + 1]; ++e) {
        const int o = pt_obs[e];
        const double *jp = Jp + (int64_t)o * 6;
        const double2 tt = *reinterpret_cast<const double2 *>(t + 2 * (int64_t)o);
        s0 += jp[0] * tt.x + jp[3] * tt.y;
        s1 += jp[1] * tt.x + jp[4] * tt.y;
        s2 += jp[2] * tt.x + jp[5] * tt.y;
    }
    const double d0 = d_p[(int64_t)p * 3], d1 = d_p[(int64_t)p * 3 + 1], d2 = d_p[(int64_t)p * 3 + 2];
    s0 *= d0; s1 *= d1; s2 *= d2;
    const double *y = Y + (int64_t)p * 6;
    const double h0 = y[0] * s0 + y[1] * s1 + y[2] * s2, h1 = y[1] * s0 + y[3] * s1 + y[4] * s2,
                 h2 = y[2] * s0 + y[4] * s1 + y[5] * s2;
    if (FINAL) {
        out[(int64_t)p * 3] = yg[(int64_t)p * 3] - h0;
        out[(int64_t)p * 3 + 1] = yg[(int64_t)p * 3 + 1] - h1;
        out[(int64_t)p * 3 + 2] = yg[(int64_t)p * 3 + 2] - h2;
    } else {
        out[(int64_t)p * 3] = d0 * h0;
        out[(int64_t)p * 3 + 1] = d1 * h1;
        out[(int64_t)p * 3 + 2] = d2 * h2;
    }
}

// one workgroup per camera over its (camera-major, contiguous) observations.
// BLOCKS = false: q_raw[c] = d_c .* sum Jc^T (t - Jp zp)                              (7)
// BLOCKS = true : the same with t = r (the right-hand side of the reduced system) and the
//                 camera's diagonal block of S before regularisation,
//                 d_c d_c^T .* sum Jc^T (I - Jp (D_p Y D_p) Jp^T) Jc   (28 unique)  -> sraw[c][35]
// CALIB: additionally the camera's partial sums of the calibration block, unscaled, to
//         ckpart[c][KC]: sum Jk^T e (8); with BLOCKS in front of them sum Jk^T Jk (36 unique)
template <bool BLOCKS, bool CALIB>
__global__ __launch_bounds__(256) void schur_adj_kernel(const double *__restrict__ Jc,
                                                        const double *__restrict__ Jp,
                                                        const double *__restrict__ Jk,
                                                        const int32_t *__restrict__ cam_ptr,
                                                        const int32_t *__restrict__ pt_idx,
                                                        const double *__restrict__ t,
                                                        const double *__restrict__ zp,
                                                        const double *__restrict__ d_c,
                                                        const double *__restrict__ d_p,
                                                        const double *__restrict__ Y,
                                                        const double *__restrict__ state,
                                                        const double *__restrict__ pv,
                                                        const double *__restrict__ dreg_c,
                                                        double *__restrict__ pqpart,
                                                        double *__restrict__ out,
                                                        double *__restrict__ ckpart)
{
    constexpr int KB = BLOCKS ? 35 : 7;                  // camera sums
    constexpr int KC = CALIB ? (BLOCKS ? NK_TRI + NK : NK) : 0;
    constexpr int K = KB + KC;
    __shared__ double sh[4 * K];
    if (!BLOCKS && state && state[ST_STOP] != 0.0) return;
    const int c = blockIdx.x;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    for (int o = cam_ptr[c] + threadIdx.x; o < cam_ptr[c + 1]; o += 256) {
        const int p = pt_idx[o];
        const double *jc = Jc + (int64_t)o * 14;
        const double *jp = Jp + (int64_t)o * 6;
        double j0[7], j1[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) { j0[k] = jc[k]; j1[k] = jc[7 + k]; }
        const double a0 = jp[0], a1 = jp[1], a2 = jp[2], b0 = jp[3], b1 = jp[4], b2 = jp[5];
        const double2 tt = *reinterpret_cast<const double2 *>(t + 2 * (int64_t)o);
        const double z0 = zp[(int64_t)p * 3], z1 = zp[(int64_t)p * 3 + 1], z2 = zp[(int64_t)p * 3 + 2];
        const double e0 = tt.x - (a0 * z0 + a1 * z1 + a2 * z2);
        const double e1 = tt.y - (b0 * z0 + b1 * z1 + b2 * z2);
        if (BLOCKS) {
            const double d0 = d_p[(int64_t)p * 3], d1 = d_p[(int64_t)p * 3 + 1],
                         d2 = d_p[(int64_t)p * 3 + 2];
            const double *y = Y + (int64_t)p * 6;
            // G = I - E Ys E^T,  E = Jp_o (rows a, b),  Ys = D_p Y D_p
            const double sa0 = d0 * a0, sa1 = d1 * a1, sa2 = d2 * a2;
            const double sb0 = d0 * b0, sb1 = d1 * b1, sb2 = d2 * b2;
            const double ya0 = y[0] * sa0 + y[1] * sa1 + y[2] * sa2;
            const double ya1 = y[1] * sa0 + y[3] * sa1 + y[4] * sa2;
            const double ya2 = y[2] * sa0 + y[4] * sa1 + y[5] * sa2;
            const double yb0 = y[0] * sb0 + y[1] * sb1 + y[2] * sb2;
            const double yb1 = y[1] * sb0 + y[3] * sb1 + y[4] * sb2;
            const double yb2 = y[2] * sb0 + y[4] * sb1 + y[5] * sb2;
            const double g00 = 1.0 - (sa0 * ya0 + sa1 * ya1 + sa2 * ya2);
            const double g01 = -(sa0 * yb0 + sa1 * yb1 + sa2 * yb2);
            const double g11 = 1.0 - (sb0 * yb0 + sb1 * yb1 + sb2 * yb2);
            double h0[7], h1[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                h0[k] = g00 * j0[k] + g01 * j1[k];
                h1[k] = g01 * j0[k] + g11 * j1[k];
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) {
#pragma unroll
                for (int j = i; j < 7; ++j) acc[tri7(i, j)] += j0[i] * h0[j] + j1[i] * h1[j];
            }
        }
        constexpr int B = BLOCKS ? 28 : 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[B + k] += j0[k] * e0 + j1[k] * e1;
        if (CALIB) {
            const double *jk = Jk + (int64_t)o * 16;
            double k0[NK], k1[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) { k0[k] = jk[k]; k1[k] = jk[NK + k]; }
            constexpr int BK = KB + (BLOCKS ? NK_TRI : 0);
#pragma unroll
            for (int k = 0; k < NK; ++k) acc[BK + k] += k0[k] * e0 + k1[k] * e1;
            if (BLOCKS) {
#pragma unroll
                for (int i = 0; i < NK; ++i) {
#pragma unroll
                    for (int j = i; j < NK; ++j) acc[KB + tri8(i, j)] += k0[i] * k0[j] + k1[i] * k1[j];
                }
            }
        }
    }
    block_sum_k<K>(acc, sh);
    if (threadIdx.x == 0) {
        if (CALIB) {
#pragma unroll
            for (int k = 0; k < KC; ++k) ckpart[(int64_t)c * KC + k] = acc[KB + k];
        }
        const double *dc = d_c + (int64_t)c * 7;
        if (BLOCKS) {
            double *s = out + (int64_t)c * 35;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
#pragma unroll
                for (int j = i; j < 7; ++j) s[tri7(i, j)] = dc[i] * dc[j] * acc[tri7(i, j)];
                s[28 + i] = dc[i] * acc[28 + i];
            }
        } else {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const double q = dc[k] * acc[k];
                out[(int64_t)c * 7 + k] = q;
                if (pqpart) {                      // one rank: q is complete, p.(q + Dreg^2 p) here
                    const double l = dreg_c[(int64_t)c * 7 + k], p = pv[(int64_t)c * 7 + k];
                    s += p * (q + l * l * p);
                }
            }
            if (pqpart) pqpart[c] = s;
        }
    }
}

// The calibration partials of schur_adj summed over the cameras (one workgroup), scaled:
// KC = 8 : qk[k] = d_k[k] * sum_c ckpart[c][k]; with pqslot also p_k . (q_k + Dreg_k^2 p_k)
// KC = 44: sk[tri8(i,j)] = d_k[i] d_k[j] * sum (36), sk[36 + k] = d_k[k] * sum (the right-hand side)
template <int KC>
__global__ __launch_bounds__(256) void schur_calib_reduce_kernel(const double *__restrict__ ckpart,
                                                                 int n_cams,
                                                                 const double *__restrict__ d_k,
                                                                 const double *__restrict__ dreg_k,
                                                                 const double *__restrict__ pk,
                                                                 const double *__restrict__ state,
                                                                 double *__restrict__ outk,
                                                                 double *__restrict__ pqslot)
{
    __shared__ double sh[4 * KC];
    if (state && state[ST_STOP] != 0.0) return;
    double acc[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) acc[k] = 0.0;
    for (int c = threadIdx.x; c < n_cams; c += 256) {
#pragma unroll
        for (int k = 0; k < KC; ++k) acc[k] += ckpart[(int64_t)c * KC + k];
    }
    block_sum_k<KC>(acc, sh);
    if (threadIdx.x == 0) {
        if (KC == NK) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const double q = d_k[k] * acc[k];
                outk[k] = q;
                if (pqslot) s += pk[k] * (q + dreg_k[k] * dreg_k[k] * pk[k]);
            }
            if (pqslot) *pqslot = s;
        } else {
#pragma unroll
            for (int i = 0; i < NK; ++i) {
#pragma unroll
                for (int j = i; j < NK; ++j) outk[tri8(i, j)] = d_k[i] * d_k[j] * acc[tri8(i, j)];
                outk[NK_TRI + i] = d_k[i] * acc[NK_TRI + i];
            }
        }
    }
}

// symmetric 8x8 (packed upper triangle) times vector
__device__ __forceinline__ void sym8_apply(const double *__restrict__ m /* 36 */, const double *v,
                                           double *out)
{
    for (int i = 0; i < NK; ++i) out[i] = 0.0;
    for (int i = 0; i < NK; ++i) {
        for (int j = i; j < NK; ++j) {
            const double a = m[tri8(i, j)];
            out[i] += a * v[j];
            if (j != i) out[j] += a * v[i];
        }
    }
}

// the calibration block of the preconditioner and of the start of the recurrence (one thread):
// a = sk (scaled) + Dreg_k^2, M_k = a^-1 through a Cholesky factorisation (diagonal fallback),
// x_k = 0, r_k = rhs_k, z_k = p_k = M_k r_k, y_k = d_k .* p_k.  Returns r_k . z_k.
__device__ double calib_factor(const double *__restrict__ sk, const double *__restrict__ d_k,
                               const double *__restrict__ dreg_k, double *__restrict__ mk,
                               double *__restrict__ x, double *__restrict__ r, double *__restrict__ z,
                               double *__restrict__ pv, double *__restrict__ yv)
{
    double a[NK][NK], L[NK][NK], Li[NK][NK];
    for (int i = 0; i < NK; ++i)
        for (int j = i; j < NK; ++j) a[i][j] = a[j][i] = sk[tri8(i, j)] + (i == j ? dreg_k[i] * dreg_k[i] : 0.0);
    bool ok = true;
    for (int j = 0; j < NK; ++j) {
        double dsum = a[j][j];
        for (int k = 0; k < j; ++k) dsum -= L[j][k] * L[j][k];
        if (!(dsum > 1e-14 * a[j][j]) || !(a[j][j] > 0.0)) { ok = false; dsum = 1.0; }
        L[j][j] = sqrt(dsum);
        for (int i = j + 1; i < NK; ++i) {
            double v = a[i][j];
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
            L[i][j] = v / L[j][j];
        }
    }
    if (ok) {
        for (int j = 0; j < NK; ++j) {                 // Li = L^-1 (lower)
            Li[j][j] = 1.0 / L[j][j];
            for (int i = j + 1; i < NK; ++i) {
                double v = 0.0;
                for (int k = j; k < i; ++k) v -= L[i][k] * Li[k][j];
                Li[i][j] = v / L[i][i];
            }
        }
        for (int i = 0; i < NK; ++i)
            for (int j = i; j < NK; ++j) {             // M = Li^T Li
                double v = 0.0;
                for (int k = j; k < NK; ++k) v += Li[k][i] * Li[k][j];
                mk[tri8(i, j)] = v;
            }
    } else {
        for (int i = 0; i < NK; ++i)
            for (int j = i; j < NK; ++j) mk[tri8(i, j)] = (i == j) ? (a[i][i] > 0.0 ? 1.0 / a[i][i] : 1.0) : 0.0;
    }
    double rc[NK], zc[NK], rz = 0.0;
    for (int k = 0; k < NK; ++k) rc[k] = sk[NK_TRI + k];
    sym8_apply(mk, rc, zc);
    for (int k = 0; k < NK; ++k) {
        x[k] = 0.0;
        r[k] = rc[k];
        z[k] = zc[k];
        pv[k] = zc[k];
        yv[k] = d_k[k] * zc[k];
        rz += rc[k] * zc[k];
    }
    return rz;
}

// sum over the 1024 threads of the single update workgroup (all threads get the result)
__device__ __forceinline__ double block_sum_1024(double v, double *sh /* [16] */)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sh[k];
    return s;
}

__device__ __forceinline__ void sym7_apply(const double *__restrict__ m /* 28 */, const double *v,
                                           double *out)
{
#pragma unroll
    for (int i = 0; i < 7; ++i) out[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
#pragma unroll
        for (int j = i; j < 7; ++j) {
            const double a = m[tri7(i, j)];
            out[i] += a * v[j];
            if (j != i) out[j] += a * v[i];
        }
    }
}

// One workgroup: S'_cc = sraw (all-reduced) + Dreg_c^2, its inverse through a Cholesky
// factorisation (diagonal fallback for a block that is not positive definite), and the start of
// the recurrence: x = 0, r = rhs, z = M^-1 r, p = z, y = d_c .* p, rz = rz0 = r.z
__global__ __launch_bounds__(1024) void schur_factor_kernel(const double *__restrict__ sraw,
                                                            const double *__restrict__ d_c,
                                                            const double *__restrict__ dreg_c,
                                                            const double *__restrict__ d_k,
                                                            const double *__restrict__ dreg_k,
                                                            int n_cams, double eta, double qtol,
                                                            double maxiter,
                                                            double *__restrict__ minv,
                                                            double *__restrict__ x,
                                                            double *__restrict__ r,
                                                            double *__restrict__ z,
                                                            double *__restrict__ pv,
                                                            double *__restrict__ yv,
                                                            double *__restrict__ state)
{
    __shared__ double sh[16];
    double rz = 0.0;
    for (int c = threadIdx.x; c < n_cams; c += 1024) {
        const double *s = sraw + (int64_t)c * 35;
        double a[28], L[28], Li[28];
#pragma unroll
        for (int k = 0; k < 28; ++k) a[k] = s[k];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const double l = dreg_c[(int64_t)c * 7 + i];
            a[tri7(i, i)] += l * l;
        }
        // Cholesky a = L L^T; L stored by (row i >= col j) at tri7(j, i)
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            double dsum = a[tri7(j, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) dsum -= L[tri7(k, j)] * L[tri7(k, j)];
            if (!(dsum > 1e-14 * a[tri7(j, j)]) || !(a[tri7(j, j)] > 0.0)) { ok = false; dsum = 1.0; }
            const double ljj = sqrt(dsum);
            L[tri7(j, j)] = ljj;
            const double inv = 1.0 / ljj;
#pragma unroll
            for (int i = j + 1; i < 7; ++i) {
                double v = a[tri7(j, i)];
#pragma unroll
                for (int k = 0; k < j; ++k) v -= L[tri7(k, i)] * L[tri7(k, j)];
                L[tri7(j, i)] = v * inv;
            }
        }
        double *m = minv + (int64_t)c * 28;
        if (ok) {
            // Li = L^-1 (lower; Li[i][j] at tri7(j, i)), then M = Li^T Li
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                Li[tri7(j, j)] = 1.0 / L[tri7(j, j)];
#pragma unroll
                for (int i = j + 1; i < 7; ++i) {
                    double v = 0.0;
#pragma unroll
                    for (int k = j; k < i; ++k) v -= L[tri7(k, i)] * Li[tri7(j, k)];
                    Li[tri7(j, i)] = v / L[tri7(i, i)];
                }
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) {
#pragma unroll
                for (int j = i; j < 7; ++j) {
                    double v = 0.0;
#pragma unroll
                    for (int k = j; k < 7; ++k) v += Li[tri7(i, k)] * Li[tri7(j, k)];
                    m[tri7(i, j)] = v;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
#pragma unroll
                for (int j = i; j < 7; ++j)
                    m[tri7(i, j)] = (i == j) ? (a[tri7(i, i)] > 0.0 ? 1.0 / a[tri7(i, i)] : 1.0) : 0.0;
            }
        }
        double rc[7], zc[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) rc[k] = s[28 + k];
        sym7_apply(m, rc, zc);
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int64_t i = (int64_t)c * 7 + k;
            x[i] = 0.0;
            r[i] = rc[k];
            z[i] = zc[k];
            pv[i] = zc[k];
            yv[i] = d_c[i] * zc[k];
            rz += rc[k] * zc[k];
        }
    }
    if (d_k && threadIdx.x == 0) {
        // the calibration block sits behind the cameras in every vector of the recurrence
        const int64_t nc = (int64_t)n_cams * 7;
        rz += calib_factor(sraw + (int64_t)n_cams * 35, d_k, dreg_k, minv + (int64_t)n_cams * 28,
                           x + nc, r + nc, z + nc, pv + nc, yv + nc);
    }
    rz = block_sum_1024(rz, sh);
    if (threadIdx.x == 0) {
        for (int k = 0; k < 2 * ST_COUNT; ++k) state[k] = 0.0;
        state[ST_RZ] = rz;
        state[ST_RZ0] = rz;
        state[ST_ETA] = eta;
        state[ST_QTOL] = qtol;
        state[ST_MAXIT] = maxiter;
        // nothing to solve (zero right-hand side) or a preconditioner that is not positive
        state[ST_STOP] = (rz > 0.0) ? 0.0 : (rz == 0.0 ? 1.0 : 3.0);
    }
}

// ---- the CG recurrence: three small multi-workgroup launches per iteration ------------------------
// The scalars live in a double-buffered state block (iteration i reads buffer i & 1 and writes
// the other one), the two inner products go through per-camera partials that EVERY workgroup
// adds up in the same order (bit-identical in all of them, no atomics, no grid-wide barrier).
//   schur_pq_kernel      pqpart[c] = p_c . (q_raw_c + Dreg_c^2 p_c)      (one rank: schur_adj does it)
//   schur_update1_kernel alpha = rz / sum pqpart; x += alpha p; r -= alpha q; z = M^-1 r;
//                        rzpart[c] = r_c . z_c
//   schur_update2_kernel beta = rz' / rz; stopping tests (1: sqrt(rz'/rz0) <= eta; 4: i (Q_{i-1} -
//                        Q_i) <= qtol (-Q_i), the decrease of the quadratic model has levelled
//                        off -- Nash & Sofer's truncated-Newton test; 2: iteration count; 3: p.q <= 0
//                        or NaN, breakdown: x is kept); p = z + beta p; y = d_c .* p; workgroup 0
//                        writes the next state buffer.  Behind a latched stop it only copies the
//                        state across, so that everything enqueued later stays a no-op.
__device__ __forceinline__ double sum_partials_256(const double *__restrict__ part, int n, double *sh)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += part[i];
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void schur_pq_kernel(const double *__restrict__ qraw,
                                                       const double *__restrict__ dreg_c,
                                                       const double *__restrict__ dreg_k,
                                                       const double *__restrict__ pv, int n_cams,
                                                       const double *__restrict__ state,
                                                       double *__restrict__ pqpart)
{
    if (state[ST_STOP] != 0.0) return;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (dreg_k && c == n_cams) {                       // the calibration block: slot n_cams
        double s = 0.0;
        for (int k = 0; k < NK; ++k) {
            const int64_t i = (int64_t)n_cams * 7 + k;
            s += pv[i] * (qraw[i] + dreg_k[k] * dreg_k[k] * pv[i]);
        }
        pqpart[c] = s;
        return;
    }
    if (c >= n_cams) return;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int64_t i = (int64_t)c * 7 + k;
        const double l = dreg_c[i], p = pv[i];
        s += p * (qraw[i] + l * l * p);
    }
    pqpart[c] = s;
}

__global__ __launch_bounds__(256) void schur_update1_kernel(const double *__restrict__ qraw,
                                                            const double *__restrict__ dreg_c,
                                                            const double *__restrict__ dreg_k,
                                                            const double *__restrict__ minv,
                                                            int n_cams, double *__restrict__ x,
                                                            double *__restrict__ r,
                                                            double *__restrict__ z,
                                                            const double *__restrict__ pv,
                                                            const double *__restrict__ state,
                                                            const double *__restrict__ pqpart,
                                                            double *__restrict__ rzpart)
{
    __shared__ double sh[4];
    if (state[ST_STOP] != 0.0) return;
    const int n_part = n_cams + (dreg_k ? 1 : 0);
    const double pq = sum_partials_256(pqpart, n_part, sh);
    if (!(pq > 0.0)) return;                       // update2 latches the breakdown
    const double alpha = state[ST_RZ] / pq;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (dreg_k && c == n_cams) {                       // the calibration block
        const int64_t nc = (int64_t)n_cams * 7;
        double rk[NK], zk[NK], s = 0.0;
        for (int k = 0; k < NK; ++k) {
            const double l = dreg_k[k], p = pv[nc + k];
            x[nc + k] += alpha * p;
            rk[k] = r[nc + k] - alpha * (qraw[nc + k] + l * l * p);
            r[nc + k] = rk[k];
        }
        sym8_apply(minv + (int64_t)n_cams * 28, rk, zk);
        for (int k = 0; k < NK; ++k) {
            z[nc + k] = zk[k];
            s += rk[k] * zk[k];
        }
        rzpart[c] = s;
        return;
    }
    if (c >= n_cams) return;
    double rc[7], zc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int64_t i = (int64_t)c * 7 + k;
        const double l = dreg_c[i], p = pv[i];
        x[i] += alpha * p;
        rc[k] = r[i] - alpha * (qraw[i] + l * l * p);
        r[i] = rc[k];
    }
    sym7_apply(minv + (int64_t)c * 28, rc, zc);
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        z[(int64_t)c * 7 + k] = zc[k];
        s += rc[k] * zc[k];
    }
    rzpart[c] = s;
}

__global__ __launch_bounds__(256) void schur_update2_kernel(const double *__restrict__ d_c,
                                                            const double *__restrict__ d_k, int n_cams,
                                                            const double *__restrict__ z,
                                                            double *__restrict__ pv,
                                                            double *__restrict__ yv,
                                                            const double *__restrict__ state,
                                                            double *__restrict__ next,
                                                            const double *__restrict__ pqpart,
                                                            const double *__restrict__ rzpart)
{
    __shared__ double sh[4];
    if (state[ST_STOP] != 0.0) {
        if (blockIdx.x == 0 && threadIdx.x < ST_COUNT) next[threadIdx.x] = state[threadIdx.x];
        return;
    }
    const int n_part = n_cams + (d_k ? 1 : 0);
    const double pq = sum_partials_256(pqpart, n_part, sh);
    if (!(pq > 0.0)) {
        if (blockIdx.x == 0 && threadIdx.x < ST_COUNT)
            next[threadIdx.x] = threadIdx.x == ST_STOP ? 3.0 : (threadIdx.x == ST_PQ ? pq : state[threadIdx.x]);
        return;
    }
    const double rz = state[ST_RZ];
    const double alpha = rz / pq;
    const double rzn = sum_partials_256(rzpart, n_part, sh);
    const double beta = rzn / rz;
    const double iter = state[ST_ITER] + 1.0;
    const double eta = state[ST_ETA];
    // the quadratic model Q(x) = x.Sx / 2 - b.x falls by alpha r.z / 2 in this iteration
    const double dq = 0.5 * alpha * rz;
    const double q = state[ST_Q] + dq;                              // = -Q(x_i) > 0
    double stop = 0.0;
    if (!(rzn == rzn)) stop = 3.0;                                  // NaN in the recurrence
    else if (rzn <= eta * eta * state[ST_RZ0]) stop = 1.0;
    else if (iter * dq <= state[ST_QTOL] * q) stop = 4.0;
    else if (iter >= state[ST_MAXIT]) stop = 2.0;
    if (stop == 0.0) {
        const int c = blockIdx.x * 256 + threadIdx.x;
        if (c < n_cams) {
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int64_t i = (int64_t)c * 7 + k;
                const double p = z[i] + beta * pv[i];
                pv[i] = p;
                yv[i] = d_c[i] * p;
            }
        } else if (d_k && c == n_cams) {
            for (int k = 0; k < NK; ++k) {
                const int64_t i = (int64_t)n_cams * 7 + k;
                const double p = z[i] + beta * pv[i];
                pv[i] = p;
                yv[i] = d_k[k] * p;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int k = 0; k < ST_COUNT; ++k) next[k] = state[k];
        next[ST_RZ] = rzn;
        next[ST_ITER] = iter;
        next[ST_ALPHA] = alpha;
        next[ST_BETA] = beta;
        next[ST_PQ] = pq;
        next[ST_Q] = q;
        next[ST_STOP] = stop;
    }
}

__global__ __launch_bounds__(256) void schur_scale_kernel(int64_t n, const double *__restrict__ d,
                                                          const double *__restrict__ x,
                                                          double *__restrict__ y,
                                                          double *__restrict__ copy)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        y[i] = d[i] * x[i];
        if (copy) copy[i] = x[i];
    }
}

// diag(U), diag(V) -> the n-vector of the column sums of J.^2 (x_scale='jac')
__global__ __launch_bounds__(256) void block_diag_kernel(const double *__restrict__ U,
                                                         const double *__restrict__ V, int64_t nc,
                                                         int64_t n, double *__restrict__ out)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (i < nc) {
            const int64_t c = i / 7, k = i - c * 7;
            out[i] = U[c * 49 + k * 8];
        } else {
            const int64_t j = i - nc, p = j / 3, k = j - p * 3;
            out[i] = V[p * 9 + k * 4];
        }
    }
}

inline unsigned grid_for(int64_t n)
{
    int64_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int iamx_ba_accumulate(const double *Jc, const double *Jp, const double *r,
                                  const int32_t *cam_ptr, const int32_t *pt_ptr,
                                  const int32_t *pt_obs, int64_t n_obs, int n_cams, int n_pts,
                                  double *U, double *V, double *gc, double *gp, void *stream)
{
    IAMX_REQUIRE(Jc && Jp && r && cam_ptr && pt_ptr && pt_obs && U && V && gc && gp, "null pointer");
    IAMX_REQUIRE(n_obs >= 0 && n_cams >= 0 && n_pts >= 0, "negative size");
    hipStream_t st = iamx::as_stream(stream);
    if (n_cams)
        hipLaunchKernelGGL(acc_cam_kernel, dim3(n_cams), dim3(256), 0, st, Jc, r, cam_ptr, U, gc);
    if (n_pts)
        hipLaunchKernelGGL(acc_pt_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, st, Jp, r, pt_ptr,
                           pt_obs, n_pts, V, gp);
    return iamx::check_launch("iamx_ba_accumulate");
}

extern "C" int iamx_ba_block_diag(const double *U, const double *V, int n_cams, int n_pts,
                                  double *out, void *stream)
{
    IAMX_REQUIRE(U && V && out, "null pointer");
    IAMX_REQUIRE(n_cams >= 0 && n_pts >= 0, "negative size");
    const int64_t nc = (int64_t)n_cams * 7, n = nc + (int64_t)n_pts * 3;
    if (n == 0) return IAMX_OK;
    hipLaunchKernelGGL(block_diag_kernel, dim3(grid_for(n)), dim3(256), 0, iamx::as_stream(stream),
                       U, V, nc, n, out);
    return iamx::check_launch("iamx_ba_block_diag");
}

extern "C" int iamx_ba_schur_state_size(void) { return 2 * ST_COUNT; }

// Jk (DEV [O][2][8], the calibration columns of iamx_ba_residual_jac) selects the bordered form
// (optimize_calib='global'); nullptr: cameras and points only.  With Jk: d / dreg carry the 8
// calibration entries behind the points, every vector of the recurrence (x, r, z, p, y, qraw) has
// 7 C + 8 entries, minv 28 C + 36, part 2 (C + 1), sraw 35 C + 44 (the calibration block + its
// right-hand side behind the cameras'), ckpart [C][44] scratch.
extern "C" int iamx_ba_schur_prepare(const double *Jc, const double *Jp, const double *Jk,
                                     const double *r, const int32_t *cam_ptr, const int32_t *pt_idx,
                                     int64_t n_obs, int n_cams, int n_pts, const double *V,
                                     const double *gp, const double *d, const double *dreg, double *Y,
                                     double *yg, double *zp, double *sraw, double *ckpart,
                                     void *stream)
{
    IAMX_REQUIRE(Jc && Jp && r && cam_ptr && pt_idx && V && gp && d && dreg && Y && yg && zp && sraw,
                 "null pointer");
    IAMX_REQUIRE(!Jk || ckpart, "ckpart is required with calibration columns");
    IAMX_REQUIRE(n_obs >= 0 && n_cams > 0 && n_pts >= 0, "bad size");
    hipStream_t st = iamx::as_stream(stream);
    const double *d_p = d + (int64_t)n_cams * 7, *dreg_p = dreg + (int64_t)n_cams * 7;
    if (n_pts)
        hipLaunchKernelGGL(schur_points_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, st, V, gp,
                           d_p, dreg_p, n_pts, Y, yg, zp);
    if (Jk) {
        const double *d_k = d_p + (int64_t)n_pts * 3;
        hipLaunchKernelGGL((schur_adj_kernel<true, true>), dim3(n_cams), dim3(256), 0, st, Jc, Jp, Jk,
                           cam_ptr, pt_idx, r, zp, d, d_p, Y, (const double *)nullptr,
                           (const double *)nullptr, (const double *)nullptr, (double *)nullptr, sraw,
                           ckpart);
        hipLaunchKernelGGL(schur_calib_reduce_kernel<NK_TRI + NK>, dim3(1), dim3(256), 0, st,
                           (const double *)ckpart, n_cams, d_k, (const double *)nullptr,
                           (const double *)nullptr, (const double *)nullptr,
                           sraw + (int64_t)n_cams * 35, (double *)nullptr);
    } else {
        hipLaunchKernelGGL((schur_adj_kernel<true, false>), dim3(n_cams), dim3(256), 0, st, Jc, Jp,
                           (const double *)nullptr, cam_ptr, pt_idx, r, zp, d, d_p, Y,
                           (const double *)nullptr, (const double *)nullptr, (const double *)nullptr,
                           (double *)nullptr, sraw, (double *)nullptr);
    }
    return iamx::check_launch("iamx_ba_schur_prepare");
}

// n_pts: only used to find the calibration entries of d / dreg (with_calib != 0)
extern "C" int iamx_ba_schur_factor(const double *sraw, const double *d, const double *dreg,
                                    int n_cams, int n_pts, int with_calib, double eta, double qtol,
                                    int max_iter, double *minv, double *x, double *r, double *z,
                                    double *p, double *y, double *state, void *stream)
{
    IAMX_REQUIRE(sraw && d && dreg && minv && x && r && z && p && y && state, "null pointer");
    IAMX_REQUIRE(n_cams > 0 && n_pts >= 0 && eta >= 0 && qtol >= 0 && max_iter > 0, "bad size");
    const int64_t nk = (int64_t)n_cams * 7 + (int64_t)n_pts * 3;
    hipLaunchKernelGGL(schur_factor_kernel, dim3(1), dim3(1024), 0, iamx::as_stream(stream), sraw, d,
                       dreg, with_calib ? d + nk : (const double *)nullptr,
                       with_calib ? dreg + nk : (const double *)nullptr, n_cams, eta, qtol,
                       (double)max_iter, minv, x, r, z, p, y, state);
    return iamx::check_launch("iamx_ba_schur_factor");
}

extern "C" int iamx_ba_schur_iterate(const double *Jc, const double *Jp, const double *Jk,
                                     const int32_t *cam_idx, const int32_t *pt_idx,
                                     const int32_t *cam_ptr, const int32_t *pt_ptr,
                                     const int32_t *pt_obs, int64_t n_obs, int n_cams, int n_pts,
                                     const double *d, const double *dreg, const double *Y,
                                     const double *minv, double *t, double *zp, double *qraw,
                                     double *part, double *ckpart, double *x, double *r, double *z,
                                     double *p, double *y, double *state, int first_iter, int n_iter,
                                     int phase, void *stream)
{
    IAMX_REQUIRE(Jc && Jp && cam_idx && pt_idx && cam_ptr && pt_ptr && pt_obs && d && dreg && Y &&
                     minv && t && zp && qraw && part && x && r && z && p && y && state,
                 "null pointer");
    IAMX_REQUIRE(!Jk || ckpart, "ckpart is required with calibration columns");
    IAMX_REQUIRE(n_cams > 0 && n_pts >= 0 && n_obs >= 0 && n_iter >= 1 && first_iter >= 0, "bad size");
    IAMX_REQUIRE(phase >= -1 && phase <= 1, "phase is -1 (whole iterations), 0 or 1");
    hipStream_t st = iamx::as_stream(stream);
    const int64_t nc = (int64_t)n_cams * 7;
    const double *d_p = d + nc;
    const double *d_k = Jk ? d_p + (int64_t)n_pts * 3 : nullptr;
    const double *dreg_k = Jk ? dreg + nc + (int64_t)n_pts * 3 : nullptr;
    const int n_part = n_cams + (Jk ? 1 : 0);
    double *pqpart = part, *rzpart = part + n_part;
    const unsigned gc = (unsigned)((n_part + 255) / 256);
    for (int it = 0; it < n_iter; ++it) {
        const int par = (first_iter + it) & 1;
        const double *cur = state + par * ST_COUNT;
        double *next = state + (par ^ 1) * ST_COUNT;
        if (phase != 1) {
            if (n_obs)
                hipLaunchKernelGGL(schur_fwd_kernel, dim3(grid_for(n_obs)), dim3(256), 0, st, Jc, Jk,
                                   cam_idx, n_obs, (const double *)y, (const double *)(y + nc), cur, t);
            if (n_pts)
                hipLaunchKernelGGL(schur_pt_kernel<false>, dim3((n_pts + 255) / 256), dim3(256), 0,
                                   st, Jp, pt_ptr, pt_obs, n_pts, 0, n_pts, (const double *)t, d_p, Y,
                                   (const double *)nullptr, cur, zp);
            // one rank: q is complete, the adjoint kernel also emits the partials of p.q
            if (Jk) {
                hipLaunchKernelGGL((schur_adj_kernel<false, true>), dim3(n_cams), dim3(256), 0, st, Jc,
                                   Jp, Jk, cam_ptr, pt_idx, (const double *)t, (const double *)zp, d,
                                   d_p, Y, cur, (const double *)p, dreg,
                                   phase == -1 ? pqpart : (double *)nullptr, qraw, ckpart);
                hipLaunchKernelGGL(schur_calib_reduce_kernel<NK>, dim3(1), dim3(256), 0, st,
                                   (const double *)ckpart, n_cams, d_k, dreg_k,
                                   (const double *)(p + nc), cur, qraw + nc,
                                   phase == -1 ? pqpart + n_cams : (double *)nullptr);
            } else {
                hipLaunchKernelGGL((schur_adj_kernel<false, false>), dim3(n_cams), dim3(256), 0, st, Jc,
                                   Jp, (const double *)nullptr, cam_ptr, pt_idx, (const double *)t,
                                   (const double *)zp, d, d_p, Y, cur, (const double *)p, dreg,
                                   phase == -1 ? pqpart : (double *)nullptr, qraw, (double *)nullptr);
            }
        }
        if (phase != 0) {
            if (phase == 1)
                hipLaunchKernelGGL(schur_pq_kernel, dim3(gc), dim3(256), 0, st, (const double *)qraw,
                                   dreg, dreg_k, (const double *)p, n_cams, cur, pqpart);
            hipLaunchKernelGGL(schur_update1_kernel, dim3(gc), dim3(256), 0, st, (const double *)qraw,
                               dreg, dreg_k, minv, n_cams, x, r, z, (const double *)p, cur,
                               (const double *)pqpart, rzpart);
            hipLaunchKernelGGL(schur_update2_kernel, dim3(gc), dim3(256), 0, st, d, d_k, n_cams,
                               (const double *)z, p, y, cur, next, (const double *)pqpart,
                               (const double *)rzpart);
        }
    }
    return iamx::check_launch("iamx_ba_schur_iterate");
}

// step = (d_c .* x_c, point part by back-substitution, d_k .* x_k behind the points with Jk)
extern "C" int iamx_ba_schur_finish(const double *Jc, const double *Jp, const double *Jk,
                                    const int32_t *cam_idx, const int32_t *pt_ptr,
                                    const int32_t *pt_obs, int64_t n_obs, int n_cams, int n_pts,
                                    int pt_lo, int pt_hi, const double *d, const double *Y,
                                    const double *yg, const double *x, double *y, double *t,
                                    double *step, void *stream)
{
    IAMX_REQUIRE(Jc && Jp && cam_idx && pt_ptr && pt_obs && d && Y && yg && x && y && t && step,
                 "null pointer");
    IAMX_REQUIRE(n_cams > 0 && n_pts >= 0 && n_obs >= 0 && 0 <= pt_lo && pt_lo <= pt_hi &&
                     pt_hi <= n_pts, "bad size");
    hipStream_t st = iamx::as_stream(stream);
    const int64_t nc = (int64_t)n_cams * 7, np3 = (int64_t)n_pts * 3;
    hipLaunchKernelGGL(schur_scale_kernel, dim3(grid_for(nc)), dim3(256), 0, st, nc, d, x, y, step);
    if (Jk)
        hipLaunchKernelGGL(schur_scale_kernel, dim3(1), dim3(256), 0, st, (int64_t)NK, d + nc + np3,
                           x + nc, y + nc, step + nc + np3);
    if (n_obs)
        hipLaunchKernelGGL(schur_fwd_kernel, dim3(grid_for(n_obs)), dim3(256), 0, st, Jc, Jk, cam_idx,
                           n_obs, (const double *)y, (const double *)(y + nc), (const double *)nullptr, t);
    if (n_pts)
        hipLaunchKernelGGL(schur_pt_kernel<true>, dim3((n_pts + 255) / 256), dim3(256), 0, st, Jp,
                           pt_ptr, pt_obs, n_pts, pt_lo, pt_hi, (const double *)t,
                           d + nc, Y, yg, (const double *)nullptr, step + nc);
    return iamx::check_launch("iamx_ba_schur_finish");
}
