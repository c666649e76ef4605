// Initial feature positions by ray / ground-plane intersection (gfx950) --
// scripts/lib/match_cleanup.py:320-347 triangulate_smart (per feature, per observation):
//     v = unit_vector(body2ned . cam2body . IK . [u, v, 1])        (lib/project.py:540-548)
//     if v[2] > 0:  d = -(ned[2] + base_elev);  p = ned + [v0*d/v2, v1*d/v2, d];  sum += p
//     match[0] = sum / (number of observations of the feature)     (sky rays count in the divisor)
// One thread per feature, its observations added in stored order (f64, separately rounded
// products and sums in numpy's evaluation order).  The per-image 3x3 matrix
// M = (body2ned . cam2body) . IK is formed on the host with numpy, exactly as the reference
// nests the products.
#include "iamx_common.h"

namespace {

__global__ __launch_bounds__(256) void triangulate_ground_kernel(
    const double *__restrict__ M, const double *__restrict__ ned, const double *__restrict__ base_elev,
    const int32_t *__restrict__ obs_img, const double *__restrict__ obs_uv,
    const int64_t *__restrict__ feat_ptr, int64_t n_feat, double *__restrict__ out,
    int32_t *__restrict__ n_sky)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_feat) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    int sky = 0;
    const int64_t b = feat_ptr[f], e = feat_ptr[f + 1];
    for (int64_t o = b; o < e; ++o) {
        const int im = obs_img[o];
        const double *m = M + (int64_t)im * 9, *c = ned + (int64_t)im * 3;
        const double u = obs_uv[2 * o], v = obs_uv[2 * o + 1];
        // M . [u, v, 1]: (m0*u + m1*v) + m2*1
        const double p0 = __dadd_rn(__dadd_rn(__dmul_rn(m[0], u), __dmul_rn(m[1], v)), m[2]);
        const double p1 = __dadd_rn(__dadd_rn(__dmul_rn(m[3], u), __dmul_rn(m[4], v)), m[5]);
        const double p2 = __dadd_rn(__dadd_rn(__dmul_rn(m[6], u), __dmul_rn(m[7], v)), m[8]);
        const double nrm = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(p0, p0), __dmul_rn(p1, p1)), __dmul_rn(p2, p2)));
        const double v0 = p0 / nrm, v1 = p1 / nrm, v2 = p2 / nrm;
        if (v2 > 0.0) {
            const double d = -(__dadd_rn(c[2], base_elev[im]));
            const double factor = d / v2;
            s0 = __dadd_rn(s0, __dadd_rn(c[0], __dmul_rn(v0, factor)));
            s1 = __dadd_rn(s1, __dadd_rn(c[1], __dmul_rn(v1, factor)));
            s2 = __dadd_rn(s2, __dadd_rn(c[2], d));
        } else {
            ++sky;
        }
    }
    const double cnt = (double)(e - b);
    out[3 * f] = s0 / cnt;
    out[3 * f + 1] = s1 / cnt;
    out[3 * f + 2] = s2 / cnt;
    if (sky) atomicAdd(n_sky, sky);
}

}  // namespace

extern "C" int iamx_triangulate_ground(const double *M, const double *ned, const double *base_elev,
                                       int n_images, const int32_t *obs_img, const double *obs_uv,
                                       const int64_t *feat_ptr, int64_t n_feat, double *out_ned,
                                       int32_t *n_sky, void *stream)
{
    IAMX_REQUIRE(M && ned && base_elev && obs_img && obs_uv && feat_ptr && out_ned && n_sky,
                 "null pointer");
    IAMX_REQUIRE(n_images > 0 && n_feat >= 0, "bad size");
    if (n_feat == 0) return IAMX_OK;
    hipLaunchKernelGGL(triangulate_ground_kernel, dim3((unsigned)((n_feat + 255) / 256)), dim3(256), 0,
                       iamx::as_stream(stream), M, ned, base_elev, obs_img, obs_uv, feat_ptr, n_feat,
                       out_ned, n_sky);
    return iamx::check_launch("iamx_triangulate_ground");
}

// ---------------------------------------------------------------------------------
// Two-view linear (DLT) triangulation of matched keypoints -- scripts/lib/smart.py:26-63
// triangulate_features(): cv2.triangulatePoints(PROJ1, PROJ2, IK.uv1, IK.uv2), then / w.
// Per match the 4x4 system  [x1*P1_3 - P1_1; y1*P1_3 - P1_2; x2*P2_3 - P2_1; y2*P2_3 - P2_2]
// whose null direction (right singular vector of the smallest singular value) is the point;
// found by one-sided Jacobi (Hestenes) rotations on the columns, f64.  One thread per match;
// all pairs of a batch in one launch.
//   pair_img [n_pairs][2] image slots, PROJ [n_images][12] = [R | t] row major,
//   IK [9] inverse camera matrix, match lists as iamx_match_postfilter leaves them.
// out_z [n_pairs][clip]: NED "down" of every triangulated match (w-normalised).
// ---------------------------------------------------------------------------------
namespace {

__device__ void null_vector_4x4(double A[4][4], double v_out[4])
{
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 12; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    alpha += A[i][p] * A[i][p];
                    beta += A[i][q] * A[i][q];
                    gamma += A[i][p] * A[i][q];
                }
                off = fmax(off, fabs(gamma) / sqrt(fmax(alpha * beta, 1e-300)));
                if (fabs(gamma) < 1e-300) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double ap = A[i][p], aq = A[i][q];
                    A[i][p] = c * ap - s * aq;
                    A[i][q] = s * ap + c * aq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq;
                    V[i][q] = s * vp + c * vq;
                }
            }
        }
        if (off < 1e-15) break;
    }
    int best = 0;
    double bn = 1e300;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double nrm = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) nrm += A[i][j] * A[i][j];
        if (nrm < bn) { bn = nrm; best = j; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double v = V[i][0];
        if (best == 1) v = V[i][1];
        if (best == 2) v = V[i][2];
        if (best == 3) v = V[i][3];
        v_out[i] = v;
    }
}

template <bool XYZ>
__global__ __launch_bounds__(256) void triangulate_pairs_kernel(
    const int32_t *__restrict__ pair_img, const double *__restrict__ PROJ, const double *__restrict__ IK,
    const int64_t *__restrict__ kp_off, const float *__restrict__ xy,
    const int32_t *__restrict__ m_cnt, const int32_t *__restrict__ m_pairs, int clip,
    double *__restrict__ out_z)
{
    const int p = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= m_cnt[p]) return;
    const int im1 = pair_img[2 * p], im2 = pair_img[2 * p + 1];
    const int q = m_pairs[((int64_t)p * clip + k) * 2], t = m_pairs[((int64_t)p * clip + k) * 2 + 1];
    const float *p1 = xy + 2 * (kp_off[im1] + q), *p2 = xy + 2 * (kp_off[im2] + t);
    const double uv[2][2] = {{(double)p1[0], (double)p1[1]}, {(double)p2[0], (double)p2[1]}};
    const double *P[2] = {PROJ + (int64_t)im1 * 12, PROJ + (int64_t)im2 * 12};
    double A[4][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        // normalised image point: IK . [u, v, 1], first two components
        const double x = IK[0] * uv[j][0] + IK[1] * uv[j][1] + IK[2];
        const double y = IK[3] * uv[j][0] + IK[4] * uv[j][1] + IK[5];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            A[2 * j][c] = x * P[j][8 + c] - P[j][c];
            A[2 * j + 1][c] = y * P[j][8 + c] - P[j][4 + c];
        }
    }
    double X[4];
    null_vector_4x4(A, X);
    if (XYZ) {
        // (points /= points[3] of smart.py:62: north, east, down)
        double *o = out_z + ((int64_t)p * clip + k) * 3;
        o[0] = X[0] / X[3];
        o[1] = X[1] / X[3];
        o[2] = X[2] / X[3];
    } else {
        out_z[(int64_t)p * clip + k] = X[2] / X[3];
    }
}

// The same triangulation over PACKED match lists with one pair of projection matrices PER PAIR:
// find_matches' surface stage.  The reference rewrites both images' camera poses after every
// pair (scripts/lib/matcher.py:990-993 -> lib/image.py:434-457), so the matrices a pair
// triangulates with depend on the pairs before it; the host replays that chain
// (smart.PoseFeedback) and hands every pair its own [R | t] x 2.
//   pair_img [n_pairs][2] image slots (keypoint arena), pair_proj [n_pairs][2][12],
//   m_off [n_pairs + 1] first match of every pair in m_pairs [total][2]; out_z [total].
__global__ __launch_bounds__(256) void triangulate_packed_kernel(
    const int32_t *__restrict__ pair_img, const double *__restrict__ pair_proj,
    const double *__restrict__ IK, const int64_t *__restrict__ kp_off, const float *__restrict__ xy,
    const int64_t *__restrict__ m_off, const int32_t *__restrict__ m_pairs, int n_pairs,
    int64_t total, double *__restrict__ out_z)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= total) return;
    // the pair this match belongs to: last p with m_off[p] <= k
    int lo = 0, hi = n_pairs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (m_off[mid] <= k) lo = mid; else hi = mid - 1;
    }
    const int p = lo;
    const int im1 = pair_img[2 * p], im2 = pair_img[2 * p + 1];
    const int q = m_pairs[2 * k], t = m_pairs[2 * k + 1];
    const float *p1 = xy + 2 * (kp_off[im1] + q), *p2 = xy + 2 * (kp_off[im2] + t);
    const double uv[2][2] = {{(double)p1[0], (double)p1[1]}, {(double)p2[0], (double)p2[1]}};
    const double *P[2] = {pair_proj + (int64_t)p * 24, pair_proj + (int64_t)p * 24 + 12};
    double A[4][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double x = IK[0] * uv[j][0] + IK[1] * uv[j][1] + IK[2];
        const double y = IK[3] * uv[j][0] + IK[4] * uv[j][1] + IK[5];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            A[2 * j][c] = x * P[j][8 + c] - P[j][c];
            A[2 * j + 1][c] = y * P[j][8 + c] - P[j][4 + c];
        }
    }
    double X[4];
    null_vector_4x4(A, X);
    out_z[k] = X[2] / X[3];
}

// ---------------------------------------------------------------------------------
// 4-DOF similarity (rotation, uniform scale, translation) between the matched keypoints of an
// image pair -- the matrix the reference asks cv2.estimateAffinePartial2D for
// (scripts/lib/smart.py:66-89 find_affine), with a DETERMINISTIC robust fit in place of
// OpenCV's RANSAC: least squares on all matches, then nine re-fits on the matches whose
// residual is at most 200, 50, 10, 3, 3, ... px under the current model.
// grid = (pairs, 2): direction 0 maps image b's pixels onto image a's (find_affine(a, b)),
// direction 1 the other way (find_affine(b, a) on the mirrored list).  Fixed reduction trees.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void block_sum5(double (&v)[5], double (*sh)[5])
{
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v[k] += __shfl_xor(v[k], m);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) sh[threadIdx.x >> 6][k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = sh[0][k] + sh[1][k] + sh[2][k] + sh[3][k];
}

__global__ __launch_bounds__(256) void similarity_pairs_kernel(
    const int32_t *__restrict__ pair_img, const int64_t *__restrict__ kp_off,
    const float *__restrict__ xy, const int32_t *__restrict__ m_cnt,
    const int32_t *__restrict__ m_pairs, int clip, double *__restrict__ out_aff,
    int32_t *__restrict__ out_ok)
{
    __shared__ double sh[4][5];
    const int p = blockIdx.x, dir = blockIdx.y;
    const int n = m_cnt[p];
    const int ima = pair_img[2 * p], imb = pair_img[2 * p + 1];
    const float *xa = xy + 2 * kp_off[ima], *xb = xy + 2 * kp_off[imb];
    const int32_t *mp = m_pairs + (int64_t)p * clip * 2;
    double M[6] = {0, 0, 0, 0, 0, 0};
    bool have = false;
    const double thr[10] = {-1.0, 200.0, 50.0, 10.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0};
    for (int it = 0; it < 10; ++it) {
        const double t2 = thr[it] * thr[it];
        auto weight = [&](double px, double py, double qx, double qy) {
            if (it == 0) return true;
            const double rx = M[0] * px + M[1] * py + M[2] - qx, ry = M[3] * px + M[4] * py + M[5] - qy;
            return sqrt(rx * rx + ry * ry) <= thr[it];
        };
        (void)t2;
        // pass 1: count and centroids of the matches in the fit
        double s[5] = {0, 0, 0, 0, 0};
        for (int k = threadIdx.x; k < n; k += 256) {
            const float *a = xa + 2 * mp[2 * k], *b = xb + 2 * mp[2 * k + 1];
            const double px = dir == 0 ? b[0] : a[0], py = dir == 0 ? b[1] : a[1];      // from
            const double qx = dir == 0 ? a[0] : b[0], qy = dir == 0 ? a[1] : b[1];      // to
            if (weight(px, py, qx, qy)) { s[0] += 1.0; s[1] += px; s[2] += py; s[3] += qx; s[4] += qy; }
        }
        block_sum5(s, sh);
        if (s[0] < 2.0) break;
        const double cnt = s[0], cpx = s[1] / cnt, cpy = s[2] / cnt, cqx = s[3] / cnt, cqy = s[4] / cnt;
        // pass 2: centred second moments
        double c[5] = {0, 0, 0, 0, 0};
        for (int k = threadIdx.x; k < n; k += 256) {
            const float *a = xa + 2 * mp[2 * k], *b = xb + 2 * mp[2 * k + 1];
            const double px = dir == 0 ? b[0] : a[0], py = dir == 0 ? b[1] : a[1];
            const double qx = dir == 0 ? a[0] : b[0], qy = dir == 0 ? a[1] : b[1];
            if (weight(px, py, qx, qy)) {
                const double ux = px - cpx, uy = py - cpy, vx = qx - cqx, vy = qy - cqy;
                c[0] += ux * ux + uy * uy;
                c[1] += ux * vx + uy * vy;
                c[2] += ux * vy - uy * vx;
            }
        }
        block_sum5(c, sh);
        if (c[0] == 0.0) break;
        const double a_ = c[1] / c[0], b_ = c[2] / c[0];
        M[0] = a_; M[1] = -b_; M[2] = cqx - (a_ * cpx - b_ * cpy);
        M[3] = b_; M[4] = a_;  M[5] = cqy - (b_ * cpx + a_ * cpy);
        have = true;
    }
    if (threadIdx.x == 0) {
        double *o = out_aff + ((int64_t)p * 2 + dir) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = M[k];
        out_ok[p * 2 + dir] = have ? 1 : 0;
    }
}

}  // namespace

extern "C" int iamx_similarity_pairs(const int32_t *pair_img, const int64_t *kp_off, const float *xy,
                                     const int32_t *m_cnt, const int32_t *m_pairs, int n_pairs,
                                     int clip, double *out_aff, int32_t *out_ok, void *stream)
{
    IAMX_REQUIRE(pair_img && kp_off && xy && m_cnt && m_pairs && out_aff && out_ok, "null pointer");
    IAMX_REQUIRE(n_pairs >= 0 && clip > 0, "bad size");
    if (n_pairs == 0) return IAMX_OK;
    hipLaunchKernelGGL(similarity_pairs_kernel, dim3((unsigned)n_pairs, 2), dim3(256), 0,
                       iamx::as_stream(stream), pair_img, kp_off, xy, m_cnt, m_pairs, clip, out_aff,
                       out_ok);
    return iamx::check_launch("iamx_similarity_pairs");
}

extern "C" int iamx_triangulate_pairs(const int32_t *pair_img, const double *PROJ, const double *IK,
                                      const int64_t *kp_off, const float *xy, const int32_t *m_cnt,
                                      const int32_t *m_pairs, int n_pairs, int clip, double *out_z,
                                      void *stream)
{
    IAMX_REQUIRE(pair_img && PROJ && IK && kp_off && xy && m_cnt && m_pairs && out_z, "null pointer");
    IAMX_REQUIRE(n_pairs >= 0 && clip > 0, "bad size");
    if (n_pairs == 0) return IAMX_OK;
    hipLaunchKernelGGL(triangulate_pairs_kernel<false>, dim3((unsigned)((clip + 255) / 256), (unsigned)n_pairs),
                       dim3(256), 0, iamx::as_stream(stream), pair_img, PROJ, IK, kp_off, xy, m_cnt,
                       m_pairs, clip, out_z);
    return iamx::check_launch("iamx_triangulate_pairs");
}

extern "C" int iamx_triangulate_pairs_xyz(const int32_t *pair_img, const double *PROJ, const double *IK,
                                          const int64_t *kp_off, const float *xy, const int32_t *m_cnt,
                                          const int32_t *m_pairs, int n_pairs, int clip,
                                          double *out_xyz, void *stream)
{
    IAMX_REQUIRE(pair_img && PROJ && IK && kp_off && xy && m_cnt && m_pairs && out_xyz, "null pointer");
    IAMX_REQUIRE(n_pairs >= 0 && clip > 0, "bad size");
    if (n_pairs == 0) return IAMX_OK;
    hipLaunchKernelGGL(triangulate_pairs_kernel<true>, dim3((unsigned)((clip + 255) / 256), (unsigned)n_pairs),
                       dim3(256), 0, iamx::as_stream(stream), pair_img, PROJ, IK, kp_off, xy, m_cnt,
                       m_pairs, clip, out_xyz);
    return iamx::check_launch("iamx_triangulate_pairs_xyz");
}

extern "C" int iamx_triangulate_packed(const int32_t *pair_img, const double *pair_proj, const double *IK,
                                       const int64_t *kp_off, const float *xy, const int64_t *m_off,
                                       const int32_t *m_pairs, int n_pairs, int64_t total,
                                       double *out_z, void *stream)
{
    IAMX_REQUIRE(pair_img && pair_proj && IK && kp_off && xy && m_off && m_pairs && out_z, "null pointer");
    IAMX_REQUIRE(n_pairs >= 0 && total >= 0, "bad size");
    if (n_pairs == 0 || total == 0) return IAMX_OK;
    hipLaunchKernelGGL(triangulate_packed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       iamx::as_stream(stream), pair_img, pair_proj, IK, kp_off, xy, m_off, m_pairs,
                       n_pairs, total, out_z);
    return iamx::check_launch("iamx_triangulate_packed");
}
