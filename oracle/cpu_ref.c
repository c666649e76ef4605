/* ORACLE / TEST INFRASTRUCTURE ONLY -- plain-C CPU restatement of the two
 * arithmetic kernels of the hot path, used (a) by tests/ to check the HIP path
 * at sizes numpy is too slow for, and (b) by bench.py's cpu_baseline leg
 * ("kind": "port").  Never linked or loaded by the product (imageanalysis_amd/).
 * Checked against tests/golden/ through oracle/match_oracle.py / ba_oracle.py
 * in tests/test_oracle.py.  Citations are relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define D 128

/* scripts/lib/matcher.py:203-216 raw_matches -> knnMatch(des1, des2, k=2) restated as
 * exact brute force (cv2.BFMatcher(NORM_L2) semantics): squared L2 in int32,
 * 2 nearest train rows per query row, ties -> lowest train index. */
__attribute__((target_clones("avx2", "default")))
static void knn2_rows(const uint8_t *q, int q0, int q1, const uint8_t *t, int nt,
                      int32_t *idx, int32_t *d2)
{
    for (int i = q0; i < q1; ++i) {
        const uint8_t *a = q + (size_t)i * D;
        int32_t b0 = INT32_MAX, b1 = INT32_MAX, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; ++j) {
            const uint8_t *b = t + (size_t)j * D;
            int32_t s = 0;
            for (int k = 0; k < D; ++k) {
                int32_t d = (int32_t)a[k] - (int32_t)b[k];
                s += d * d;
            }
            if (s < b0) { b1 = b0; i1 = i0; b0 = s; i0 = j; }
            else if (s < b1) { b1 = s; i1 = j; }
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        d2[2 * i] = b0;  d2[2 * i + 1] = b1;
    }
}

int oracle_knn2_l2_u8(const uint8_t *q, int nq, const uint8_t *t, int nt,
                      int32_t *idx, int32_t *d2, int nthreads)
{
    if (nt < 2) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < nq; i += 16)
        knn2_rows(q, i, i + 16 < nq ? i + 16 : nq, t, nt, idx, d2);
#else
    (void)nthreads;
    knn2_rows(q, 0, nq, t, nt, idx, d2);
#endif
    return 0;
}

/* Batched form for the CPU baseline: many ordered (query image, train image) pairs of equal
 * size in ONE parallel region (work item = 64 query rows of one ordered pair), so that all host
 * cores stay busy the way a tuned CPU matcher would keep them. */
int oracle_knn2_l2_u8_batch(const uint8_t *images, int n_rows, const int32_t *pairs, int n_pairs,
                            int32_t *idx, int32_t *d2, int nthreads)
{
    if (n_rows < 2) return -1;
    const int blocks = (n_rows + 63) / 64;
    const long items = (long)n_pairs * blocks;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (long it = 0; it < items; ++it) {
        const int p = (int)(it / blocks), b = (int)(it % blocks);
        const uint8_t *q = images + (size_t)pairs[2 * p] * n_rows * D;
        const uint8_t *t = images + (size_t)pairs[2 * p + 1] * n_rows * D;
        const int r0 = b * 64, r1 = r0 + 64 < n_rows ? r0 + 64 : n_rows;
        knn2_rows(q, r0, r1, t, n_rows, idx + (size_t)p * n_rows * 2, d2 + (size_t)p * n_rows * 2);
    }
    return 0;
}

/* (a container's CPU quota is not what omp_get_max_threads() reports: the caller sets it) */
void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* scripts/lib/optimizer.py:174-229 Optimizer.fun + :120-126 nedquat2rvectvec +
 * scripts/lib/archive/transformations.py:1395-1420 quaternion_matrix +
 * scripts/lib/project.py:300-329 (Brown model cv2.projectPoints applies).
 * cams C x 7 (ned, quat wxyz); pts P x 3; obs camera-major; r = observed - projected. */
static void cam_rt(const double *c, double R[9], double t[3])
{
    double w = c[3], x = c[4], y = c[5], z = c[6];
    double n = w * w + x * x + y * y + z * z;
    double B[9];                               /* body2ned */
    if (n < 2.220446049250313e-16 * 4.0) {
        B[0] = 1; B[1] = 0; B[2] = 0; B[3] = 0; B[4] = 1; B[5] = 0; B[6] = 0; B[7] = 0; B[8] = 1;
    } else {
        double s = sqrt(2.0 / n);
        w *= s; x *= s; y *= s; z *= s;
        B[0] = 1.0 - y * y - z * z; B[1] = x * y - z * w;       B[2] = x * z + y * w;
        B[3] = x * y + z * w;       B[4] = 1.0 - x * x - z * z; B[5] = y * z - x * w;
        B[6] = x * z - y * w;       B[7] = y * z + x * w;       B[8] = 1.0 - x * x - y * y;
    }
    /* R = body2cam . body2ned^T ; body2cam = inv([[0,0,1],[1,0,0],[0,1,0]]) = [[0,1,0],[0,0,1],[1,0,0]]
     * => row0(R) = col1(B)^T... : R[i][j] = sum_k body2cam[i][k] * B[j][k] */
    for (int j = 0; j < 3; ++j) {
        R[0 * 3 + j] = B[j * 3 + 1];
        R[1 * 3 + j] = B[j * 3 + 2];
        R[2 * 3 + j] = B[j * 3 + 0];
    }
    for (int i = 0; i < 3; ++i)
        t[i] = -(R[i * 3] * c[0] + R[i * 3 + 1] * c[1] + R[i * 3 + 2] * c[2]);
}

int oracle_ba_residual(const double *cams, int n_cams, const double *pts, int n_pts,
                       const int32_t *cam_idx, const int32_t *pt_idx, const double *uv,
                       int64_t n_obs, const double *intr /* fx fy cu cv */,
                       const double *dist /* k1 k2 p1 p2 k3 */, double *r, int nthreads)
{
    (void)n_pts;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t o = 0; o < n_obs; ++o) {
        int c = cam_idx[o];
        if (c < 0 || c >= n_cams) continue;
        double R[9], t[3];
        cam_rt(cams + (size_t)c * 7, R, t);
        const double *X = pts + (size_t)pt_idx[o] * 3;
        double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
        double yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
        double zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
        double x = xc / zc, y = yc / zc;
        double r2 = x * x + y * y;
        double rad = 1.0 + dist[0] * r2 + dist[1] * r2 * r2 + dist[4] * r2 * r2 * r2;
        double xd = x * rad + 2.0 * dist[2] * x * y + dist[3] * (r2 + 2.0 * x * x);
        double yd = y * rad + dist[2] * (r2 + 2.0 * y * y) + 2.0 * dist[3] * x * y;
        r[2 * o]     = uv[2 * o]     - (intr[0] * xd + intr[2]);
        r[2 * o + 1] = uv[2 * o + 1] - (intr[1] * yd + intr[3]);
    }
    return 0;
}
