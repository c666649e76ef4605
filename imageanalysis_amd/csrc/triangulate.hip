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
