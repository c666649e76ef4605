// micro-benchmark: what bounds the BA residual kernel (1.96 M observations, 64 B/obs algorithmic)?
// variants: 0 = shipped form (2 obs/thread, scalar camera block), 1 = no point gather (sequential
// points), 2 = no camera block loads, 3 = 4 obs/thread, 4 = pure stream (uv -> r copy + idx reads)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__device__ __forceinline__ double2 resid(const double *R, const double *X, double2 obs, const double *cal)
{
    const double a = X[0] - R[9], b = X[1] - R[10], c = X[2] - R[11];
    const double y0 = R[0] * a + R[1] * b + R[2] * c, y1 = R[3] * a + R[4] * b + R[5] * c,
                 y2 = R[6] * a + R[7] * b + R[8] * c;
    const double iz = 1.0 / y0, px = y1 * iz, py = y2 * iz, r2 = px * px + py * py;
    const double rad = 1.0 + r2 * (cal[4] + r2 * (cal[5] + r2 * cal[8]));
    const double xd = px * rad + 2.0 * cal[6] * px * py + cal[7] * (r2 + 2.0 * px * px);
    const double yd = py * rad + cal[6] * (r2 + 2.0 * py * py) + 2.0 * cal[7] * px * py;
    return make_double2(obs.x - (cal[0] * xd + cal[2]), obs.y - (cal[1] * yd + cal[3]));
}

template <int V, int NPT>
__global__ __launch_bounds__(256) void k(const double *rt, const double *pts, const int *cam, const int *pt,
                                         const double *uv, long n, const double *calib, double *r)
{
    const long o0 = ((long)blockIdx.x * 256 + threadIdx.x) * NPT;
    if (o0 + NPT > n) return;
    double cal[9];
    for (int i = 0; i < 9; ++i) cal[i] = calib[i];
#pragma unroll
    for (int j = 0; j < NPT; j += 2) {
        const long o = o0 + j;
        const int2 ci = *(const int2 *)(cam + o);
        int2 pi = *(const int2 *)(pt + o);
        const double4 ob = *(const double4 *)(uv + 2 * o);
        if (V == 4) { *(double4 *)(r + 2 * o) = make_double4(ob.x + ci.x, ob.y + pi.x, ob.z, ob.w); continue; }
        if (V == 1) { pi.x = (int)(o % 200000); pi.y = pi.x + 1; }
        const int c0 = (V == 2) ? 0 : __builtin_amdgcn_readfirstlane(ci.x);
        const double *R = rt + (long)c0 * 12;
        const double2 r0 = resid(R, pts + (long)pi.x * 3, make_double2(ob.x, ob.y), cal);
        const double2 r1 = resid(R, pts + (long)pi.y * 3, make_double2(ob.z, ob.w), cal);
        *(double4 *)(r + 2 * o) = make_double4(r0.x, r0.y, r1.x, r1.y);
    }
}

template <int V, int NPT>
void run(const char *name, const double *rt, const double *pts, const int *cam, const int *pt, const double *uv,
         long n, const double *cal, double *r)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned g = (unsigned)((n / NPT + 255) / 256);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<V, NPT>), dim3(g), dim3(256), 0, 0, rt, pts, cam, pt, uv, n, cal, r);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<V, NPT>), dim3(g), dim3(256), 0, 0, rt, pts, cam, pt, uv, n, cal, r);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.2f us  %6.2f TB/s (64 B/obs)\n", name, ms * 20, 64.0 * n / (ms / 50 * 1e-3) / 1e12);
}

int main()
{
    const long n = 1960896; const int C = 2812, P = 271555;
    std::vector<int> cam(n), pt(n); std::vector<double> uv(2 * n), rt(C * 12), pts(P * 3), cal = {3666, 3666, 2736, 1824, 0, 0, 0, 0, 0};
    srand(1);
    for (long i = 0; i < n; ++i) { cam[i] = (int)(i * C / n); pt[i] = (int)(((long)cam[i] * 96 + rand() % 4000) % P); uv[2 * i] = rand() % 5000; uv[2 * i + 1] = rand() % 3000; }
    for (auto &v : rt) v = (rand() % 1000) / 1000.0 + 0.1;
    for (auto &v : pts) v = (rand() % 1000) / 10.0;
    int *dc, *dp; double *duv, *drt, *dpts, *dcal, *dr;
    (void)hipMalloc(&dc, n * 4); (void)hipMalloc(&dp, n * 4); (void)hipMalloc(&duv, n * 16); (void)hipMalloc(&dr, n * 16);
    (void)hipMalloc(&drt, C * 96); (void)hipMalloc(&dpts, P * 24); (void)hipMalloc(&dcal, 72);
    (void)hipMemcpy(dc, cam.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dp, pt.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(duv, uv.data(), n * 16, hipMemcpyHostToDevice); (void)hipMemcpy(drt, rt.data(), C * 96, hipMemcpyHostToDevice);
    (void)hipMemcpy(dpts, pts.data(), P * 24, hipMemcpyHostToDevice); (void)hipMemcpy(dcal, cal.data(), 72, hipMemcpyHostToDevice);
    run<0, 2>("shipped form: 2 obs/thread", drt, dpts, dc, dp, duv, n, dcal, dr);
    run<1, 2>("sequential points (no random gather)", drt, dpts, dc, dp, duv, n, dcal, dr);
    run<2, 2>("camera 0 for everyone", drt, dpts, dc, dp, duv, n, dcal, dr);
    run<0, 4>("4 obs/thread", drt, dpts, dc, dp, duv, n, dcal, dr);
    run<0, 8>("8 obs/thread", drt, dpts, dc, dp, duv, n, dcal, dr);
    run<4, 2>("pure stream (idx + uv -> r), 2 obs/thread", drt, dpts, dc, dp, duv, n, dcal, dr);
    run<4, 4>("pure stream, 4 obs/thread", drt, dpts, dc, dp, duv, n, dcal, dr);
    return 0;
}
