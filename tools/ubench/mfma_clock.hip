// micro-benchmark: sustained v_mfma_i32_32x32x32_i8 rate and shader clock on gfx950 for
//   (a) independent accumulators vs dependent chains of 4, (b) zero vs random operand data,
//   (c) 2 vs 4 waves per SIMD.  clock = s_memtime ticks / s_memrealtime (100 MHz) ticks.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: 2 independent accumulators  1: dependent chain of 4 then new C
__global__ __launch_bounds__(256) void k(const v4i *data, int iters, long long *clk, int *sink)
{
    v4i a[4], b[4];
    for (int s = 0; s < 4; ++s) {
        a[s] = data[(blockIdx.x * 256 + threadIdx.x) * 8 + s];
        b[s] = data[(blockIdx.x * 256 + threadIdx.x) * 8 + 4 + s];
    }
    v16i acc0 = {0}, acc1 = {0};
    v16i c0 = {0};
    long long t0 = clock64(), w0 = wall_clock64();
    int keep = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[s], a[s], acc1, 0, 0, 0);
            }
        } else {
            v16i x = c0, y = c0;
#pragma unroll
            for (int s = 0; s < 4; ++s) x = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[s], x, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) y = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[s], a[s], y, 0, 0, 0);
            keep ^= x[0] ^ y[5];
            a[0][0] += 1;          // new operands every round (no CSE)
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
    int s = keep;
    for (int q = 0; q < 16; ++q) s += acc0[q] + acc1[q];
    if (s == 0x12345678) sink[0] = s;
}

template <int MODE>
void run(const char *name, const v4i *d, int blocks_per_cu)
{
    long long *clk; int *sink;
    const int nb = 256 * blocks_per_cu;
    (void)hipMalloc(&clk, nb * 16); (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;   // 1.6 M MFMAs per wave, ~30-50 ms
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, d, 1000, clk, sink);
    (void)hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, d, iters, clk, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double mfma_per_simd = (double)iters * 8 * blocks_per_cu;
    const double ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;
    printf("%-44s %d waves/SIMD: %7.2f ms  %.2f ns/MFMA/SIMD  shader clock %.3f GHz  => %.1f cycles/MFMA\n",
           name, blocks_per_cu, ms, ms * 1e6 / mfma_per_simd, ghz, ms * 1e6 / mfma_per_simd * ghz);
    (void)hipFree(clk); (void)hipFree(sink);
}

int main()
{
    const size_t n = 256 * 4 * 256 * 8;
    v4i *h = (v4i *)malloc(n * sizeof(v4i)), *dz, *dr;
    (void)hipMalloc(&dz, n * sizeof(v4i)); (void)hipMalloc(&dr, n * sizeof(v4i));
    (void)hipMemset(dz, 0, n * sizeof(v4i));
    srand(1);
    for (size_t i = 0; i < n * 4; ++i) ((int *)h)[i] = (rand() << 16) ^ rand();
    (void)hipMemcpy(dr, h, n * sizeof(v4i), hipMemcpyHostToDevice);
    for (int w : {2, 4}) {
        run<0>("independent accumulators, zero data", dz, w);
        run<0>("independent accumulators, random data", dr, w);
        run<1>("chains of 4 (fresh C), zero data", dz, w);
        run<1>("chains of 4 (fresh C), random data", dr, w);
    }
    return 0;
}
