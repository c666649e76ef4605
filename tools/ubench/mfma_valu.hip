// micro-benchmark: how do i8 MFMAs and the epilogue's integer VALU ops co-issue on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o /tmp/mfma_valu ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NV, int MODE>   // NV = VALU ops per MFMA, MODE 0: dependent on nothing (register-only)
__global__ __launch_bounds__(1024) void k(int iters, int *out)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    v16i acc0 = {0}, acc1 = {0};
    int m1 = 0x7fffffff, m2 = 0x7fffffff, x = threadIdx.x, t = blockIdx.x;
    int n1 = 0x7fffffff, n2 = 0x7fffffff;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE != 2) {
                if (u & 1) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc1, 0, 0, 0);
                else       acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < NV; v += 3) {
                // the epilogue triple, on values that do not depend on the MFMA
                int key;
                if (MODE == 3) {   // two independent chains
                    asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(key) : "v"(x), "v"(t));
                    if ((v / 3) & 1) { asm volatile("v_med3_i32 %0, %1, %0, %2" : "+v"(m2) : "v"(m1), "v"(key));
                                       asm volatile("v_min_i32 %0, %0, %1" : "+v"(m1) : "v"(key)); }
                    else             { asm volatile("v_med3_i32 %0, %1, %0, %2" : "+v"(n2) : "v"(n1), "v"(key));
                                       asm volatile("v_min_i32 %0, %0, %1" : "+v"(n1) : "v"(key)); }
                } else {
                    asm volatile("v_lshl_add_u32 %0, %1, 9, %2" : "=v"(key) : "v"(x), "v"(t));
                    asm volatile("v_med3_i32 %0, %1, %0, %2" : "+v"(m2) : "v"(m1), "v"(key));
                    asm volatile("v_min_i32 %0, %0, %1" : "+v"(m1) : "v"(key));
                }
            }
        }
    }
    int s = m1 + m2 + n1 + n2;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 0x12345678) out[0] = s;
}

template <int NV, int MODE>
void run(const char *name, int threads, int blocks_per_cu)
{
    int *out; hipMalloc(&out, 4);
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 g(256 * blocks_per_cu), b(threads);
    hipLaunchKernelGGL((k<NV, MODE>), g, b, 0, 0, 100, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, MODE>), g, b, 0, 0, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int waves_per_simd = threads / 256 * blocks_per_cu;
    double ns_per_mfma_slot = ms * 1e6 / (iters * 4.0) / waves_per_simd;   // per SIMD per MFMA-slot
    printf("%-34s thr=%4d blk/CU=%d waves/SIMD=%d  NV=%2d : %8.3f ms  %.2f ns per (MFMA+NV VALU) per SIMD\n",
           name, threads, blocks_per_cu, waves_per_simd, NV, ms, ns_per_mfma_slot);
    hipFree(out);
}

int main()
{
    for (int w = 1; w <= 4; ++w) {
        int thr = 256 * (w > 2 ? 2 : w), bpc = (w > 2 ? w / 2 : 1);
        if (w == 3) { thr = 768; bpc = 1; }
        run<0, 0>("mfma only", thr, bpc);
        run<15, 2>("15 valu only", thr, bpc);
        run<6, 0>("mfma + 6 valu", thr, bpc);
        run<9, 0>("mfma + 9 valu", thr, bpc);
        run<12, 0>("mfma + 12 valu", thr, bpc);
        run<15, 0>("mfma + 15 valu", thr, bpc);
        run<15, 3>("mfma + 15 valu (2 chains)", thr, bpc);
        run<15, 2>("15 valu only", thr, bpc);
        run<30, 2>("30 valu only", thr, bpc);
    }
    return 0;
}
