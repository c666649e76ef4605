// Checks the lane/register mapping of the transposing half-wave minimum used by
// imageanalysis_amd/csrc/match_knn2sym.hip (DPP bank masks, row_ror, v_permlane16_swap).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 rowmin_check.hip -o rowmin_check && ./rowmin_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void half_wave_min16(const int (&r)[16], int &m0, int &m1)
{
    int s[8], u[4], w[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int a = r[2 * k], b = r[2 * k + 1];
        const int t1 = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xF, 0x3, false);
        const int t2 = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xF, 0xC, false);
        s[k] = min(t1, t2);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int a = s[2 * k], b = s[2 * k + 1];
        const int t1 = __builtin_amdgcn_update_dpp(b, a, 0x12C, 0xF, 0x5, false);
        const int t2 = __builtin_amdgcn_update_dpp(a, b, 0x124, 0xF, 0xA, false);
        u[k] = min(t1, t2);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const v2u x = __builtin_amdgcn_permlane16_swap((unsigned)u[2 * k], (unsigned)u[2 * k + 1], false, false);
        w[k] = min((int)x[0], (int)x[1]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        w[k] = min(w[k], __builtin_amdgcn_update_dpp(0, w[k], 0xB1, 0xF, 0xF, true));
        w[k] = min(w[k], __builtin_amdgcn_update_dpp(0, w[k], 0x4E, 0xF, 0xF, true));
    }
    m0 = w[0];
    m1 = w[1];
}

__global__ void k(const int *in, int *out)
{
    int r[16];
    for (int i = 0; i < 16; ++i) r[i] = in[threadIdx.x * 16 + i];
    int m0, m1;
    half_wave_min16(r, m0, m1);
    out[threadIdx.x * 2] = m0;
    out[threadIdx.x * 2 + 1] = m1;
}

int main()
{
    int h[64 * 16], o[128], *din, *dout;
    srand(1);
    for (int i = 0; i < 64 * 16; ++i) h[i] = rand() % 100000;
    hipMalloc(&din, sizeof(h));
    hipMalloc(&dout, sizeof(o));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 5, b4 = (lane >> 4) & 1, b3 = (lane >> 3) & 1, b2 = (lane >> 2) & 1;
        for (int which = 0; which < 2; ++which) {
            const int reg = 8 * which + 4 * b4 + 2 * b2 + b3;
            int want = 1 << 30;
            for (int l = 32 * g; l < 32 * g + 32; ++l) want = h[l * 16 + reg] < want ? h[l * 16 + reg] : want;
            if (o[lane * 2 + which] != want) {
                if (bad < 8) {
                    // which (half, register) does the value belong to?
                    int found = -1;
                    for (int gg = 0; gg < 2 && found < 0; ++gg)
                        for (int rr = 0; rr < 16 && found < 0; ++rr) {
                            int mm = 1 << 30;
                            for (int l = 32 * gg; l < 32 * gg + 32; ++l) mm = h[l * 16 + rr] < mm ? h[l * 16 + rr] : mm;
                            if (mm == o[lane * 2 + which]) found = gg * 16 + rr;
                        }
                    printf("lane %d m%d: got %d want %d (reg %d); value is the min of half %d reg %d\n", lane,
                           which, o[lane * 2 + which], want, reg, found >> 4, found & 15);
                }
                ++bad;
            }
        }
    }
    printf("rowmin_check: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
    return bad != 0;
}
