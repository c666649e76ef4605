// K2 (fast form) -- exact top-2 *distances* with 2 VALU instructions per distance (gfx950).
//
// Why a second form: on gfx950 the 32-bit min/max/med3/lshl_add VALU ops are half rate and an
// MFMA hides only ~5 of them (profiles/r1_ubench_valu_mfma.txt), so the general kernel
// (match_knn2.hip: key = (acc<<9)+term, med3, min = 3 ops per distance) is VALU-issue bound.
// Here the per-train-row term rides in through the MFMA's C operand and the index is not
// tracked in the sweep at all:
//     d2(q,t) = norm_q[q] + NB[t] + 2*acc,   NB = |s_t|^2 + 2*sum(s_t),   acc = sum(~s_q * s_t)
//     val     = (NB >> 1) + acc              (C operand initialised with NB >> 1)
//     d2      = norm_q[q] + 2*val + (NB & 1)
// Inside one parity class (NB & 1 equal) ordering by `val` IS ordering by d2, so the train
// rows of an image are stored partitioned by that parity (stable, each class padded to 128
// rows) and the sweep keeps (m1, m2) per class with  m2 = med3(m1, m2, val); m1 = min(m1, val).
// Distances of different parity cannot tie, so merging the two classes at the end is exact.
// Besides (d2_best, d2_second) the kernel records the 32-row tile in which the best value first
// appeared (1 compare + 1 select per 16 distances); the train index is then recovered only for
// the queries that survive the reference's metric threshold (scripts/lib/matcher.py:253-263) by
// knn2v2_resolve_kernel, which recomputes the 32 exact distances of that tile with v_dot4 and
// takes the lowest original row that reaches d2_best (= cv2.BFMatcher order, lowest index on
// ties: the partition is stable, so the first tile / first row is the lowest original index).
//
// Bound form (BOUND = true, the shipped fast path): the sweep keeps only the two smallest
// *16-row group minima* per query (min3 tree: 8 + 4 ops per 16 distances instead of 34), i.e.
//     best   = exact smallest distance,
//     second = smallest distance outside the best's group  >=  true second.
// The metric test of the reference is monotone in `second`, so thresholding with this upper
// bound keeps a superset of the true survivors; knn2v2_finish_kernel then recomputes the 32
// distances of the best's tile for those rows, takes second = min(bound, second inside the
// tile), re-applies the test, records the train index and compacts the survivors in place.
// Everything that leaves the path (survivor lists, their metrics, d2 of every survivor) is
// identical to the exact form; d2[.][1] of rows that fail the test stays an upper bound.
#include "iamx_common.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

#ifndef IAMX_DESC_OFFSET
#define IAMX_DESC_OFFSET 128      // stored byte s = value - offset (int8); d^2 does not depend on it
#endif
constexpr int D = IAMX_DESC_DIM;
constexpr int WAVES = 4, CHUNK = 128;
constexpr int QW_PRODUCT = 2;             // 32-query blocks per wave in the shipped kernel
constexpr int BIG = 0x3F000000;          // value of padding rows: loses every comparison

__device__ __forceinline__ int med3_after(int a, int b, int c, int dep)
{
    // v_med3_i32 as inline asm (keeps register pressure at ~150 VGPRs; the plain max/min form
    // lets the scheduler hoist whole tiles and spill).  hipcc does not pad hazards inside asm
    // statements, and `c` is a raw MFMA accumulator here: `dep` must be the result of a
    // compiler-generated VALU instruction that already read the same accumulator (the v_min in
    // front), so the MFMA -> VALU wait states are in place before this statement can issue.
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(dep));
    return r;
}

// ---------------------------------------------------------------------------------
// pack: one workgroup per image; stable partition of the rows by the parity of NB
// ---------------------------------------------------------------------------------
// Batched pack in three passes (all images of a batch per launch):
//   A  per row (8 threads): s2 = sum s^2, NB = s2 + 2*sum s            -> scratch
//   B  per image (one 1024-thread block): stable partition positions, meta, padding rows
//   C  per row (8 threads): convert + scatter the row, norm2 / cinit / perm
struct PackArgs {
    const void *src;             // [rows][128] u8 or f32, images back to back
    const int64_t *src_off;      // DEV [n_img+1] first source row of each image, or NULL:
    int64_t single_n;            //   one image of single_n rows
    const int32_t *dst_off;      // DEV [n_img] first packed row of each image (NULL: 0)
    int8_t *dst;
    int32_t *norm2, *cinit, *perm, *meta;
    int32_t *nb, *s2, *pos;      // scratch, one int per source row each
    int n_img;
};

template <typename SRC>
__global__ __launch_bounds__(256) void pack2_rows_kernel(PackArgs P, int64_t total_rows)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t >> 3;
    const int part = (int)(t & 7);
    int s2 = 0, s1 = 0;
    if (row < total_rows) {
        const SRC *p = static_cast<const SRC *>(P.src) + row * D + part * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int v;
            if constexpr (sizeof(SRC) == 1) {
                v = (int)p[i];
            } else {
                v = (int)rintf((float)p[i]);
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
            }
            const int s = v - IAMX_DESC_OFFSET;
            s2 += s * s;
            s1 += s;
        }
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        s2 += __shfl_xor(s2, m, 8);
        s1 += __shfl_xor(s1, m, 8);
    }
    if (row < total_rows && part == 0) {
        P.s2[row] = s2;
        P.nb[row] = s2 + 2 * s1;
    }
}

__global__ __launch_bounds__(1024) void pack2_partition_kernel(PackArgs P)
{
    __shared__ int wcnt[16];
    __shared__ int s_run_even, s_run_odd, s_ne;
    const int img = blockIdx.x;
    const int64_t r0 = P.src_off ? P.src_off[img] : 0;
    const int n = (int)(P.src_off ? P.src_off[img + 1] - r0 : P.single_n);
    const int d0 = P.dst_off ? P.dst_off[img] : 0;
    const int rows_cap = (n + CHUNK - 1) / CHUNK * CHUNK + CHUNK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int cnt = 0;
    for (int r = tid; r < n; r += 1024) cnt += (P.nb[r0 + r] & 1) == 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m);
    if (lane == 0) wcnt[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int i = 0; i < 16; ++i) t += wcnt[i];
        s_ne = t;
        s_run_even = 0;
        s_run_odd = 0;
    }
    __syncthreads();
    const int ne = s_ne, no = n - ne;
    const int ne_pad = (ne + CHUNK - 1) / CHUNK * CHUNK, no_pad = (no + CHUNK - 1) / CHUNK * CHUNK;
    for (int base = 0; base < n; base += 1024) {
        const int r = base + tid;
        const bool valid = r < n;
        const bool even = valid && (P.nb[r0 + r] & 1) == 0;
        const unsigned long long mask = __ballot(even);
        const int before_w = __popcll(mask & ((1ull << lane) - 1ull));
        __syncthreads();
        if (lane == 0) wcnt[wave] = __popcll(mask);
        __syncthreads();
        int before = before_w, total_even = 0;
        for (int i = 0; i < 16; ++i) {
            if (i < wave) before += wcnt[i];
            total_even += wcnt[i];
        }
        const int run_e = s_run_even, run_o = s_run_odd;
        if (valid) P.pos[r0 + r] = even ? run_e + before : ne_pad + run_o + (tid - before);
        __syncthreads();
        if (tid == 0) {
            const int rows = n - base < 1024 ? n - base : 1024;
            s_run_even = run_e + total_even;
            s_run_odd = run_o + rows - total_even;
        }
        __syncthreads();
    }
    for (int p = tid; p < rows_cap; p += 1024) {
        const bool pad = (p >= ne && p < ne_pad) || p >= ne_pad + no;
        if (pad) {
            uint4 *d4 = reinterpret_cast<uint4 *>(P.dst + (int64_t)(d0 + p) * D);
#pragma unroll
            for (int i = 0; i < 8; ++i) d4[i] = make_uint4(0, 0, 0, 0);
            P.norm2[d0 + p] = 0;
            P.cinit[d0 + p] = BIG;
            P.perm[d0 + p] = -1;
        }
    }
    if (tid == 0) {
        P.meta[4 * img + 0] = n;
        P.meta[4 * img + 1] = ne_pad / CHUNK;
        P.meta[4 * img + 2] = no_pad / CHUNK;
        P.meta[4 * img + 3] = ne;
    }
}

template <typename SRC>
__global__ __launch_bounds__(256) void pack2_scatter_kernel(PackArgs P)
{
    // grid = (blocks over the rows of an image, images)
    const int img = blockIdx.y;
    const int64_t r0 = P.src_off ? P.src_off[img] : 0;
    const int n = (int)(P.src_off ? P.src_off[img + 1] - r0 : P.single_n);
    const int d0 = P.dst_off ? P.dst_off[img] : 0;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int r = t >> 3, part = t & 7;
    if (r >= n) return;
    const SRC *p = static_cast<const SRC *>(P.src) + (r0 + r) * D + part * 16;
    unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int v;
        if constexpr (sizeof(SRC) == 1) {
            v = (int)p[i];
        } else {
            v = (int)rintf((float)p[i]);
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        w[i >> 2] |= (unsigned)((v - IAMX_DESC_OFFSET) & 0xFF) << (8 * (i & 3));
    }
    const int pos = d0 + P.pos[r0 + r];
    *reinterpret_cast<uint4 *>(P.dst + (int64_t)pos * D + part * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    if (part == 0) {
        P.norm2[pos] = P.s2[r0 + r];
        P.cinit[pos] = P.nb[r0 + r] >> 1;
        P.perm[pos] = r;
    }
}

// ---------------------------------------------------------------------------------
// sweep
// ---------------------------------------------------------------------------------
struct Args2 {
    const int8_t *desc_q;        // query store (original row order, 128-row padded, v1 layout)
    const int32_t *norm_q;
    const int32_t *qimg_off, *qimg_n;
    const int8_t *desc_t;        // train store (parity partitioned)
    const int32_t *cinit;
    const int32_t *timg_off;     // first packed row of each image in the train store
    const int32_t *tmeta;        // [n_img][4]: n, even chunks, odd chunks, n_even
    const int32_t *pairs, *wg_off;
    const int64_t *out_off;
    int32_t *out_d2;             // [rows][2]
    int32_t *out_tile;           // [rows]
    int n_pairs, total_wg;
};

// VARIANT != 0: timing ablations (iamxdbg_knn2v2_variant): bit0 no epilogue, bit1 no MFMA,
// bit2 no re-staging / barriers, bit3 operands without LDS traffic, bit4 = cost model of a
// single sweep that also serves the reverse direction (per-row minima over the queries:
// shift-add of the query term, running min per accumulator register, cross-lane DPP reduction
// per tile, one store per row and tile) -- timing only, results meaningless
template <int VARIANT, int QW, int OCC, int NW = WAVES, bool BOUND = false, bool DLDS = false>
__global__ __launch_bounds__(NW * 64, OCC) void knn2v2_kernel(Args2 A)
{
    constexpr int QB = NW * QW * 32;
    constexpr int NT = NW * 64;
    constexpr int PIECES = CHUNK * D / 16 / NT;
    __shared__ __attribute__((aligned(16))) int8_t lds[2 * CHUNK * D + 2 * CHUNK * 4];
    int8_t *lds_tile = lds;
    int *lds_tb = reinterpret_cast<int *>(lds + 2 * CHUNK * D);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, g = lane >> 5;

    int vid;
    {
        const int total = A.total_wg, bid = blockIdx.x;
        const int xcd = bid & 7, k = bid >> 3, q = total >> 3, r = total & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    int lo = 0, hi = A.n_pairs;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (A.wg_off[mid] <= vid) lo = mid; else hi = mid;
    }
    const int qimg = A.pairs[2 * lo], timg = A.pairs[2 * lo + 1];
    const int qoff = A.qimg_off[qimg], nq = A.qimg_n[qimg];
    const int toff = A.timg_off[timg];
    const int ne_ch = A.tmeta[4 * timg + 1], no_ch = A.tmeta[4 * timg + 2];
    const int64_t obase = A.out_off[lo];
    const int q0 = (vid - A.wg_off[lo]) * QB + wave * (QW * 32);

    v4i bq[QW][4];
#pragma unroll
    for (int qb = 0; qb < QW; ++qb) {
        int row = q0 + qb * 32 + c;
        row = row < nq ? row : nq - 1;
        const v4i *src = reinterpret_cast<const v4i *>(A.desc_q + (int64_t)(qoff + row) * D);
#pragma unroll
        for (int s = 0; s < 4; ++s) bq[qb][s] = ~src[2 * s + g];
    }
    int m1[QW], m2[QW], t1[QW];           // running class
    int e1[QW], e2[QW], et[QW];           // finished even class
#pragma unroll
    for (int qb = 0; qb < QW; ++qb) {
        m1[qb] = m2[qb] = BIG; t1[qb] = 0;
        e1[qb] = e2[qb] = BIG; et[qb] = 0;
    }

    const int8_t *tbase = A.desc_t + (int64_t)toff * D;
    const int32_t *tci = A.cinit + toff;
    v4i st[PIECES];
    int st_tb = BIG;
    auto load_chunk = [&](int ch) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j)
            st[j] = *reinterpret_cast<const v4i *>(tbase + (int64_t)(ch * CHUNK) * D + (j * NT + tid) * 16);
        if (tid < CHUNK) st_tb = tci[ch * CHUNK + tid];
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int e = j * NT + tid, row = e >> 3, slot = e & 7;
            *reinterpret_cast<v4i *>(lds_tile + buf * (CHUNK * D) + row * D + ((slot ^ ((row >> 1) & 7)) * 16)) = st[j];
        }
        if (tid < CHUNK) lds_tb[buf * CHUNK + tid] = st_tb;
    };

    // DLDS: global -> LDS directly (global_load_lds_dwordx4, no VGPR round trip, no ds_write).
    // The LDS destination of lane i is base + 16*i, so the XOR swizzle is applied on the
    // SOURCE side: position (row, p) of the tile receives logical slot p ^ ((row>>1)&7).
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto stage_direct = [&](int ch, int buf) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            const int e = j * NT + tid, row = e >> 3, slot = (e & 7) ^ ((row >> 1) & 7);
            const int8_t *gsrc = tbase + (int64_t)(ch * CHUNK + row) * D + slot * 16;
            int8_t *ldst = lds_tile + buf * (CHUNK * D) + (j * NT + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds(gsrc, (lds_ptr)ldst, 16, 0, 0);
        }
        if (wave < CHUNK / 64)
            __builtin_amdgcn_global_load_lds(tci + ch * CHUNK + tid,
                                             (lds_ptr)(lds_tb + buf * CHUNK + wave * 64), 4, 0, 0);
    };
    auto wait_direct = [&]() { __builtin_amdgcn_s_waitcnt(0x0F70); };       // vmcnt(0)

    const int nchunks = ne_ch + no_ch;
    if (nchunks > 0) {
        if constexpr (DLDS) {
            stage_direct(0, 0);
            wait_direct();
        } else {
            load_chunk(0);
            store_chunk(0);
        }
    }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if constexpr (!(VARIANT & 4)) {
            if (ch + 1 < nchunks) {
                if constexpr (DLDS) stage_direct(ch + 1, buf ^ 1);
                else load_chunk(ch + 1);
            }
        }
        if (ch == ne_ch) {                 // class boundary: park the even class
#pragma unroll
            for (int qb = 0; qb < QW; ++qb) {
                e1[qb] = m1[qb]; e2[qb] = m2[qb]; et[qb] = t1[qb];
                m1[qb] = m2[qb] = BIG;
            }
        }
        const int8_t *tile_base = lds_tile + ((VARIANT & 4) ? 0 : buf) * (CHUNK * D);
        const int *tb_base = lds_tb + ((VARIANT & 4) ? 0 : buf) * CHUNK;
        // operands of tile t+1 are fetched from LDS before the MFMAs of tile t are issued
        auto load_ops = [&](int tile, v4i (&a)[4], v4i (&tbv)[4]) {
            const int r = tile * 32 + c, swz = (r >> 1) & 7;
            if constexpr (VARIANT & 8) {       // ablation: operands without LDS traffic
#pragma unroll
                for (int s = 0; s < 4; ++s) { a[s] = bq[0][s] + ch + tile; tbv[s] = bq[1][s]; }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    a[s] = *reinterpret_cast<const v4i *>(tile_base + r * D + (((2 * s + g) ^ swz) * 16));
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    tbv[k] = *reinterpret_cast<const v4i *>(tb_base + tile * 32 + 8 * k + 4 * g);
            }
        };
        // C operand + 4 MFMAs (K = 128) of one 32x32 block
        auto chain = [&](v16i &acc, const v4i (&a)[4], const v4i (&tbv)[4], int qb) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[reg] = tbv[reg >> 2][reg & 3];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if constexpr (VARIANT & 2) acc[s] += a[s][0] ^ bq[qb][s][1];
                else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[qb][s], acc, 0, 0, 0);
            }
        };
        auto epilogue = [&](const v16i &acc, int qb, int tile_id) {
            if constexpr (VARIANT & 1) {
                asm volatile("" ::"v"(acc));
            } else {
                const int before = m1[qb];
                if constexpr (BOUND) {
                    // (m1, m2) = two smallest 16-row group minima: 8 + 2 ops per 16 distances
                    int tm = min(min(acc[0], acc[1]), acc[2]);
#pragma unroll
                    for (int reg = 3; reg < 15; reg += 2) tm = min(min(tm, acc[reg]), acc[reg + 1]);
                    tm = min(tm, acc[15]);
                    const int lo1 = min(m1[qb], tm);
                    m2[qb] = med3_after(m1[qb], m2[qb], tm, lo1);
                    m1[qb] = lo1;
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int lo1 = min(m1[qb], acc[reg]);
                        m2[qb] = med3_after(m1[qb], m2[qb], acc[reg], lo1);
                        m1[qb] = lo1;
                    }
                }
                t1[qb] = m1[qb] < before ? tile_id : t1[qb];
            }
        };
        v4i a[4], tbv[4];
        load_ops(0, a, tbv);
#pragma unroll
        for (int tile = 0; tile < CHUNK / 32; ++tile) {
            v4i a_nx[4], tb_nx[4];
            if (tile + 1 < CHUNK / 32) load_ops(tile + 1, a_nx, tb_nx);
            const int tile_id = ch * (CHUNK / 32) + tile;
            int rowmin[16];
            if constexpr (VARIANT & 16) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) rowmin[reg] = BIG;
            }
#pragma unroll
            for (int qb = 0; qb < QW; ++qb) {
                v16i acc;
                chain(acc, a, tbv, qb);
                epilogue(acc, qb, tile_id);
                if constexpr (VARIANT & 16) {
                    const int nqv = bq[qb][0][0];                  // stands for norm_q of the column
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) rowmin[reg] = min(rowmin[reg], (acc[reg] << 1) + nqv);
                }
            }
            if constexpr (VARIANT & 16) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    int v = rowmin[reg];
                    v = min(v, __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true));     // quad xor 1
                    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true));     // quad xor 2
                    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true));    // row_half_mirror
                    v = min(v, __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true));    // row_mirror
                    v = min(v, __shfl_xor(v, 16));
                    rowmin[reg] = v;
                }
                if (c == 0) {
                    const unsigned slot = ((unsigned)(vid * NW + wave) * 977u + (unsigned)tile_id) & 0xFFFFu;
                    int *dst = A.out_tile + slot * 32 + g * 16;
#pragma unroll
                    for (int reg = 0; reg < 16; reg += 4)
                        *reinterpret_cast<v4i *>(dst + reg) = v4i{rowmin[reg], rowmin[reg + 1], rowmin[reg + 2], rowmin[reg + 3]};
                }
            }
            if (tile + 1 < CHUNK / 32) {
#pragma unroll
                for (int s = 0; s < 4; ++s) { a[s] = a_nx[s]; tbv[s] = tb_nx[s]; }
            }
        }
        if constexpr (!(VARIANT & 4)) {
            if constexpr (DLDS) {
                wait_direct();
            } else {
                if (ch + 1 < nchunks) store_chunk(buf ^ 1);
            }
            __syncthreads();
        }
    }
    if (nchunks == ne_ch) {                // no odd chunks at all: the running class is the even one
#pragma unroll
        for (int qb = 0; qb < QW; ++qb) {
            e1[qb] = m1[qb]; e2[qb] = m2[qb]; et[qb] = t1[qb];
            m1[qb] = m2[qb] = BIG;
        }
    }

    // ---- merge lane halves per class, then the two classes; store
#pragma unroll
    for (int qb = 0; qb < QW; ++qb) {
        auto merge_halves = [&](int &a1, int &a2, int &ta) {
            const int b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2, 32), tb = __shfl_xor(ta, 32);
            const int n1 = min(a1, b1);
            const int n2 = min(max(a1, b1), min(a2, b2));
            const int nt = a1 < b1 ? ta : (b1 < a1 ? tb : min(ta, tb));
            a1 = n1; a2 = n2; ta = nt;
        };
        merge_halves(e1[qb], e2[qb], et[qb]);
        merge_halves(m1[qb], m2[qb], t1[qb]);          // odd class
        const int row = q0 + qb * 32 + c;
        if (g == 0 && row < nq) {
            const int na = A.norm_q[qoff + row];
            // d2 = norm_q + 2*val + parity; BIG stays far above every real distance
            const int de1 = na + 2 * e1[qb], de2 = na + 2 * e2[qb];
            const int do1 = na + 2 * m1[qb] + 1, do2 = na + 2 * m2[qb] + 1;
            int best, second, tile;
            if (de1 < do1) { best = de1; tile = et[qb]; second = min(de2, do1); }
            else           { best = do1; tile = t1[qb]; second = min(do2, de1); }
            v2i od = {best, second};
            *reinterpret_cast<v2i *>(A.out_d2 + 2 * (obase + row)) = od;
            A.out_tile[obase + row] = tile;
        }
    }
}

// ---------------------------------------------------------------------------------
// train index of the survivors: one workgroup per ordered pair, 32 lanes per survivor
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resolve_kernel(const int8_t *__restrict__ desc_q,
                                                      const int32_t *__restrict__ norm_q,
                                                      const int32_t *__restrict__ qimg_off,
                                                      const int8_t *__restrict__ desc_t,
                                                      const int32_t *__restrict__ norm2_t,
                                                      const int32_t *__restrict__ perm,
                                                      const int32_t *__restrict__ timg_off,
                                                      const int32_t *__restrict__ pairs,
                                                      const int64_t *__restrict__ out_off,
                                                      const int32_t *__restrict__ d2,
                                                      const int64_t *__restrict__ surv_off,
                                                      const int32_t *__restrict__ surv_q,
                                                      int32_t *__restrict__ surv_t /* in: tile, out: train row */,
                                                      int32_t *__restrict__ n_unresolved)
{
    const int p = blockIdx.x;
    const int qoff = qimg_off[pairs[2 * p]], toff = timg_off[pairs[2 * p + 1]];
    const int64_t b = surv_off[p], e = surv_off[p + 1], ob = out_off[p];
    const int grp = threadIdx.x >> 5, j = threadIdx.x & 31;
    for (int64_t s = b + grp; s < e; s += 8) {
        const int q = surv_q[s], tile = surv_t[s];
        const int best = d2[2 * (ob + q)];
        const int trow = toff + tile * 32 + j;
        const v4i *qa = reinterpret_cast<const v4i *>(desc_q + (int64_t)(qoff + q) * D);
        const v4i *ta = reinterpret_cast<const v4i *>(desc_t + (int64_t)trow * D);
        int dot = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const v4i a = qa[k], b = ta[k];
            dot = __builtin_amdgcn_sdot4(a.x, b.x, dot, false);
            dot = __builtin_amdgcn_sdot4(a.y, b.y, dot, false);
            dot = __builtin_amdgcn_sdot4(a.z, b.z, dot, false);
            dot = __builtin_amdgcn_sdot4(a.w, b.w, dot, false);
        }
        const int dd = norm_q[qoff + q] + norm2_t[trow] - 2 * dot;
        const int orig = perm[trow];
        const unsigned long long m = __ballot(orig >= 0 && dd == best);
        const unsigned half = (unsigned)(m >> (32 * ((threadIdx.x >> 5) & 1)));
        if (j == 0) {
            if (half) {
                surv_t[s] = perm[toff + tile * 32 + (__ffs(half) - 1)];
            } else {
                surv_t[s] = -1;
                atomicAdd(n_unresolved, 1);
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// bound form: exact second + final threshold + train index + in-place compaction;
// one workgroup per ordered pair, 32 lanes per candidate, candidates in ascending query order
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finish_kernel(const int8_t *__restrict__ desc_q,
                                                     const int32_t *__restrict__ norm_q,
                                                     const int32_t *__restrict__ qimg_off,
                                                     const int8_t *__restrict__ desc_t,
                                                     const int32_t *__restrict__ norm2_t,
                                                     const int32_t *__restrict__ perm,
                                                     const int32_t *__restrict__ timg_off,
                                                     const int32_t *__restrict__ pairs,
                                                     const int64_t *__restrict__ out_off,
                                                     int32_t *__restrict__ d2, double thresh,
                                                     const int64_t *__restrict__ surv_off,
                                                     int32_t *__restrict__ surv_q,
                                                     int32_t *__restrict__ surv_t /* in: tile, out: train row */,
                                                     double *__restrict__ surv_metric,
                                                     int32_t *__restrict__ surv_cnt,
                                                     int32_t *__restrict__ zero_div,
                                                     int32_t *__restrict__ n_unresolved)
{
    __shared__ int s_q[8], s_t[8], s_ok[8];
    __shared__ double s_m[8];
    const int p = blockIdx.x;
    const int qoff = qimg_off[pairs[2 * p]], toff = timg_off[pairs[2 * p + 1]];
    const int64_t b = surv_off[p], e = surv_off[p + 1], ob = out_off[p];
    const int grp = threadIdx.x >> 5, j = threadIdx.x & 31;
    int64_t w = b;                                   // next write position (w <= read position)
    for (int64_t s0 = b; s0 < e; s0 += 8) {
        const int64_t s = s0 + grp;
        if (s < e) {
            const int q = surv_q[s], tile = surv_t[s];
            const v2i dd2 = *reinterpret_cast<const v2i *>(d2 + 2 * (ob + q));
            const int best = dd2.x;
            const int trow = toff + tile * 32 + j;
            const v4i *qa = reinterpret_cast<const v4i *>(desc_q + (int64_t)(qoff + q) * D);
            const v4i *ta = reinterpret_cast<const v4i *>(desc_t + (int64_t)trow * D);
            int dot = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {            // 16-byte loads: 8 per row instead of 32
                const v4i a = qa[k], b = ta[k];
                dot = __builtin_amdgcn_sdot4(a.x, b.x, dot, false);
                dot = __builtin_amdgcn_sdot4(a.y, b.y, dot, false);
                dot = __builtin_amdgcn_sdot4(a.z, b.z, dot, false);
                dot = __builtin_amdgcn_sdot4(a.w, b.w, dot, false);
            }
            const int orig = perm[trow];
            const int dd = orig >= 0 ? norm_q[qoff + q] + norm2_t[trow] - 2 * dot : 0x7FFFFFFF;
            const unsigned long long m = __ballot(dd == best);
            const unsigned half = (unsigned)(m >> (32 * ((threadIdx.x >> 5) & 1)));
            // second inside the tile: the best again if it occurs twice, else the next value
            int other = (half & (half - 1)) ? best : (dd == best ? 0x7FFFFFFF : dd);
#pragma unroll
            for (int sh = 16; sh >= 1; sh >>= 1) other = min(other, __shfl_xor(other, sh));
            if (j == 0) {
                const int second = min(dd2.y, other);
                d2[2 * (ob + q) + 1] = second;
                const float f0 = (float)sqrt((double)best);
                const float f1 = (float)sqrt((double)second);
                double mt;
                bool ok = false;
                if (f1 == 0.0f) {
                    mt = __longlong_as_double(0x7FF8000000000000LL);
                    atomicAdd(zero_div, 1);
                } else {
                    const double ratio = (double)f0 / (double)f1;
                    mt = (double)f0 * ratio;
                    ok = mt < thresh;
                }
                int t = -1;
                if (half) t = perm[toff + tile * 32 + (__ffs(half) - 1)];
                else if (ok) atomicAdd(n_unresolved, 1);
                s_q[grp] = q; s_t[grp] = t; s_m[grp] = mt; s_ok[grp] = ok ? 1 : 0;
            }
        } else if (j == 0) {
            s_ok[grp] = 0;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (s_ok[k]) {
                    surv_q[w] = s_q[k]; surv_t[w] = s_t[k]; surv_metric[w] = s_m[k];
                    ++w;
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) surv_cnt[p] = (int)(w - b);
}

}  // namespace

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" int64_t iamx_desc2_rows_cap(int64_t n_rows)
{
    if (n_rows < 0) return 0;
    return (n_rows + CHUNK - 1) / CHUNK * CHUNK + CHUNK;
}

template <typename SRC>
static int pack2_launch(PackArgs P, int64_t total_rows, int max_rows, void *stream, const char *what)
{
    hipStream_t st = iamx::as_stream(stream);
    if (total_rows > 0)
        hipLaunchKernelGGL(pack2_rows_kernel<SRC>, dim3((unsigned)((total_rows * 8 + 255) / 256)),
                           dim3(256), 0, st, P, total_rows);
    hipLaunchKernelGGL(pack2_partition_kernel, dim3((unsigned)P.n_img), dim3(1024), 0, st, P);
    if (max_rows > 0)
        hipLaunchKernelGGL(pack2_scatter_kernel<SRC>,
                           dim3((unsigned)(((int64_t)max_rows * 8 + 255) / 256), (unsigned)P.n_img),
                           dim3(256), 0, st, P);
    return iamx::check_launch(what);
}

template <typename SRC>
static int pack2_single(const SRC *src, int64_t n_rows, int8_t *dst, int32_t *norm2, int32_t *cinit,
                        int32_t *perm, int32_t *meta, int32_t *scratch, void *stream,
                        const char *what)
{
    if (n_rows < 0 || n_rows > (1 << 24) || (n_rows > 0 && !src) || !dst || !norm2 || !cinit ||
        !perm || !meta || !scratch)
        return iamx::fail(IAMX_EINVAL, "%s: null pointer or bad row count", what);
    PackArgs P{src, nullptr, n_rows, nullptr, dst, norm2, cinit, perm, meta,
               scratch, scratch + n_rows, scratch + 2 * n_rows, 1};
    return pack2_launch<SRC>(P, n_rows, (int)n_rows, stream, what);
}

extern "C" int iamx_desc2_pack_u8(const uint8_t *src, int64_t n_rows, int8_t *dst, int32_t *norm2,
                                  int32_t *cinit, int32_t *perm, int32_t *meta, int32_t *scratch,
                                  void *stream)
{
    return pack2_single(src, n_rows, dst, norm2, cinit, perm, meta, scratch, stream,
                        "iamx_desc2_pack_u8");
}

extern "C" int iamx_desc2_pack_f32(const float *src, int64_t n_rows, int8_t *dst, int32_t *norm2,
                                   int32_t *cinit, int32_t *perm, int32_t *meta, int32_t *scratch,
                                   void *stream)
{
    return pack2_single(src, n_rows, dst, norm2, cinit, perm, meta, scratch, stream,
                        "iamx_desc2_pack_f32");
}

extern "C" int iamx_desc2_pack_batch_u8(const uint8_t *src, const int64_t *src_off,
                                        const int32_t *dst_off, int n_img, int64_t total_rows,
                                        int max_rows_per_image, int8_t *dst, int32_t *norm2,
                                        int32_t *cinit, int32_t *perm, int32_t *meta,
                                        int32_t *scratch, void *stream)
{
    IAMX_REQUIRE(src && src_off && dst_off && dst && norm2 && cinit && perm && meta && scratch,
                 "null pointer");
    IAMX_REQUIRE(n_img > 0 && total_rows >= 0 && max_rows_per_image >= 0, "bad count");
    PackArgs P{src, src_off, 0, dst_off, dst, norm2, cinit, perm, meta,
               scratch, scratch + total_rows, scratch + 2 * total_rows, n_img};
    return pack2_launch<uint8_t>(P, total_rows, max_rows_per_image, stream,
                                 "iamx_desc2_pack_batch_u8");
}

extern "C" int iamx_knn2v2_pairs(const int8_t *desc_q, const int32_t *norm_q,
                                 const int32_t *qimg_off, const int32_t *qimg_n,
                                 const int8_t *desc_t, const int32_t *cinit,
                                 const int32_t *timg_off, const int32_t *tmeta,
                                 const int32_t *pairs, const int32_t *wg_off,
                                 const int64_t *out_off, int n_pairs, int total_wg,
                                 int rows_per_wg, int exact_second, int32_t *out_d2,
                                 int32_t *out_tile, void *stream)
{
    IAMX_REQUIRE(desc_q && norm_q && qimg_off && qimg_n && desc_t && cinit && timg_off && tmeta &&
                     pairs && wg_off && out_off && out_d2 && out_tile,
                 "null pointer");
    IAMX_REQUIRE(n_pairs >= 0 && total_wg >= 0, "negative count");
    IAMX_REQUIRE(rows_per_wg == 256 || rows_per_wg == 512 || (rows_per_wg == 1024 && !exact_second),
                 "rows_per_wg must be 256, 512 or (bound form only) 1024");
    if (n_pairs == 0 || total_wg == 0) return IAMX_OK;
    Args2 a{desc_q, norm_q, qimg_off, qimg_n, desc_t, cinit, timg_off, tmeta, pairs, wg_off,
            out_off, out_d2, out_tile, n_pairs, total_wg};
    const dim3 g((unsigned)total_wg), b(WAVES * 64);
    hipStream_t st = iamx::as_stream(stream);
    // 512 rows = 4 query blocks per wave: fewer LDS reads per MFMA (-7 %)
    // 1024 rows = 8 waves x 4 query blocks, train chunks staged global -> LDS directly (-2 %)
    if (rows_per_wg == 1024)
        hipLaunchKernelGGL((knn2v2_kernel<0, 4, 2, 8, true, true>), g, dim3(512), 0, st, a);
    else if (rows_per_wg == 512 && exact_second) hipLaunchKernelGGL((knn2v2_kernel<0, 4, 2>), g, b, 0, st, a);
    else if (rows_per_wg == 512) hipLaunchKernelGGL((knn2v2_kernel<0, 4, 2, WAVES, true>), g, b, 0, st, a);
    else if (exact_second) hipLaunchKernelGGL((knn2v2_kernel<0, QW_PRODUCT, 2>), g, b, 0, st, a);
    else hipLaunchKernelGGL((knn2v2_kernel<0, QW_PRODUCT, 2, WAVES, true>), g, b, 0, st, a);
    return iamx::check_launch("iamx_knn2v2_pairs");
}

#ifdef IAMX_ABLATE   // scaffolding of tools/*_ablate.py: built into libiamx_ablate.so only
extern "C" int iamxdbg_knn2v2_variant(int variant, const int8_t *desc_q, const int32_t *norm_q,
                                      const int32_t *qimg_off, const int32_t *qimg_n,
                                      const int8_t *desc_t, const int32_t *cinit,
                                      const int32_t *timg_off, const int32_t *tmeta,
                                      const int32_t *pairs, const int32_t *wg_off,
                                      const int64_t *out_off, int n_pairs, int total_wg,
                                      int32_t *out_d2, int32_t *out_tile, void *stream)
{
    Args2 a{desc_q, norm_q, qimg_off, qimg_n, desc_t, cinit, timg_off, tmeta, pairs, wg_off,
            out_off, out_d2, out_tile, n_pairs, total_wg};
    dim3 g((unsigned)total_wg), b(WAVES * 64);
    hipStream_t st = iamx::as_stream(stream);
    switch (variant) {
    case 0: hipLaunchKernelGGL((knn2v2_kernel<0, 2, 2>), g, b, 0, st, a); break;
    case 1: hipLaunchKernelGGL((knn2v2_kernel<1, 2, 2>), g, b, 0, st, a); break;
    case 2: hipLaunchKernelGGL((knn2v2_kernel<2, 2, 2>), g, b, 0, st, a); break;
    case 4: hipLaunchKernelGGL((knn2v2_kernel<4, 2, 2>), g, b, 0, st, a); break;
    case 5: hipLaunchKernelGGL((knn2v2_kernel<5, 2, 2>), g, b, 0, st, a); break;
    case 6: hipLaunchKernelGGL((knn2v2_kernel<6, 2, 2>), g, b, 0, st, a); break;
    case 13: hipLaunchKernelGGL((knn2v2_kernel<13, 2, 2>), g, b, 0, st, a); break;
    case 12: hipLaunchKernelGGL((knn2v2_kernel<12, 2, 2>), g, b, 0, st, a); break;
    case 9: hipLaunchKernelGGL((knn2v2_kernel<9, 2, 2>), g, b, 0, st, a); break;
    case 30: hipLaunchKernelGGL((knn2v2_kernel<0, 3, 2>), g, b, 0, st, a); break;
    case 31: hipLaunchKernelGGL((knn2v2_kernel<0, 4, 2>), g, b, 0, st, a); break;
    case 32: hipLaunchKernelGGL((knn2v2_kernel<0, 4, 1>), g, b, 0, st, a); break;
    case 33: hipLaunchKernelGGL((knn2v2_kernel<0, 1, 4>), g, b, 0, st, a); break;
    case 34: hipLaunchKernelGGL((knn2v2_kernel<0, 2, 3>), g, b, 0, st, a); break;
    case 35: hipLaunchKernelGGL((knn2v2_kernel<1, 4, 2>), g, b, 0, st, a); break;
    case 40: hipLaunchKernelGGL((knn2v2_kernel<0, 2, 2, 8>), g, dim3(512), 0, st, a); break;
    case 41: hipLaunchKernelGGL((knn2v2_kernel<0, 4, 2, 8>), g, dim3(512), 0, st, a); break;
    case 42: hipLaunchKernelGGL((knn2v2_kernel<0, 3, 2, 8>), g, dim3(512), 0, st, a); break;
    case 43: hipLaunchKernelGGL((knn2v2_kernel<0, 1, 4, 8>), g, dim3(512), 0, st, a); break;
    case 50: hipLaunchKernelGGL((knn2v2_kernel<0, 2, 2, 4, true>), g, b, 0, st, a); break;
    case 51: hipLaunchKernelGGL((knn2v2_kernel<0, 4, 2, 4, true>), g, b, 0, st, a); break;
    case 52: hipLaunchKernelGGL((knn2v2_kernel<0, 2, 2, 8, true>), g, dim3(512), 0, st, a); break;
    case 53: hipLaunchKernelGGL((knn2v2_kernel<0, 4, 1, 4, true>), g, b, 0, st, a); break;
    case 54: hipLaunchKernelGGL((knn2v2_kernel<0, 2, 4, 4, true>), g, b, 0, st, a); break;
    case 55: hipLaunchKernelGGL((knn2v2_kernel<0, 3, 2, 4, true>), g, b, 0, st, a); break;
    case 56: hipLaunchKernelGGL((knn2v2_kernel<0, 4, 2, 4, true, true>), g, b, 0, st, a); break;
    case 57: hipLaunchKernelGGL((knn2v2_kernel<0, 2, 2, 4, true, true>), g, b, 0, st, a); break;
    case 58: hipLaunchKernelGGL((knn2v2_kernel<0, 4, 2, 8, true, true>), g, dim3(512), 0, st, a); break;
    case 59: hipLaunchKernelGGL((knn2v2_kernel<0, 3, 2, 4, true, true>), g, b, 0, st, a); break;
    case 60: hipLaunchKernelGGL((knn2v2_kernel<1, 4, 2, 4, true>), g, b, 0, st, a); break;
    case 70: hipLaunchKernelGGL((knn2v2_kernel<16, 4, 2, 4, true>), g, b, 0, st, a); break;
    case 71: hipLaunchKernelGGL((knn2v2_kernel<16, 4, 1, 4, true>), g, b, 0, st, a); break;
    case 72: hipLaunchKernelGGL((knn2v2_kernel<16, 2, 2, 4, true>), g, b, 0, st, a); break;
    case 61: hipLaunchKernelGGL((knn2v2_kernel<4, 4, 2, 4, true>), g, b, 0, st, a); break;
    case 62: hipLaunchKernelGGL((knn2v2_kernel<12, 4, 2, 4, true>), g, b, 0, st, a); break;
    case 63: hipLaunchKernelGGL((knn2v2_kernel<13, 4, 2, 4, true>), g, b, 0, st, a); break;
    case 64: hipLaunchKernelGGL((knn2v2_kernel<5, 4, 2, 4, true>), g, b, 0, st, a); break;
    case 65: hipLaunchKernelGGL((knn2v2_kernel<2, 4, 2, 4, true>), g, b, 0, st, a); break;
    default: return iamx::fail(IAMX_EINVAL, "unknown variant");
    }
    return iamx::check_launch("iamxdbg_knn2v2_variant");
}
#endif

extern "C" int iamx_knn2v2_resolve(const int8_t *desc_q, const int32_t *norm_q,
                                   const int32_t *qimg_off, const int8_t *desc_t,
                                   const int32_t *norm2_t, const int32_t *perm,
                                   const int32_t *timg_off, const int32_t *pairs,
                                   const int64_t *out_off, const int32_t *d2,
                                   const int64_t *surv_off, const int32_t *surv_q, int32_t *surv_t,
                                   int n_pairs, int32_t *n_unresolved, void *stream)
{
    IAMX_REQUIRE(desc_q && norm_q && qimg_off && desc_t && norm2_t && perm && timg_off && pairs &&
                     out_off && d2 && surv_off && surv_q && surv_t && n_unresolved,
                 "null pointer");
    if (n_pairs <= 0) return IAMX_OK;
    hipLaunchKernelGGL(resolve_kernel, dim3((unsigned)n_pairs), dim3(256), 0,
                       iamx::as_stream(stream), desc_q, norm_q, qimg_off, desc_t, norm2_t, perm,
                       timg_off, pairs, out_off, d2, surv_off, surv_q, surv_t, n_unresolved);
    return iamx::check_launch("iamx_knn2v2_resolve");
}

extern "C" int iamx_knn2v2_finish(const int8_t *desc_q, const int32_t *norm_q,
                                  const int32_t *qimg_off, const int8_t *desc_t,
                                  const int32_t *norm2_t, const int32_t *perm,
                                  const int32_t *timg_off, const int32_t *pairs,
                                  const int64_t *out_off, int32_t *d2, double thresh,
                                  const int64_t *surv_off, int32_t *surv_q, int32_t *surv_t,
                                  double *surv_metric, int32_t *surv_cnt, int n_pairs,
                                  int32_t *zero_div, int32_t *n_unresolved, void *stream)
{
    IAMX_REQUIRE(desc_q && norm_q && qimg_off && desc_t && norm2_t && perm && timg_off && pairs &&
                     out_off && d2 && surv_off && surv_q && surv_t && surv_metric && surv_cnt &&
                     zero_div && n_unresolved,
                 "null pointer");
    if (n_pairs <= 0) return IAMX_OK;
    hipLaunchKernelGGL(finish_kernel, dim3((unsigned)n_pairs), dim3(256), 0,
                       iamx::as_stream(stream), desc_q, norm_q, qimg_off, desc_t, norm2_t, perm,
                       timg_off, pairs, out_off, d2, thresh, surv_off, surv_q, surv_t, surv_metric,
                       surv_cnt, zero_div, n_unresolved);
    return iamx::check_launch("iamx_knn2v2_finish");
}
