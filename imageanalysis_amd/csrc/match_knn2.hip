// K2 -- exact 2-NN descriptor matching for gfx950 (MI355X).
//
// Replaces the_matcher.knnMatch(des1, des2, k=2) of the reference
// (scripts/lib/matcher.py:203-216) and the metric/threshold loop (:253-263).
//
// Arithmetic.  Descriptors are integers a in 0..255.  With s = a-128 (int8):
//     sum((a-b)^2) = |s_a|^2 + (|s_b|^2 + 2*sum(s_b)) + 2*sum((~s_a) * s_b)
// because ~s = -s-1 = 127-a.  The last sum is a 32x32x(4x32) i8 MFMA contraction; every
// term is an exact int32, so results are bit-exact and order independent.
//
// Mapping.  A workgroup owns 256 query rows of one ordered (query image, train image)
// pair: 4 waves x 2 blocks of 32 queries.  Queries sit on the MFMA *columns* (B operand),
// so one lane sees ONE query (column lane&31) against 16 train rows per 32x32 tile: the
// running top-2 of a query is lane-local and costs 3 VALU ops per distance:
//     key  = (acc << 9) + ((norm_t << 8) | (row & 255))     (v_lshl_add_u32)
//     m2   = med3(m1, m2, key) ; m1 = min(m1, key)           (v_med3_i32, v_min_i32)
// The packed key orders by (distance, train row); its 8 index bits are folded into full
// (distance, index) pairs every 256 train rows.  The two lane halves (rows +4) and nothing
// else are merged at the end -- no cross-lane traffic in the sweep.
// Train rows stream global -> registers -> LDS in 128-row (16 KiB) chunks, double
// buffered; the 16-byte slots of a row are XOR-swizzled with (row>>1)&7 so the A-fragment
// ds_read_b128 (16 distinct rows per lane group) is bank-conflict free.
#include "iamx_common.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));

#ifndef IAMX_DESC_OFFSET
#define IAMX_DESC_OFFSET 128      // stored byte s = value - offset (int8); d^2 does not depend on it
#endif
constexpr int D = IAMX_DESC_DIM;        // 128 bytes per row
constexpr int QW = 2;                   // 32-query blocks per wave
constexpr int WAVES = 4;
constexpr int QB = WAVES * QW * 32;     // 256 query rows per workgroup
constexpr int CHUNK = 128;              // train rows staged per step
constexpr int KEY_INVALID = 0x7FFFFFFF;

struct Knn2Args {
    const int8_t *desc_q;       // packed store the query image offsets refer to
    const int8_t *desc_t;       // packed store the train image offsets refer to (same in batches)
    const int32_t *norm_q;
    const int32_t *norm_t;
    const int32_t *img_off;     // NULL => single pair described by the s_* fields
    const int32_t *img_n;
    const int32_t *pairs;
    const int32_t *wg_off;
    const int64_t *out_off;
    int32_t *out_idx;
    int32_t *out_d2;
    int n_pairs;
    int total_wg;
    int s_nq, s_nt;             // single-pair form: both images start at row 0 of their store
};

__device__ __forceinline__ int med3_i32(int a, int b, int c)
{
    // inline asm on purpose: it also keeps the scheduler from hoisting whole tiles (the plain
    // max/min form selects v_med3_i32 too but the kernel then needs 256 VGPRs and spills).
    // Its `c` operand is always the compiler-generated key (a VALU result), never a raw MFMA
    // accumulator, so no MFMA->VALU hazard hides inside the asm statement.
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ---------------------------------------------------------------------------------
// pack: u8 / f32 rows -> int8 (value-128), zero padded, + per-row norms
// ---------------------------------------------------------------------------------
template <typename SRC>
__global__ __launch_bounds__(256) void pack_kernel(const SRC *__restrict__ src, int64_t n_rows,
                                                   int64_t pad_rows, int8_t *__restrict__ dst,
                                                   int32_t *__restrict__ norm_q,
                                                   int32_t *__restrict__ norm_t)
{
    // 8 threads per row, 16 elements each
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t row = t >> 3;
    int part = (int)(t & 7);
    if (row >= pad_rows) return;
    int s2 = 0, s1 = 0;
    unsigned w[4] = {0, 0, 0, 0};
    if (row < n_rows) {
        const SRC *p = src + row * D + part * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int v;
            if constexpr (sizeof(SRC) == 1) {
                v = (int)p[i];
            } else {
                float f = (float)p[i];
                v = (int)rintf(f);
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
            }
            int s = v - IAMX_DESC_OFFSET;
            s2 += s * s;
            s1 += s;
            w[i >> 2] |= (unsigned)(s & 0xFF) << (8 * (i & 3));
        }
    }
    *reinterpret_cast<uint4 *>(dst + row * D + part * 16) = make_uint4(w[0], w[1], w[2], w[3]);
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
        s2 += __shfl_xor(s2, m, 8);
        s1 += __shfl_xor(s1, m, 8);
    }
    if (part == 0) {
        norm_q[row] = s2;
        norm_t[row] = s2 + 2 * s1;
    }
}

// ---------------------------------------------------------------------------------
// the distance + top-2 kernel
// ---------------------------------------------------------------------------------
struct Top2 {
    int d1, i1, d2, i2;
};

__device__ __forceinline__ bool lex_less(int da, int ia, int db, int ib)
{
    return da < db || (da == db && ia < ib);
}

// VARIANT is 0 in the product; other values are timing ablations reachable only through
// the iamxdbg_knn2_variant entry point (tools/knn2_ablate.py): bit0 skip the top-2 epilogue,
// bit1 skip the MFMAs, bit2 stage only the first chunk (no barriers in the sweep).
// QW_ = 32-query blocks per wave, WAVES_ = waves per workgroup, OCC = launch-bounds waves/SIMD,
// FLAGS bit0: s_setprio(1) around the MFMA groups, bit1: two independent top-2 chains per query.
template <int VARIANT, int QW_, int WAVES_, int OCC, int FLAGS>
__global__ __launch_bounds__(WAVES_ * 64, OCC) void knn2_pairs_kernel(Knn2Args A)
{
    constexpr int NT = WAVES_ * 64;            // threads
    constexpr int QB_ = WAVES_ * QW_ * 32;     // query rows per workgroup
    constexpr int PIECES = CHUNK * D / 16 / NT;  // 16-byte pieces staged per thread
    constexpr int NCH = (FLAGS & 2) ? 2 : 1;   // independent top-2 chains
    __shared__ __attribute__((aligned(16))) int8_t lds[2 * CHUNK * D + 2 * CHUNK * 4];
    int8_t *lds_tile = lds;                                   // [2][CHUNK][128]
    int *lds_tb = reinterpret_cast<int *>(lds + 2 * CHUNK * D);  // [2][CHUNK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = lane & 31;     // query column inside a 32-block / train row inside a tile
    const int g = lane >> 5;     // k-half for operands, +4 row offset for results

    // ---- XCD-aware block -> work id (block b runs on XCD b%8; give each XCD a
    //      contiguous run of work ids so the workgroups of a pair share one L2)
    int vid;
    {
        const int total = A.total_wg;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, k = bid >> 3;
        const int q = total >> 3, r = total & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    // ---- pair = largest p with wg_off[p] <= vid
    int qoff = 0, toff = 0, nq = A.s_nq, nt = A.s_nt, wg0 = 0;
    int64_t obase = 0;
    if (A.img_off != nullptr) {
        int lo = 0, hi = A.n_pairs;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (A.wg_off[mid] <= vid) lo = mid; else hi = mid;
        }
        const int qimg = A.pairs[2 * lo], timg = A.pairs[2 * lo + 1];
        qoff = A.img_off[qimg]; nq = A.img_n[qimg];
        toff = A.img_off[timg]; nt = A.img_n[timg];
        wg0 = A.wg_off[lo];
        obase = A.out_off[lo];
    }
    const int q0 = (vid - wg0) * QB_ + wave * (QW_ * 32);

    // ---- query fragments: B operand, lane (c,g) holds bytes [32s+16g, +16) of row c
    v4i bq[QW_][4];
#pragma unroll
    for (int qb = 0; qb < QW_; ++qb) {
        int row = q0 + qb * 32 + c;
        row = row < nq ? row : nq - 1;
        const v4i *src = reinterpret_cast<const v4i *>(A.desc_q + (int64_t)(qoff + row) * D);
#pragma unroll
        for (int s = 0; s < 4; ++s) bq[qb][s] = ~src[2 * s + g];
    }

    int m1[QW_][NCH], m2[QW_][NCH];
    Top2 best[QW_];
#pragma unroll
    for (int qb = 0; qb < QW_; ++qb) {
#pragma unroll
        for (int h = 0; h < NCH; ++h) m1[qb][h] = m2[qb][h] = KEY_INVALID;
        best[qb].d1 = best[qb].d2 = KEY_INVALID;
        best[qb].i1 = best[qb].i2 = 0;
    }

    // ---- staging: thread loads PIECES x 16 B of the 16 KiB chunk + (tid<128) one key term
    const int8_t *tbase = A.desc_t + (int64_t)toff * D;
    const int32_t *tnorm = A.norm_t + toff;
    v4i st[PIECES];
    int st_tb = KEY_INVALID;
    auto load_chunk = [&](int ch) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            int e = j * NT + tid;
            st[j] = *reinterpret_cast<const v4i *>(tbase + (int64_t)(ch * CHUNK) * D + e * 16);
        }
        if (tid < CHUNK) {
            int rr = ch * CHUNK + tid;
            st_tb = rr < nt ? tnorm[rr] * 256 + (rr & 255) : KEY_INVALID;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            int e = j * NT + tid;
            int row = e >> 3, slot = e & 7;
            int phys = slot ^ ((row >> 1) & 7);
            *reinterpret_cast<v4i *>(lds_tile + buf * (CHUNK * D) + row * D + phys * 16) = st[j];
        }
        if (tid < CHUNK) lds_tb[buf * CHUNK + tid] = st_tb;
    };

    const int nchunks = (nt + CHUNK - 1) / CHUNK;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if constexpr (!(VARIANT & 4))
            if (ch + 1 < nchunks) load_chunk(ch + 1);

        const int8_t *tile_base = lds_tile + ((VARIANT & 4) ? 0 : buf) * (CHUNK * D);
        const int *tb_base = lds_tb + ((VARIANT & 4) ? 0 : buf) * CHUNK;
#pragma unroll
        for (int tile = 0; tile < CHUNK / 32; ++tile) {
            const int r = tile * 32 + c;
            const int swz = (r >> 1) & 7;
            v4i a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                a[s] = *reinterpret_cast<const v4i *>(tile_base + r * D + (((2 * s + g) ^ swz) * 16));
            v4i tbv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                tbv[k] = *reinterpret_cast<const v4i *>(tb_base + tile * 32 + 8 * k + 4 * g);
#pragma unroll
            for (int qb = 0; qb < QW_; ++qb) {
                v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (FLAGS & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if constexpr (VARIANT & 2) {
                        acc[s] += a[s][0] ^ bq[qb][s][1];
                    } else {
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[qb][s], acc, 0, 0, 0);
                    }
                }
                if constexpr (FLAGS & 1) __builtin_amdgcn_s_setprio(0);
                if constexpr (VARIANT & 1) {
                    asm volatile("" ::"v"(acc));
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int h = (NCH == 2) ? (reg & 1) : 0;
                        int key = tbv[reg >> 2][reg & 3] + (acc[reg] << 9);
                        m2[qb][h] = med3_i32(m1[qb][h], m2[qb][h], key);
                        m1[qb][h] = min(m1[qb][h], key);
                    }
                }
            }
        }

        // fold the packed keys of this 256-row epoch into (distance, index) pairs
        if ((ch & 1) || ch == nchunks - 1) {
            const int sbase = (ch >> 1) * 256;
#pragma unroll
            for (int qb = 0; qb < QW_; ++qb) {
                int k1 = m1[qb][0], k2 = m2[qb][0];
                if constexpr (NCH == 2) {
                    // merge the two sorted key pairs (keys are unique inside an epoch)
                    const int a1 = m1[qb][0], a2 = m2[qb][0], b1 = m1[qb][1], b2 = m2[qb][1];
                    k1 = min(a1, b1);
                    k2 = min(max(a1, b1), min(a2, b2));
                }
                int d1k = k1 >> 8, i1k = sbase + (k1 & 255);
                int d2k = k2 >> 8, i2k = sbase + (k2 & 255);
                Top2 &b = best[qb];
                if (d1k < b.d1) {
                    if (d2k < b.d1) { b.d2 = d2k; b.i2 = i2k; }
                    else            { b.d2 = b.d1; b.i2 = b.i1; }
                    b.d1 = d1k; b.i1 = i1k;
                } else if (d1k < b.d2) {
                    b.d2 = d1k; b.i2 = i1k;
                }
#pragma unroll
                for (int h = 0; h < NCH; ++h) m1[qb][h] = m2[qb][h] = KEY_INVALID;
            }
        }

        if constexpr (!(VARIANT & 4)) {
            if (ch + 1 < nchunks) store_chunk(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- merge the two lane halves, add the query norm, store
#pragma unroll
    for (int qb = 0; qb < QW_; ++qb) {
        Top2 a = best[qb], b;
        b.d1 = __shfl_xor(a.d1, 32);
        b.i1 = __shfl_xor(a.i1, 32);
        b.d2 = __shfl_xor(a.d2, 32);
        b.i2 = __shfl_xor(a.i2, 32);
        int f_d, f_i, s_d, s_i;
        if (lex_less(b.d1, b.i1, a.d1, a.i1)) {
            f_d = b.d1; f_i = b.i1;
            if (lex_less(a.d1, a.i1, b.d2, b.i2)) { s_d = a.d1; s_i = a.i1; }
            else                                  { s_d = b.d2; s_i = b.i2; }
        } else {
            f_d = a.d1; f_i = a.i1;
            if (lex_less(b.d1, b.i1, a.d2, a.i2)) { s_d = b.d1; s_i = b.i1; }
            else                                  { s_d = a.d2; s_i = a.i2; }
        }
        const int row = q0 + qb * 32 + c;
        if (g == 0 && row < nq) {
            const int na = A.norm_q[qoff + row];
            v2i oi = {f_i, s_i};
            v2i od = {f_d + na, s_d + na};
            *reinterpret_cast<v2i *>(A.out_idx + 2 * (obase + row)) = oi;
            *reinterpret_cast<v2i *>(A.out_d2 + 2 * (obase + row)) = od;
        }
    }
}

// the configuration the product launches
#define IAMX_KNN2_PRODUCT knn2_pairs_kernel<0, QW, WAVES, 2, 0>

// ---------------------------------------------------------------------------------
// metric / threshold (scripts/lib/matcher.py:253-263), one workgroup per ordered pair
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void metric_kernel(const int32_t *__restrict__ d2,
                                                     const int64_t *__restrict__ seg_off,
                                                     double thresh, double *__restrict__ metric,
                                                     uint8_t *__restrict__ keep,
                                                     int32_t *__restrict__ seg_count,
                                                     int32_t *__restrict__ zero_div)
{
    __shared__ int wsum[4];
    const int seg = blockIdx.x;
    const int64_t b = seg_off[seg], e = seg_off[seg + 1];
    int cnt = 0, zd = 0;
    for (int64_t i = b + threadIdx.x; i < e; i += 256) {
        const v2i dd = *reinterpret_cast<const v2i *>(d2 + 2 * i);
        // cv2 L2 distance: float32 sqrt of the (exact) float32 sum
        // (correctly rounded: f64 sqrt of an integer < 2^24 rounded once more to f32 cannot
        //  land on a rounding boundary; tests/test_match_gpu.py checks all 8.3M values)
        const float d0 = (float)sqrt((double)dd.x);
        const float d1 = (float)sqrt((double)dd.y);
        double m;
        bool k = false;
        if (d1 == 0.0f) {
            m = __longlong_as_double(0x7FF8000000000000LL);   // python raises ZeroDivisionError
            zd++;
        } else {
            const double ratio = (double)d0 / (double)d1;
            m = (double)d0 * ratio;
            k = m < thresh;
        }
        metric[i] = m;
        keep[i] = k ? 1 : 0;
        cnt += k ? 1 : 0;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        cnt += __shfl_xor(cnt, m);
        zd += __shfl_xor(zd, m);
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    if ((threadIdx.x & 63) == 0 && zd) atomicAdd(zero_div, zd);
    __syncthreads();
    if (threadIdx.x == 0) seg_count[seg] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// order-preserving compaction, one workgroup per ordered pair
__global__ __launch_bounds__(256) void compact_kernel(const int32_t *__restrict__ idx, int idx_stride,
                                                      const double *__restrict__ metric,
                                                      const uint8_t *__restrict__ keep,
                                                      const int64_t *__restrict__ seg_off,
                                                      const int64_t *__restrict__ surv_off,
                                                      int32_t *__restrict__ surv_q,
                                                      int32_t *__restrict__ surv_t,
                                                      double *__restrict__ surv_metric)
{
    __shared__ int wcnt[4];
    const int seg = blockIdx.x;
    const int64_t b = seg_off[seg], e = seg_off[seg + 1];
    int64_t out = surv_off[seg];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = b; base < e; base += 256) {
        const int64_t i = base + threadIdx.x;
        const bool k = i < e && keep[i];
        const unsigned long long mask = __ballot(k);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave] = __popcll(mask);
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) woff += wcnt[w];
            tot += wcnt[w];
        }
        if (k) {
            const int64_t o = out + woff + before;
            surv_q[o] = (int32_t)(i - b);
            surv_t[o] = idx[(int64_t)idx_stride * i];
            surv_metric[o] = metric[i];
        }
        out += tot;
        __syncthreads();
    }
}

// single-workgroup exclusive scan (n is the number of ordered pairs in a batch: small)
__global__ __launch_bounds__(1024) void scan_kernel(const int32_t *__restrict__ in, int64_t n,
                                                    int64_t *__restrict__ out)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        long long v = i < n ? (long long)in[i] : 0;
        long long x = v;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            long long y = __shfl_up(x, m);
            if (lane >= m) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        long long woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            if (w < wave) woff += wsum[w];
            tot += wsum[w];
        }
        const long long carry = carry_s;
        if (i < n) out[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry_s;
}

}  // namespace

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" int64_t iamx_desc_padded_rows(int64_t n_rows)
{
    if (n_rows < 0) return 0;
    return (n_rows + IAMX_ROW_PAD - 1) / IAMX_ROW_PAD * IAMX_ROW_PAD;
}

template <typename SRC>
static int pack_impl(const SRC *src, int64_t n_rows, int8_t *dst, int32_t *norm_q,
                     int32_t *norm_t, void *stream, const char *what)
{
    if (n_rows < 0 || (n_rows > 0 && !src) || !dst || !norm_q || !norm_t)
        return iamx::fail(IAMX_EINVAL, "%s: null pointer or negative row count", what);
    const int64_t pad = iamx_desc_padded_rows(n_rows);
    if (pad == 0) return IAMX_OK;
    const int64_t threads = pad * 8;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(pack_kernel<SRC>, dim3(grid), dim3(256), 0, iamx::as_stream(stream), src,
                       n_rows, pad, dst, norm_q, norm_t);
    return iamx::check_launch(what);
}

extern "C" int iamx_desc_pack_u8(const uint8_t *src, int64_t n_rows, int8_t *dst,
                                 int32_t *norm_q, int32_t *norm_t, void *stream)
{
    return pack_impl(src, n_rows, dst, norm_q, norm_t, stream, "iamx_desc_pack_u8");
}

extern "C" int iamx_desc_pack_f32(const float *src, int64_t n_rows, int8_t *dst,
                                  int32_t *norm_q, int32_t *norm_t, void *stream)
{
    return pack_impl(src, n_rows, dst, norm_q, norm_t, stream, "iamx_desc_pack_f32");
}

// rows of `n_img` images of the original-order store back as uint8, laid back to back (image i:
// rows [dst_off[i], dst_off[i + 1]) of dst): value = stored + 128.  What rebuilds another layout of
// images whose source descriptors are no longer on the device (DescriptorStore.ensure_train_layout).
namespace {
__global__ __launch_bounds__(256) void unpack_u8_kernel(const int8_t *__restrict__ desc,
                                                        const int32_t *__restrict__ img_off,
                                                        const int64_t *__restrict__ dst_off,
                                                        uint8_t *__restrict__ dst)
{
    const int img = blockIdx.y;
    const int64_t n = dst_off[img + 1] - dst_off[img];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;          // 16 bytes per thread
    if (t >= n * 8) return;
    const uint4 v = reinterpret_cast<const uint4 *>(desc + (int64_t)img_off[img] * D)[t];
    reinterpret_cast<uint4 *>(dst + dst_off[img] * D)[t] =
        make_uint4(v.x ^ 0x80808080u, v.y ^ 0x80808080u, v.z ^ 0x80808080u, v.w ^ 0x80808080u);
}
}  // namespace

extern "C" int iamx_desc_unpack_u8(const int8_t *desc, const int32_t *img_off, const int64_t *dst_off,
                                   int n_img, int max_rows_per_image, uint8_t *dst, void *stream)
{
    IAMX_REQUIRE(desc && img_off && dst_off && dst, "null pointer");
    IAMX_REQUIRE(n_img >= 0 && max_rows_per_image >= 0, "bad count");
    if (n_img == 0 || max_rows_per_image == 0) return IAMX_OK;
    hipLaunchKernelGGL(unpack_u8_kernel, dim3((unsigned)(((int64_t)max_rows_per_image * 8 + 255) / 256), (unsigned)n_img),
                       dim3(256), 0, iamx::as_stream(stream), desc, img_off, dst_off, dst);
    return iamx::check_launch("iamx_desc_unpack_u8");
}

extern "C" int iamx_knn2_wg_per_pair(int n_query_rows)
{
    return n_query_rows <= 0 ? 0 : (n_query_rows + QB - 1) / QB;
}

extern "C" int iamx_knn2_l2_pairs(const int8_t *desc, const int32_t *norm_q,
                                  const int32_t *norm_t, const int32_t *img_off,
                                  const int32_t *img_n, const int32_t *pairs,
                                  const int32_t *wg_off, const int64_t *out_off, int n_pairs,
                                  int total_wg, int32_t *out_idx, int32_t *out_d2, void *stream)
{
    IAMX_REQUIRE(desc && norm_q && norm_t && img_off && img_n && pairs && wg_off && out_off &&
                     out_idx && out_d2,
                 "null pointer");
    IAMX_REQUIRE(n_pairs >= 0 && total_wg >= 0, "negative count");
    if (n_pairs == 0 || total_wg == 0) return IAMX_OK;
    Knn2Args a{desc, desc, norm_q, norm_t, img_off, img_n, pairs, wg_off, out_off,
               out_idx, out_d2, n_pairs, total_wg, 0, 0};
    hipLaunchKernelGGL(IAMX_KNN2_PRODUCT, dim3((unsigned)total_wg), dim3(WAVES * 64), 0,
                       iamx::as_stream(stream), a);
    return iamx::check_launch("iamx_knn2_l2_pairs");
}

#ifdef IAMX_ABLATE   // scaffolding of tools/*_ablate.py: built into libiamx_ablate.so only
// timing ablations (not part of the C ABI; see the VARIANT comment above)
extern "C" int iamxdbg_knn2_variant(int variant, const int8_t *desc, const int32_t *norm_q,
                                    const int32_t *norm_t, const int32_t *img_off,
                                    const int32_t *img_n, const int32_t *pairs,
                                    const int32_t *wg_off, const int64_t *out_off, int n_pairs,
                                    int total_wg, int32_t *out_idx, int32_t *out_d2, void *stream)
{
    Knn2Args a{desc, desc, norm_q, norm_t, img_off, img_n, pairs, wg_off, out_off,
               out_idx, out_d2, n_pairs, total_wg, 0, 0};
    hipStream_t st = iamx::as_stream(stream);
    dim3 g((unsigned)total_wg);
#define V(id, ...) case id: hipLaunchKernelGGL((knn2_pairs_kernel<__VA_ARGS__>), g, dim3(WV * 64), 0, st, a); break;
    switch (variant) {
#define WV 4
        V(0, 0, 2, 4, 2, 0) V(1, 1, 2, 4, 2, 0) V(2, 2, 2, 4, 2, 0) V(4, 4, 2, 4, 2, 0)
        V(5, 5, 2, 4, 2, 0) V(6, 6, 2, 4, 2, 0)
        V(10, 0, 2, 4, 2, 1)      // setprio
        V(11, 0, 2, 4, 2, 2)      // dual chains
        V(12, 0, 2, 4, 2, 3)      // both
        V(13, 0, 2, 4, 3, 0)      // occupancy hint 3
        V(14, 0, 2, 4, 4, 0)      // occupancy hint 4 (<=128 VGPR)
        V(15, 0, 2, 4, 4, 2)
        V(16, 0, 1, 4, 4, 0)      // QW=1: 128 queries per workgroup
        V(17, 0, 1, 4, 4, 2)
        V(18, 0, 3, 4, 2, 0)      // QW=3
        V(19, 0, 4, 4, 1, 0)      // QW=4
#undef WV
#define WV 8
        V(20, 0, 2, 8, 2, 0)      // 8 waves share the staged tile
        V(21, 0, 2, 8, 2, 2)
        V(22, 0, 1, 8, 2, 0)
        V(23, 0, 1, 8, 4, 2)
#undef WV
    default: return iamx::fail(IAMX_EINVAL, "unknown variant");
    }
#undef V
    return iamx::check_launch("iamxdbg_knn2_variant");
}
#endif

extern "C" int iamx_knn2_l2_u8(const int8_t *q_desc, const int32_t *q_norm_q, int nq,
                               const int8_t *t_desc, const int32_t *t_norm_t, int nt,
                               int32_t *idx, int32_t *d2, void *stream)
{
    IAMX_REQUIRE(q_desc && q_norm_q && t_desc && t_norm_t && idx && d2, "null pointer");
    IAMX_REQUIRE(nq >= 0, "negative query count");
    // the reference returns no matches when either side has <= 1 rows (matcher.py:205-210)
    IAMX_REQUIRE(nt >= 2, "train image needs at least 2 descriptors");
    if (nq == 0) return IAMX_OK;
    Knn2Args a{q_desc, t_desc, q_norm_q, t_norm_t, nullptr, nullptr, nullptr, nullptr, nullptr,
               idx, d2, 1, iamx_knn2_wg_per_pair(nq), nq, nt};
    hipLaunchKernelGGL(IAMX_KNN2_PRODUCT, dim3((unsigned)a.total_wg), dim3(WAVES * 64), 0,
                       iamx::as_stream(stream), a);
    return iamx::check_launch("iamx_knn2_l2_u8");
}

extern "C" int iamx_match_metric(const int32_t *d2, const int64_t *seg_off, int n_seg,
                                 double thresh, double *metric, uint8_t *keep,
                                 int32_t *seg_count, int32_t *zero_div, void *stream)
{
    IAMX_REQUIRE(d2 && seg_off && metric && keep && seg_count && zero_div, "null pointer");
    IAMX_REQUIRE(n_seg >= 0, "negative count");
    if (n_seg == 0) return IAMX_OK;
    hipLaunchKernelGGL(metric_kernel, dim3((unsigned)n_seg), dim3(256), 0, iamx::as_stream(stream),
                       d2, seg_off, thresh, metric, keep, seg_count, zero_div);
    return iamx::check_launch("iamx_match_metric");
}

extern "C" int iamx_match_compact(const int32_t *idx, int idx_stride, const double *metric,
                                  const uint8_t *keep,
                                  const int64_t *seg_off, const int64_t *surv_off, int n_seg,
                                  int32_t *surv_q, int32_t *surv_t, double *surv_metric,
                                  void *stream)
{
    IAMX_REQUIRE(idx && metric && keep && seg_off && surv_off && surv_q && surv_t && surv_metric,
                 "null pointer");
    IAMX_REQUIRE(n_seg >= 0 && idx_stride >= 1, "bad count / stride");
    if (n_seg == 0) return IAMX_OK;
    hipLaunchKernelGGL(compact_kernel, dim3((unsigned)n_seg), dim3(256), 0,
                       iamx::as_stream(stream), idx, idx_stride, metric, keep, seg_off, surv_off,
                       surv_q, surv_t, surv_metric);
    return iamx::check_launch("iamx_match_compact");
}

extern "C" int iamx_exclusive_scan_i32(const int32_t *in, int64_t n, int64_t *out, void *stream)
{
    IAMX_REQUIRE(in && out, "null pointer");
    IAMX_REQUIRE(n >= 0, "negative count");
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, iamx::as_stream(stream), in, n, out);
    return iamx::check_launch("iamx_exclusive_scan_i32");
}
