// K2 (symmetric form) -- ONE i8-MFMA sweep of a distance matrix serves BOTH directions of an
// image pair (gfx950).  Replaces the two knnMatch calls of bidirectional_pair_matches
// (scripts/lib/matcher.py:304-347 -> :203-216) and the metric loop (:253-263).
//
// Why: the one-direction fast form (match_knn2v2.hip) runs the MFMA pipe at ~80 % of its
// power-limited rate, but every distance matrix is executed twice (queries = image i against
// train = image j, then j against i).  The only factor left is that 2x.
//
// How: the sweep does not try to be exact.  For every row of BOTH images it only produces
//     L  <= exact best squared distance,      U  >= exact second squared distance
// cheaply enough to stay MFMA bound; the reference's test  d0*(d0/d1) < 270*ratio  is monotone
// (increasing in d0, decreasing in d1), so thresholding (L, U) keeps a SUPERSET of the rows the
// reference keeps (a few rows per image pair), and symexact_kernel recomputes those rows
// exactly (all train rows, v_dot4, lowest train row wins ties = cv2.BFMatcher order).  What
// leaves the path is bit-identical to the exact forms (tests/test_match_sym_gpu.py).
//
// Arithmetic.  s = value-128 (int8), n2 = |s|^2, nb = n2 + 2*sum(s), Ct = nb>>1, Cq = n2>>1,
// p = n2&1 (= nb&1).  With the B operand holding ~s_a = -s_a-1 and the C operand Ct[t]:
//     acc_out(a,t) = Ct[t] + sum(~s_a * s_t)         d2(a,t) = 2*(acc_out + Cq[a]) + p[a] + p[t]
// Column direction (queries = B rows, lane local): v = min_t acc_out, kept as 4 interleaved
//   running minima per lane (tile & 3) x 2 lane halves = 8 groups of train rows:
//   best >= 2*(v1+Cq[a]) + p[a],  second <= 2*(v2+Cq[a]) + p[a] + 1   (v2 = 2nd smallest group
//   minimum; 8 v_min3 per 16 distances = 0.5 VALU / distance).
// Row direction (queries = A rows, across lanes): R_w(t) = min over the 128 B rows of a wave of
//   acc_out: v_min3 across the wave's four 32-row blocks (0.5 VALU / distance), then a
//   TRANSPOSING butterfly over the 32 lanes (DPP quad_perm with bank masks, row_ror,
//   v_permlane16_swap: 36 instructions for 16 registers instead of 80).  The B rows of an image
//   are stored SORTED by n2, so Cq of a wave's rows lies in a narrow [lo_w, hi_w]:
//   group minimum in [R_w + lo_w, R_w + hi_w]; per train row the 8 waves of a workgroup are
//   merged through LDS into (L, U1, U2) = (min lower bound, two smallest upper bounds).
// One sorted store serves both roles (any order is legal for the A side).
//
// Workgroup shapes (iamx_knn2sym_sweep `form`).  Form 2, images of >= 4096 rows: 1024 B rows per
// workgroup as 4 waves x 8 query blocks, ONE wave per SIMD (WPE = 1, 512 registers: the B operand
// in the AGPR half, accumulators in VGPRs -- the file is compiled with -mllvm
// -amdgpu-mfma-vgpr-form) -- the butterfly and the LDS operand reads of a tile serve 32 MFMAs,
// 5.2 VALU per MFMA (round 3; before: 8 waves x 4 blocks at two waves per SIMD, 6.4 per MFMA).
// Forms 1 / 0 (512 / 256 rows per workgroup): 4 waves x 4 / 2 blocks, two waves per SIMD.
#include "iamx_common.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

#ifndef IAMX_DESC_OFFSET
#define IAMX_DESC_OFFSET 128
#endif
constexpr int D = IAMX_DESC_DIM;
constexpr int CHUNK = 128;
constexpr int BIG = 0x3F000000;          // Ct of padding rows: loses every comparison

// ---------------------------------------------------------------------------------
// pack ("desc3"): rows of an image sorted by n2 (stable), padded to 128 rows
//   A  per row (8 threads): n2, nb                       -> scratch
//   B  per row: rank = #{r' : (n2', r') < (n2, r)}        -> scratch (O(n^2), LDS tiled)
//   C  per row (8 threads): convert + scatter, sn2 / sct / sperm / sinv
//   D  padding rows
// ---------------------------------------------------------------------------------
struct Pack3Args {
    const void *src;             // [rows][128] u8 or f32, images back to back
    const int64_t *src_off;      // DEV [n_img+1] first source row of each image, or NULL:
    int64_t single_n;            //   one image of single_n rows
    const int32_t *dst_off;      // DEV [n_img] first packed row of each image (NULL: 0)
    int8_t *dst;
    int32_t *sn2, *sct, *sperm, *sinv;
    int32_t *n2, *nb, *pos;      // scratch, one int per source row each
    int n_img;
};

template <typename SRC>
__device__ __forceinline__ int load_value(const SRC *p, int i)
{
    if constexpr (sizeof(SRC) == 1) {
        return (int)p[i];
    } else {
        int v = (int)rintf((float)p[i]);
        return v < 0 ? 0 : (v > 255 ? 255 : v);
    }
}

template <typename SRC>
__global__ __launch_bounds__(256) void pack3_rows_kernel(Pack3Args P, int64_t total_rows)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = t >> 3;
    const int part = (int)(t & 7);
    int s2 = 0, s1 = 0;
    if (row < total_rows) {
        const SRC *p = static_cast<const SRC *>(P.src) + row * D + part * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int s = load_value(p, i) - IAMX_DESC_OFFSET;
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
        P.n2[row] = s2;
        P.nb[row] = s2 + 2 * s1;
    }
}

// rank of every row inside its image = rows with a smaller key (ties: the earlier row).
// FIRST by binning (round 6; one workgroup per image): the keys of an image spread over a few
// hundred thousand values, so 4096 equal-width bins between its smallest and its largest key hold a
// few dozen rows each -- histogram in LDS, exclusive scan, the rows of a bin listed (row, key) in the
// image's still unused sperm / sinv slices, and a row's rank = start of its bin + the rows of its bin
// that come before it: n x (rows per bin) comparisons instead of n x n (51 ms per 256 frames of
// 37 k rows -> well under one).  An image whose fullest bin exceeds RANK_BIN_LIMIT rows (constant
// images, synthetic ties) is left to the counting kernel below: RANK_TODO at the head of its sct slice.
constexpr int RANK_BINS = 4096;
constexpr int RANK_BIN_LIMIT = 1024;
constexpr int RANK_TODO = 0x52414E4B;

__global__ __launch_bounds__(1024) void pack3_rank_bins_kernel(Pack3Args P)
{
    __shared__ int start[RANK_BINS + 1];
    __shared__ int cursor[RANK_BINS];
    __shared__ int wsum[16];
    __shared__ int s_lo, s_hi, s_max;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = P.src_off ? P.src_off[img] : 0;
    const int n = (int)(P.src_off ? P.src_off[img + 1] - r0 : P.single_n);
    const int d0 = P.dst_off ? P.dst_off[img] : 0;
    if (n <= 0) return;
    const int32_t *key = P.n2 + r0;
    int32_t *list_row = P.sperm + d0, *list_key = P.sinv + d0;
    if (tid == 0) { s_lo = 0x7FFFFFFF; s_hi = -0x7FFFFFFF; s_max = 0; }
    for (int b = tid; b < RANK_BINS; b += 1024) cursor[b] = 0;
    __syncthreads();
    int lo = 0x7FFFFFFF, hi = -0x7FFFFFFF;
    for (int r = tid; r < n; r += 1024) {
        const int k = key[r];
        lo = min(lo, k);
        hi = max(hi, k);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        lo = min(lo, __shfl_xor(lo, m));
        hi = max(hi, __shfl_xor(hi, m));
    }
    if (lane == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    __syncthreads();
    lo = s_lo;
    int shift = 0;
    while (((int64_t)s_hi - lo) >> shift >= RANK_BINS) ++shift;
    for (int r = tid; r < n; r += 1024) atomicAdd(&cursor[(key[r] - lo) >> shift], 1);
    __syncthreads();
    // exclusive scan of the 4096 counts: 4 per thread, waves through shuffles, 16 wave sums through LDS
    int c[4], t = 0, mx = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k] = cursor[4 * tid + k];
        t += c[k];
        mx = max(mx, c[k]);
    }
    int x = t;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const int y = __shfl_up(x, sft);
        if (lane >= sft) x += y;
    }
    if (lane == 63) wsum[wave] = x;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = max(mx, __shfl_xor(mx, m));
    if (lane == 0) atomicMax(&s_max, mx);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int run = woff + x - t;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        start[4 * tid + k] = run;
        run += c[k];
    }
    if (tid == 1023) start[RANK_BINS] = run;
    __syncthreads();
    if (s_max > RANK_BIN_LIMIT) {                        // (uniform: the counting kernel takes the image)
        if (tid == 0) P.sct[d0] = RANK_TODO;
        return;
    }
    if (tid == 0) P.sct[d0] = 0;
    for (int b = tid; b < RANK_BINS; b += 1024) cursor[b] = start[b];
    __syncthreads();
    for (int r = tid; r < n; r += 1024) {
        const int k = key[r];
        const int at = atomicAdd(&cursor[(k - lo) >> shift], 1);
        list_row[at] = r;
        list_key[at] = k;
    }
    __threadfence_block();
    __syncthreads();
    for (int r = tid; r < n; r += 1024) {
        const int k = key[r], b = (k - lo) >> shift;
        const int e = start[b + 1];
        int cnt = start[b];
        for (int i = start[b]; i < e; ++i) {
            const int kj = list_key[i], j = list_row[i];
            cnt += (kj < k) || (kj == k && j < r);
        }
        P.pos[r0 + r] = cnt;
    }
}

// The counting form (every image until round 6; now the images the binning kernel leaves).  blockIdx.z splits the comparisons of a 256-row group over `gridDim.z` workgroups
// (each adds its part to pos[], zeroed by the caller): one 38 k-row frame is 150 groups x 38 k
// compares -- 150 workgroups leave most of the chip idle (1.55 ms per frame, a fifth of the
// match stage of a 128-frame survey); split 16 ways it is 0.15 ms.
__global__ __launch_bounds__(256) void pack3_rank_kernel(Pack3Args P)
{
    __shared__ __attribute__((aligned(16))) int keys[1024];
    const int img = blockIdx.y;
    const int64_t r0 = P.src_off ? P.src_off[img] : 0;
    const int n = (int)(P.src_off ? P.src_off[img + 1] - r0 : P.single_n);
    if ((int)blockIdx.x * 256 >= n) return;
    // (only the images pack3_rank_bins_kernel left: it drops RANK_TODO at the head of the image's sct slice)
    if (P.sct[P.dst_off ? P.dst_off[img] : 0] != RANK_TODO) return;
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int mine = r < n ? P.n2[r0 + r] : 0;
    const int tiles = (n + 1023) / 1024, per = (tiles + (int)gridDim.z - 1) / (int)gridDim.z;
    const int b_lo = (int)blockIdx.z * per * 1024, b_hi = min(n, b_lo + per * 1024);
    int cnt = 0;
    for (int base = b_lo; base < b_hi; base += 1024) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = base + k * 256 + threadIdx.x;
            keys[k * 256 + threadIdx.x] = j < n ? P.n2[r0 + j] : 0x7FFFFFFF;
        }
        __syncthreads();
        // rows before r with key <= mine, rows after r with key < mine (the fill value of the
        // last tile never counts)
        for (int j = 0; j < 1024; j += 4) {
            const v4i k = *reinterpret_cast<const v4i *>(&keys[j]);
            cnt += (k.x < mine) || (k.x == mine && base + j < r);
            cnt += (k.y < mine) || (k.y == mine && base + j + 1 < r);
            cnt += (k.z < mine) || (k.z == mine && base + j + 2 < r);
            cnt += (k.w < mine) || (k.w == mine && base + j + 3 < r);
        }
    }
    if (r < n) {
        if (gridDim.z == 1) P.pos[r0 + r] = cnt;
        else if (cnt) atomicAdd(&P.pos[r0 + r], cnt);
    }
}

template <typename SRC>
__global__ __launch_bounds__(256) void pack3_scatter_kernel(Pack3Args P)
{
    const int img = blockIdx.y;
    const int64_t r0 = P.src_off ? P.src_off[img] : 0;
    const int n = (int)(P.src_off ? P.src_off[img + 1] - r0 : P.single_n);
    const int d0 = P.dst_off ? P.dst_off[img] : 0;
    const int cap = (n + CHUNK - 1) / CHUNK * CHUNK;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int r = t >> 3, part = t & 7;
    if (r >= cap) return;
    if (r >= n) {                              // padding rows n .. cap-1
        *reinterpret_cast<uint4 *>(P.dst + (int64_t)(d0 + r) * D + part * 16) = make_uint4(0, 0, 0, 0);
        if (part == 0) {
            P.sn2[d0 + r] = 0;
            P.sct[d0 + r] = BIG;
            P.sperm[d0 + r] = -1;
            P.sinv[d0 + r] = -1;
        }
        return;
    }
    const SRC *p = static_cast<const SRC *>(P.src) + (r0 + r) * D + part * 16;
    unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i)
        w[i >> 2] |= (unsigned)((load_value(p, i) - IAMX_DESC_OFFSET) & 0xFF) << (8 * (i & 3));
    const int pos = d0 + P.pos[r0 + r];
    *reinterpret_cast<uint4 *>(P.dst + (int64_t)pos * D + part * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    if (part == 0) {
        P.sn2[pos] = P.n2[r0 + r];
        P.sct[pos] = P.nb[r0 + r] >> 1;
        P.sperm[pos] = r;
        P.sinv[d0 + r] = pos - d0;
    }
}

// ---------------------------------------------------------------------------------
// sweep
// ---------------------------------------------------------------------------------
// a row partial in 8 bytes (16 until round 6: the partials are a quarter of the sweep's HBM traffic,
// all of the candidate pass's, and what bounds the pairs of a round): the lower bound L and the
// two upper bounds as 16-bit offsets from it -- an offset that does not fit (the waves' minima of a
// row lie a few thousand apart) saturates and reads back as "no bound": still an upper bound.
constexpr int ROW_NO_BOUND = 0x7F000000;
__device__ __forceinline__ int pack_row_bounds(int L, int U1, int U2)
{
#if defined(IAMX_T_NOPACK)
    return U1;                           // (timing only)
#else
    // v_cvt_pk_u16_u32: both offsets (never negative) saturated to 16 bits and packed in ONE
    // instruction -- the merge sits on the critical path of its waves (one wave per SIMD: nothing
    // hides a dependent VALU chain), every instruction in it shows in the sweep's time
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 d = __builtin_amdgcn_cvt_pk_u16((unsigned)(U1 - L), (unsigned)(U2 - L));
    return (int)__builtin_bit_cast(unsigned, d);
#endif
}
__device__ __forceinline__ void unpack_row_bounds(v2i e, int &L, int &U1, int &U2)
{
    const int d1 = e.y & 0xFFFF, d2 = (int)((unsigned)e.y >> 16);
    L = e.x;
    U1 = d1 == 0xFFFF ? ROW_NO_BOUND : e.x + d1;
    U2 = d2 == 0xFFFF ? ROW_NO_BOUND : e.x + d2;
}

struct SymArgs {
    const int8_t *sdesc;
    const int32_t *sn2, *sct;
    const int32_t *img_off, *img_n;
    const int32_t *upairs;       // [n_u][2]: (B image = register resident, A image = streamed)
    const int32_t *wg_off;       // [n_u+1] scan of ceil(n_B / rows per workgroup)
    const int64_t *col_off;      // [n_u] first column-result row (sum of B caps)
    const int64_t *rowp_off;     // [n_u] first row partial (sum of workgroups x A caps)
    int32_t *col;                // [..][2]  (v1, v2)
    int32_t *rowp;               // [..][2]  (L, (U1 - L) | (U2 - L) << 16, each saturated at 0xFFFF = "no bound")
    int n_u, total_wg;
    uint8_t *colmask;            // [..] per column-result row: the groups (bit (tile & 3) + 4 * lane half)
                                 //      whose minimum is <= v2 (NULL: not wanted) -- the only train rows
                                 //      that can be the query's best or second (narrow exact stage)
};

// min over the 32 lanes of a half wave (lanes 0-31: g = 0, lanes 32-63: g = 1) of 16 registers
// in 40 VALU instructions (32 half-rate slots) instead of 80: a DPP bank mask selects QUADS
// (lane bits 3:2) and a row mask 16-lane rows, so the levels "xor 8", "xor 4" and "xor 16" can
// keep a DIFFERENT register in the two halves they combine (a transposing butterfly: 16 -> 8
// -> 4 -> 2 registers); only the two levels inside a quad run on every remaining register.
// On return every lane of quad (b4 = lane bit 4, b3, b2) holds
//     m0 = min over the half wave of r[4*b4 + 2*b2 + b3],   m1 = ... of r[8 + 4*b4 + 2*b2 + b3].
// FUSED: the two masked levels as v_min_i32_dpp (DPP source and minimum in ONE instruction, the
// bank mask leaves the other quads' value in place: 2 instead of 3 instructions per register
// pair and no copies), written as inline assembly -- the compiler only forms v_min_dpp where
// the mask is full.  A DPP read needs two wait states after a VALU write of the register; the
// hazard recogniser does not look into inline assembly, hence the s_nop in front of each level
// (inside a level no instruction reads through DPP what an earlier one of the level wrote).
// `lo`: added to the four registers that are left after the two levels inside a 16-lane row
// (lanes l, l^4, l^8, l^12 hold the same `lo`, see GROUPLO in the sweep): 4 adds instead of 16.
template <bool FUSED>
__device__ __forceinline__ void half_wave_min16(int (&r)[16], int lo, int &m0, int &m1)
{
    int u[4], w[2];
    if constexpr (FUSED) {
        asm("s_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_min_i32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_min_i32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_min_i32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_min_i32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_min_i32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_min_i32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_min_i32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_min_i32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_min_i32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_min_i32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_min_i32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_min_i32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_min_i32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_min_i32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_min_i32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc"
            : "+v"(r[0]), "+v"(r[2]), "+v"(r[4]), "+v"(r[6]), "+v"(r[8]), "+v"(r[10]), "+v"(r[12]), "+v"(r[14])
            : "v"(r[1]), "v"(r[3]), "v"(r[5]), "v"(r[7]), "v"(r[9]), "v"(r[11]), "v"(r[13]), "v"(r[15]));
        // (quads 0,1 now hold min r[2k], quads 2,3 min r[2k+1], in the register of r[2k])
        asm("s_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
            "v_min_i32_dpp %0, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_min_i32_dpp %1, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
            "v_min_i32_dpp %1, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_min_i32_dpp %2, %2, %2 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
            "v_min_i32_dpp %2, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
            "v_min_i32_dpp %3, %3, %3 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
            "v_min_i32_dpp %3, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xa"
            : "+v"(r[0]), "+v"(r[4]), "+v"(r[8]), "+v"(r[12])
            : "v"(r[2]), "v"(r[6]), "v"(r[10]), "v"(r[14]));
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = r[4 * k] + lo;
    } else {
        int s[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {          // lanes xor 8: quads 0,1 keep r[2k], quads 2,3 r[2k+1]
            const int a = r[2 * k], b = r[2 * k + 1];
            const int t1 = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xF, 0x3, false);   // row_ror:8
            const int t2 = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xF, 0xC, false);
            s[k] = min(t1, t2);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {          // lanes xor 4: quads 0,2 keep s[2k], quads 1,3 s[2k+1]
            const int a = s[2 * k], b = s[2 * k + 1];
            const int t1 = __builtin_amdgcn_update_dpp(b, a, 0x12C, 0xF, 0x5, false);   // row_ror:12 = lane+4
            const int t2 = __builtin_amdgcn_update_dpp(a, b, 0x124, 0xF, 0xA, false);   // row_ror:4  = lane-4
            u[k] = min(t1, t2) + lo;
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {          // lanes xor 16: v_permlane16_swap exchanges the odd rows
        // of its first operand with the even rows of its second: even rows keep u[2k], odd u[2k+1]
        const v2u x = __builtin_amdgcn_permlane16_swap((unsigned)u[2 * k], (unsigned)u[2 * k + 1], false, false);
        w[k] = min((int)x[0], (int)x[1]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {          // inside the quads
        w[k] = min(w[k], __builtin_amdgcn_update_dpp(0, w[k], 0xB1, 0xF, 0xF, true));    // quad_perm:[1,0,3,2]
        w[k] = min(w[k], __builtin_amdgcn_update_dpp(0, w[k], 0x4E, 0xF, 0xF, true));    // quad_perm:[2,3,0,1]
    }
    m0 = w[0];
    m1 = w[1];
}

// VARIANT != 0: timing ablations, compiled only with -DIAMX_ABLATE (tools/knn2sym_ablate.py):
// bit0 no column direction, bit1 no row direction, bit2 row direction without the cross-lane
// butterfly, bit3 no MFMA.  Results are meaningless.
// PIPE: the MFMAs of a pair of query blocks are issued while the minima of the previous pair are
// taken (software pipeline across the 8 steps of a chunk, sched_group_barrier interleave).
// MERGEW: waves that share the per-chunk row merge (CHUNK / MERGEW rows each); SLEEP: s_sleep
// argument for the second half of the waves at the start of every chunk (de-phasing experiments)
// GROUPLO: the Cq floor is shared by the four lanes l, l^4, l^8, l^12 (they hold 4 QW consecutive
// sorted rows) and added after the two butterfly levels inside a 16-lane row instead of riding in
// the C operand; with it the fused butterfly (half_wave_min16<true>).
template <int QW, int NW, int VARIANT = 0, int PIPE = 0, int MERGEW = 2, int SLEEP = 0, bool GROUPLO = false,
          int CH = 128, int WPE = 2>
__global__ __launch_bounds__(NW * 64, WPE) void knn2sym_kernel(SymArgs A)
{
    constexpr int CHUNK = CH;        // train rows per LDS stage (experiments: 256)
    constexpr int WGROWS = NW * QW * 32;
    constexpr int NT = NW * 64;
    constexpr int PIECES = CHUNK * D / 16 / NT;
    __shared__ __attribute__((aligned(16))) int8_t lds[2 * CHUNK * D + 2 * CHUNK * 4 + 2 * NW * CHUNK * 4 + 2 * NW * 4 + NW * 256 * 4];   // (lds_cq: NW used)
    int8_t *lds_tile = lds;
    int *lds_tb = reinterpret_cast<int *>(lds + 2 * CHUNK * D);      // [2][CHUNK]     Ct
    int *lds_row = lds_tb + 2 * CHUNK;                               // [2][NW][CHUNK] R_w
    int *lds_cq = lds_row + 2 * NW * CHUNK;                          // [NW] (2 NW reserved) S_w
    int *lds_dump = lds_cq + 2 * NW;                                 // [NW][256]      unused stores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, g = lane >> 5;

    int vid;
    {
        const int total = A.total_wg, bid = blockIdx.x;
        const int xcd = bid & 7, k = bid >> 3, q = total >> 3, r = total & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    int lo = 0, hi = A.n_u;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (A.wg_off[mid] <= vid) lo = mid; else hi = mid;
    }
    const int u = lo;
    const int bimg = A.upairs[2 * u], aimg = A.upairs[2 * u + 1];
    const int boff = A.img_off[bimg], nb = A.img_n[bimg];
    const int aoff = A.img_off[aimg], na = A.img_n[aimg];
    const int capA = (na + CHUNK - 1) / CHUNK * CHUNK;
    const int nchunks = capA / CHUNK;
    const int wgi = vid - A.wg_off[u];
    const int q0 = wgi * WGROWS + wave * (QW * 32);
    const bool wave_valid = q0 < nb;
    const int64_t rbase = A.rowp_off[u] + (int64_t)wgi * capA;

    // B operand: lane (c, g) holds bytes [32s+16g, +16) of the QW consecutive (sorted) B rows
    // q0 + QW*c + qb; rows past the end repeat the last row (it belongs to this wave, so the
    // group minimum is unchanged).  Cq of a lane's rows lies in [lo_lane, lo_lane + spread]:
    // lo_lane rides into the sweep with the C operand, the wave's largest spread S_w (a few
    // hundred for SIFT-like rows, the rows are sorted) widens the upper bound afterwards.
    // (GROUPLO: lane c holds the row quadruple rc, c's bits 3:2 moved to the bottom, so that
    //  the lanes that differ in bits 3:2 hold 4 QW CONSECUTIVE rows)
    const int rc = GROUPLO ? ((c & 16) | ((c & 3) << 2) | ((c >> 2) & 3)) : c;
    v4i bq[QW][4];
#pragma unroll
    for (int qb = 0; qb < QW; ++qb) {
        int row = q0 + QW * rc + qb;
        row = row < nb ? row : nb - 1;
        const v4i *src = reinterpret_cast<const v4i *>(A.sdesc + (int64_t)(boff + row) * D);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bq[qb][s] = ~src[2 * s + g];
            // one wave per SIMD (512 registers): the B operand -- only ever an MFMA source -- belongs
            // in the AGPR half, the accumulators the VALU reads in the VGPR half; left alone the
            // allocator does the opposite and pays a v_accvgpr_read per accumulator element
            if constexpr (WPE == 1) asm volatile("" : "+a"(bq[qb][s]));
        }
    }
    int lo_lane = 0;
    {
        int spread = 0;
        if (wave_valid) {
            const int g_lo = GROUPLO ? (rc & ~3) : rc, g_hi = GROUPLO ? (rc | 3) : rc;
            const int r_lo = q0 + QW * g_lo < nb ? q0 + QW * g_lo : nb - 1;
            const int r_hi = q0 + QW * g_hi + QW - 1 < nb ? q0 + QW * g_hi + QW - 1 : nb - 1;
            lo_lane = A.sn2[boff + r_lo] >> 1;
            spread = (A.sn2[boff + r_hi] >> 1) - lo_lane;
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) spread = max(spread, __shfl_xor(spread, sh));
        if (lane == 0) lds_cq[wave] = spread;
    }
    if (!wave_valid) {
#pragma unroll
        for (int k = 0; k < 2 * CHUNK / 64; ++k)
            lds_row[((k >> 1) * NW + wave) * CHUNK + (k & 1) * 64 + lane] = BIG;
    }
    int m[QW][4];
#pragma unroll
    for (int qb = 0; qb < QW; ++qb)
#pragma unroll
        for (int k = 0; k < 4; ++k) m[qb][k] = BIG;

    const int8_t *tbase = A.sdesc + (int64_t)aoff * D;
    const int32_t *tci = A.sct + aoff;
    // global -> LDS directly; the XOR swizzle of the 16-byte slots is applied on the source side
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
    // per train row: the waves' group minima -> (L, U1, U2)
    constexpr int MROWS = CHUNK / MERGEW;                // rows a merging wave takes
    const int mrow = wave * MROWS + lane;                // (valid for wave < MERGEW, lane < MROWS)
    const bool merger = wave < MERGEW && lane < MROWS;
    // (the waves' spreads S_w, read once behind the first barrier into scalar registers: the merge
    //  sits on the critical path of its waves -- one wave per SIMD, nothing hides its LDS reads)
    int spread_w[NW];
    auto merge_rows = [&](int ch, int buf) {
        const int tid = mrow;
        int L = BIG, U1 = BIG, U2 = BIG;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int R = lds_row[(buf * NW + w) * CHUNK + tid];
            const int lw = R, uw = R + spread_w[w];
            L = min(L, lw);
            U2 = min(max(U1, uw), U2);
            U1 = min(U1, uw);
        }
        *reinterpret_cast<v2i *>(A.rowp + 2 * (rbase + ch * CHUNK + tid)) = v2i{L, pack_row_bounds(L, U1, U2)};
    };

    stage_direct(0, 0);
    wait_direct();
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) spread_w[w] = __builtin_amdgcn_readfirstlane(lds_cq[w]);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks && (!(VARIANT & 64) || ch == 0)) stage_direct(ch + 1, buf ^ 1);
        if constexpr (SLEEP > 0)
            if (wave >= NW / 2) __builtin_amdgcn_s_sleep(SLEEP);
        if constexpr (!(VARIANT & 16))
            if (ch > 0 && merger) merge_rows(ch - 1, buf ^ 1);
        if (wave_valid) {
            const int8_t *tile_base = lds_tile + buf * (CHUNK * D);
            const int *tb_base = lds_tb + buf * CHUNK;
            auto load_ops = [&](int tile, v4i (&a)[4], v4i (&tbv)[4]) {
                const int r = tile * 32 + c, swz = (r >> 1) & 7;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    a[s] = *reinterpret_cast<const v4i *>(tile_base + r * D + (((2 * s + g) ^ swz) * 16));
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    tbv[k] = *reinterpret_cast<const v4i *>(tb_base + tile * 32 + 8 * k + 4 * g);
            };
            int *row_dst = (lane & 3) == 0
                ? lds_row + (buf * NW + wave) * CHUNK + 8 * ((lane >> 4) & 1) + 4 * g + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1)
                : lds_dump + wave * 256 + lane;                 // + tile*32 (+16) stays inside [0, 256)
            if constexpr (PIPE != 0) {
                constexpr int PP = QW / 2, NS = (CHUNK / 32) * PP;
                v4i aop[2][4], tbop[2][4];
                v16i accs[2][2];
                int r[16];
                auto issue = [&](int t, int qp, v16i &acc0, v16i &acc1) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) acc0[reg] = acc1[reg] = tbop[t & 1][reg >> 2][reg & 3];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aop[t & 1][s], bq[qp][s], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aop[t & 1][s], bq[qp + 1][s], acc1, 0, 0, 0);
                    }
                    if constexpr (WPE == 1) asm volatile("" : "+v"(acc0), "+v"(acc1));
                };
                load_ops(0, aop[0], tbop[0]);
                if constexpr (!GROUPLO) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) tbop[0][k] += lo_lane;
                }
                issue(0, 0, accs[0][0], accs[0][1]);
                load_ops(1, aop[1], tbop[1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 0; st < NS; ++st) {
                    const int t = st / PP, qp = 2 * (st % PP), cur = st & 1;
                    if (st + 1 < NS) {
                        const int t1 = (st + 1) / PP, qp1 = 2 * ((st + 1) % PP);
                        if (!GROUPLO && t1 != t) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) tbop[t1 & 1][k] += lo_lane;
                        }
                        issue(t1, qp1, accs[cur ^ 1][0], accs[cur ^ 1][1]);
                        if (t1 != t && t1 + 1 < CHUNK / 32) load_ops(t1 + 1, aop[t & 1], tbop[t & 1]);
                    }
                    const v16i acc0 = accs[cur][0], acc1 = accs[cur][1];
                    int t0 = min(min(m[qp][t & 3], acc0[0]), acc0[1]);
                    int t1m = min(min(m[qp + 1][t & 3], acc1[0]), acc1[1]);
#pragma unroll
                    for (int reg = 2; reg < 16; reg += 2) {
                        t0 = min(min(t0, acc0[reg]), acc0[reg + 1]);
                        t1m = min(min(t1m, acc1[reg]), acc1[reg + 1]);
                    }
                    m[qp][t & 3] = t0;
                    m[qp + 1][t & 3] = t1m;
                    if (qp == 0) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) r[reg] = min(acc0[reg], acc1[reg]);
                    } else {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) r[reg] = min(min(r[reg], acc0[reg]), acc1[reg]);
                    }
                    if (qp == QW - 2) {
                        int m0, m1;
                        half_wave_min16<GROUPLO>(r, GROUPLO ? lo_lane : 0, m0, m1);
                        int *dst = row_dst + t * 32;
                        dst[0] = m0;
                        dst[16] = m1;
                    }
                    if (st + 1 < NS) {
                        // one MFMA, then the VALU work that fits beside it
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, PIPE, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            v4i a[4], tbv[4];
            load_ops(0, a, tbv);
#pragma unroll
            for (int tile = 0; tile < CHUNK / 32; ++tile) {
                v4i a_nx[4], tb_nx[4];
                if (tile + 1 < CHUNK / 32) {
                    if constexpr (VARIANT & 256) {      // (timing only: operands reused, no LDS reads)
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2) { a_nx[s2] = a[s2]; a_nx[s2][0] += 1; tb_nx[s2] = tbv[s2]; }
                    } else {
                        load_ops(tile + 1, a_nx, tb_nx);
                    }
                }
                int r[16];
                static_assert(QW % 2 == 0, "query blocks are processed in pairs");
                int cl[16];                    // C operand: Ct of the 16 rows + the lane's Cq floor
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) cl[reg] = tbv[reg >> 2][reg & 3] + (GROUPLO ? 0 : lo_lane);
#pragma unroll
                for (int qp = 0; qp < QW; qp += 2) {
                    // two independent accumulator chains in flight (C operand = Ct of the rows)
                    v16i acc0, acc1;
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) acc0[reg] = acc1[reg] = cl[reg];
                    if constexpr (VARIANT & 128) {      // (experiment: MFMA bursts at raised priority)
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_setprio(3);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if constexpr (VARIANT & 8) {
                            acc0[s] += a[s][0] ^ bq[qp][s][1];
                            acc1[s] += a[s][1] ^ bq[qp + 1][s][1];
                        } else {
                            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[qp][s], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[qp + 1][s], acc1, 0, 0, 0);
                        }
                    }
                    if constexpr (VARIANT & 128) {
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_setprio(0);
                    }
                    if constexpr (VARIANT & 1) {
                        asm volatile("" ::"v"(acc0), "v"(acc1));
                    } else {
                        // column direction: one of four interleaved running minima per query
                        int t0 = min(min(m[qp][tile & 3], acc0[0]), acc0[1]);
                        int t1 = min(min(m[qp + 1][tile & 3], acc1[0]), acc1[1]);
#pragma unroll
                        for (int reg = 2; reg < 16; reg += 2) {
                            t0 = min(min(t0, acc0[reg]), acc0[reg + 1]);
                            t1 = min(min(t1, acc1[reg]), acc1[reg + 1]);
                        }
                        m[qp][tile & 3] = t0;
                        m[qp + 1][tile & 3] = t1;
                    }
                    // row direction: minimum over the wave's query blocks, per accumulator register
                    if constexpr (VARIANT & 2) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) r[reg] = 0;
                    } else if (qp == 0) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) r[reg] = min(acc0[reg], acc1[reg]);
                    } else {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) r[reg] = min(min(r[reg], acc0[reg]), acc1[reg]);
                    }
                    __builtin_amdgcn_sched_barrier(0);     // keep the pairs apart: two chains live
                }
                int m0, m1;
                if constexpr (VARIANT & 6) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) asm volatile("" ::"v"(r[reg]));
                    m0 = r[0];
                    m1 = r[1];
                } else {
                    half_wave_min16<GROUPLO>(r, GROUPLO ? lo_lane : 0, m0, m1);
                }
                {
                    // accumulator register reg of lane half g is tile row 8*(reg>>2) + 4*g + (reg&3);
                    // this quad holds reg = 4*b4 + 2*b2 + b3 (m0) and 8 + that (m1).  One lane per
                    // quad stores them; the others store into a dump area (no branch in the tile
                    // loop: a branch here makes the compiler sink the column minima of all four
                    // tiles below it and keep 128 accumulator registers alive)
                    int *dst = row_dst + tile * 32;
                    dst[0] = m0;
                    dst[16] = m1;
                }
                if (tile + 1 < CHUNK / 32) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) { a[s] = a_nx[s]; tbv[s] = tb_nx[s]; }
                }
                __builtin_amdgcn_sched_barrier(0);     // keep tiles apart (no hoisting -> no spills)
            }
            }
        }
        wait_direct();
        if constexpr (!(VARIANT & 32)) __syncthreads();
    }
    if (merger) merge_rows(nchunks - 1, (nchunks - 1) & 1);

    // ---- column results: two smallest of the 4 x 2 group minima of every query, and the groups
    // whose minimum is <= the second smallest (bit k + 4 g: tiles with (tile & 3) == k, lane half g):
    // the only train rows that can be the query's best or second (narrow exact stage).  The masks of
    // a lane's QW queries travel as ONE word (4 bits each: one lane exchange, one store for QW = 8).
    if (wave_valid) {
        unsigned ownpack = 0;
        int v1s[QW], v2s[QW];
#pragma unroll
        for (int qb = 0; qb < QW; ++qb) {
            const int lo01 = min(m[qb][0], m[qb][1]), hi01 = max(m[qb][0], m[qb][1]);
            const int lo23 = min(m[qb][2], m[qb][3]), hi23 = max(m[qb][2], m[qb][3]);
            const int a1 = min(lo01, lo23), a2 = min(max(lo01, lo23), min(hi01, hi23));
            const int b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2, 32);
            const int v1 = min(a1, b1), v2 = min(max(a1, b1), min(a2, b2));
            v1s[qb] = v1;
            v2s[qb] = v2;
#if !defined(IAMX_T_NOMASK)
            const unsigned own = (m[qb][0] <= v2 ? 1u : 0u) | (m[qb][1] <= v2 ? 2u : 0u) |
                                 (m[qb][2] <= v2 ? 4u : 0u) | (m[qb][3] <= v2 ? 8u : 0u);
            ownpack |= own << (4 * qb);
#endif
        }
        const unsigned othpack = A.colmask ? (unsigned)__shfl_xor((int)ownpack, 32) : 0u;
        const int sub = GROUPLO ? 0 : lo_lane;           // (GROUPLO: the accumulators never saw it)
        const int row0 = q0 + QW * rc;
        if (g == 0) {
#pragma unroll
            for (int qb = 0; qb < QW; ++qb)
                if (row0 + qb < nb)
                    *reinterpret_cast<v2i *>(A.col + 2 * (A.col_off[u] + row0 + qb)) = v2i{v1s[qb] - sub, v2s[qb] - sub};
            if (A.colmask) {
                // byte qb = own nibble | other half's nibble << 4
                uint8_t *dst = A.colmask + A.col_off[u] + row0;
                if (QW == 8 && row0 + QW <= nb) {
                    auto spread = [](unsigned nib4) {    // four nibbles -> the low nibbles of four bytes
                        const unsigned x = (nib4 | (nib4 << 8)) & 0x00FF00FFu;
                        return (x | (x << 4)) & 0x0F0F0F0Fu;
                    };
                    const unsigned lo = spread(ownpack & 0xFFFFu) | (spread(othpack & 0xFFFFu) << 4);
                    const unsigned hi = spread(ownpack >> 16) | (spread(othpack >> 16) << 4);
                    *reinterpret_cast<v2u *>(dst) = v2u{lo, hi};
                } else {
#pragma unroll
                    for (int qb = 0; qb < QW; ++qb)
                        if (row0 + qb < nb)
                            dst[qb] = (uint8_t)(((ownpack >> (4 * qb)) & 15u) | (((othpack >> (4 * qb)) & 15u) << 4));
                }
            }
        }
    }
}

#ifdef IAMX_ABLATE
// ---------------------------------------------------------------------------------
// EXPERIMENT (ablation build only; measured, not shipped: profiles/r3_knn2sym_crosschunk.txt --
// 1 % with the barrier on the chunk boundary, register spills with it anywhere else).
// The same sweep with the software pipeline carried ACROSS the chunk boundary (round 3).
// knn2sym_kernel drains its pipeline at every chunk barrier: the last epilogue of a chunk has no
// MFMAs beside it, the first MFMAs of the next chunk no epilogue, and the barrier puts the two
// waves of every SIMD back into lockstep eight steps later.  Here the A rows are staged THREE
// chunks deep, so the one barrier per chunk does not have to sit on the boundary:
//   barrier M(ch), taken by a wave somewhere in the first steps of chunk ch (step BAR_LO for the
//   first half of the waves, BAR_HI for the second: the two waves of a SIMD arrive from different
//   points of their step sequence), says
//     (1) chunk ch+1 is in LDS (every wave waited for its part of the staging it issued behind
//         M(ch-1), a whole chunk ago),
//     (2) nobody reads chunk ch-1 any more -> its buffer takes chunk ch+2,
//     (3) every wave has stored its row minima of chunk ch-1 -> waves 0/1 merge them.
//   The last step of a chunk issues the MFMAs of the next chunk's first step (operands read from
//   the next buffer behind M(ch)), so the MFMA / epilogue interleave never stops; the step behind
//   the last chunk computes on stale rows and is dropped.  The row minima rotate through three
//   buffers as well (a wave may write chunk ch+2's while the merge of chunk ch is still running).
// Arithmetic, bounds and result layout are those of knn2sym_kernel<.., GROUPLO = true>.
// ---------------------------------------------------------------------------------
template <int QW, int NW, int PIPE, int BAR_LO, int BAR_HI>
__global__ __launch_bounds__(NW * 64, 2) void knn2sym_x_kernel(SymArgs A)
{
    constexpr int WGROWS = NW * QW * 32;
    constexpr int NT = NW * 64;
    constexpr int PIECES = CHUNK * D / 16 / NT;
    constexpr int TPC = CHUNK / 32;                  // 32-row tiles per chunk
    constexpr int PP = QW / 2, NS = TPC * PP;        // steps (pairs of query blocks) per chunk
    static_assert(QW % 2 == 0 && (TPC % 2) == 0, "operand / accumulator parity continues across chunks");
    static_assert(BAR_LO <= NS - PP - 1 && BAR_HI <= NS - PP - 1, "the barrier precedes the first read of the next chunk");
    __shared__ __attribute__((aligned(16))) int8_t lds[3 * CHUNK * D + 3 * CHUNK * 4 + 3 * NW * CHUNK * 4 + 2 * NW * 4 + NW * 256 * 4];
    int8_t *lds_tile = lds;
    int *lds_tb = reinterpret_cast<int *>(lds + 3 * CHUNK * D);      // [3][CHUNK]     Ct
    int *lds_row = lds_tb + 3 * CHUNK;                               // [3][NW][CHUNK] R_w
    int *lds_cq = lds_row + 3 * NW * CHUNK;                          // [NW] (2 NW reserved) S_w
    int *lds_dump = lds_cq + 2 * NW;                                 // [NW][256]      unused stores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, g = lane >> 5;

    int vid;
    {
        const int total = A.total_wg, bid = blockIdx.x;
        const int xcd = bid & 7, k = bid >> 3, q = total >> 3, r = total & 7;
        vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    int lo = 0, hi = A.n_u;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (A.wg_off[mid] <= vid) lo = mid; else hi = mid;
    }
    const int u = lo;
    const int bimg = A.upairs[2 * u], aimg = A.upairs[2 * u + 1];
    const int boff = A.img_off[bimg], nb = A.img_n[bimg];
    const int aoff = A.img_off[aimg], na = A.img_n[aimg];
    const int capA = (na + CHUNK - 1) / CHUNK * CHUNK;
    const int nchunks = capA / CHUNK;
    const int wgi = vid - A.wg_off[u];
    const int q0 = wgi * WGROWS + wave * (QW * 32);
    const bool wave_valid = q0 < nb;
    const int64_t rbase = A.rowp_off[u] + (int64_t)wgi * capA;

    // B operand and Cq floor as in knn2sym_kernel (GROUPLO lane -> row map)
    const int rc = (c & 16) | ((c & 3) << 2) | ((c >> 2) & 3);
    v4i bq[QW][4];
#pragma unroll
    for (int qb = 0; qb < QW; ++qb) {
        int row = q0 + QW * rc + qb;
        row = row < nb ? row : nb - 1;
        const v4i *src = reinterpret_cast<const v4i *>(A.sdesc + (int64_t)(boff + row) * D);
#pragma unroll
        for (int s = 0; s < 4; ++s) bq[qb][s] = ~src[2 * s + g];
    }
    int lo_lane = 0;
    {
        int spread = 0;
        if (wave_valid) {
            const int g_lo = rc & ~3, g_hi = rc | 3;
            const int r_lo = q0 + QW * g_lo < nb ? q0 + QW * g_lo : nb - 1;
            const int r_hi = q0 + QW * g_hi + QW - 1 < nb ? q0 + QW * g_hi + QW - 1 : nb - 1;
            lo_lane = A.sn2[boff + r_lo] >> 1;
            spread = (A.sn2[boff + r_hi] >> 1) - lo_lane;
        }
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) spread = max(spread, __shfl_xor(spread, sh));
        if (lane == 0) lds_cq[wave] = spread;
    }
    if (!wave_valid) {
#pragma unroll
        for (int k = 0; k < 3 * CHUNK / 64; ++k)
            lds_row[((k / (CHUNK / 64)) * NW + wave) * CHUNK + (k % (CHUNK / 64)) * 64 + lane] = BIG;
    }
    int m[QW][4];
#pragma unroll
    for (int qb = 0; qb < QW; ++qb)
#pragma unroll
        for (int k = 0; k < 4; ++k) m[qb][k] = BIG;

    const int8_t *tbase = A.sdesc + (int64_t)aoff * D;
    const int32_t *tci = A.sct + aoff;
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
    constexpr int MROWS = CHUNK / 2;                     // waves 0 and 1 merge 64 rows each
    const int mrow = wave * MROWS + lane;
    const bool merger = wave < 2;
    auto merge_rows = [&](int ch, int buf) {
        int L = BIG, U1 = BIG, U2 = BIG;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int R = lds_row[(buf * NW + w) * CHUNK + mrow];
            const int uw = R + lds_cq[w];
            L = min(L, R);
            U2 = min(max(U1, uw), U2);
            U1 = min(U1, uw);
        }
        *reinterpret_cast<v2i *>(A.rowp + 2 * (rbase + ch * CHUNK + mrow)) = v2i{L, pack_row_bounds(L, U1, U2)};
    };
    // barrier M(ch): see the header.  b2 = buffer of chunk ch+2 = buffer of chunk ch-1
    auto mid_barrier = [&](int ch, int b2) {
        wait_direct();
        __syncthreads();
        if (ch + 2 < nchunks) stage_direct(ch + 2, b2);
        if (ch > 0 && merger) merge_rows(ch - 1, b2);
    };

    stage_direct(0, 0);
    if (nchunks > 1) stage_direct(1, 1);
    wait_direct();
    __syncthreads();

    if (!wave_valid) {
        int b2 = 2;
        for (int ch = 0; ch < nchunks; ++ch) {
            mid_barrier(ch, b2);
            b2 = b2 == 2 ? 0 : b2 + 1;
        }
    } else {
        const int bar_st = wave < NW / 2 ? BAR_LO : BAR_HI;
        auto load_tile = [&](const int8_t *tile_base, const int *tb_base, int tile, v4i (&a)[4], v4i (&tbv)[4]) {
            const int r = tile * 32 + c, swz = (r >> 1) & 7;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                a[s] = *reinterpret_cast<const v4i *>(tile_base + r * D + (((2 * s + g) ^ swz) * 16));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                tbv[k] = *reinterpret_cast<const v4i *>(tb_base + tile * 32 + 8 * k + 4 * g);
        };
        v4i aop[2][4], tbop[2][4];
        v16i accs[2][2];
        int r[16];
        auto issue = [&](int par, int qp, v16i &acc0, v16i &acc1) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc0[reg] = acc1[reg] = tbop[par][reg >> 2][reg & 3];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aop[par][s], bq[qp][s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aop[par][s], bq[qp + 1][s], acc1, 0, 0, 0);
            }
        };
        const int row_lane = 8 * ((lane >> 4) & 1) + 4 * g + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
        const bool row_owner = (lane & 3) == 0;
        // prologue: the first step of chunk 0 is in flight when the loop starts
        load_tile(lds_tile, lds_tb, 0, aop[0], tbop[0]);
        issue(0, 0, accs[0][0], accs[0][1]);
        load_tile(lds_tile, lds_tb, 1, aop[1], tbop[1]);
        __builtin_amdgcn_sched_barrier(0);
        int b = 0;
        for (int ch = 0; ch < nchunks; ++ch) {
            const int b1 = b == 2 ? 0 : b + 1, b2 = b == 0 ? 2 : b - 1;
            const int8_t *tile_cur = lds_tile + b * (CHUNK * D), *tile_nxt = lds_tile + b1 * (CHUNK * D);
            const int *tb_cur = lds_tb + b * CHUNK, *tb_nxt = lds_tb + b1 * CHUNK;
            int *row_dst = row_owner ? lds_row + (b * NW + wave) * CHUNK + row_lane
                                     : lds_dump + wave * 256 + lane;     // + t*32 (+16) stays inside [0, 256)
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const int t = st / PP, qp = 2 * (st % PP), cur = st & 1;
                if (st == BAR_LO || st == BAR_HI) {
                    if (bar_st == st) mid_barrier(ch, b2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
                    // the next step's MFMAs (past the last step of the chunk: the next chunk's first)
                    const int t1 = (st + 1) / PP, qp1 = 2 * ((st + 1) % PP);
                    issue(t1 & 1, qp1, accs[cur ^ 1][0], accs[cur ^ 1][1]);
                    if (t1 != t) {
                        const int T = t1 + 1;            // operands of the tile after that one
                        if (T >= TPC) load_tile(tile_nxt, tb_nxt, T - TPC, aop[t & 1], tbop[t & 1]);
                        else load_tile(tile_cur, tb_cur, T, aop[t & 1], tbop[t & 1]);
                    }
                }
                const v16i acc0 = accs[cur][0], acc1 = accs[cur][1];
                int t0 = min(min(m[qp][t & 3], acc0[0]), acc0[1]);
                int t1m = min(min(m[qp + 1][t & 3], acc1[0]), acc1[1]);
#pragma unroll
                for (int reg = 2; reg < 16; reg += 2) {
                    t0 = min(min(t0, acc0[reg]), acc0[reg + 1]);
                    t1m = min(min(t1m, acc1[reg]), acc1[reg + 1]);
                }
                m[qp][t & 3] = t0;
                m[qp + 1][t & 3] = t1m;
                if (qp == 0) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) r[reg] = min(acc0[reg], acc1[reg]);
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) r[reg] = min(min(r[reg], acc0[reg]), acc1[reg]);
                }
                if (qp == QW - 2) {
                    int m0, m1;
                    half_wave_min16<true>(r, lo_lane, m0, m1);
                    int *dst = row_dst + t * 32;
                    dst[0] = m0;
                    dst[16] = m1;
                }
                // one MFMA, then the VALU work that fits beside it
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, PIPE, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            b = b1;
        }
    }
    wait_direct();
    __syncthreads();
    if (merger) merge_rows(nchunks - 1, (nchunks - 1) % 3);

    // ---- column results: two smallest of the 4 x 2 group minima of every query
    if (wave_valid) {
#pragma unroll
        for (int qb = 0; qb < QW; ++qb) {
            const int lo01 = min(m[qb][0], m[qb][1]), hi01 = max(m[qb][0], m[qb][1]);
            const int lo23 = min(m[qb][2], m[qb][3]), hi23 = max(m[qb][2], m[qb][3]);
            const int a1 = min(lo01, lo23), a2 = min(max(lo01, lo23), min(hi01, hi23));
            const int b1 = __shfl_xor(a1, 32), b2 = __shfl_xor(a2, 32);
            const int v1 = min(a1, b1), v2 = min(max(a1, b1), min(a2, b2));
            const int row = q0 + QW * rc + qb;
            if (g == 0 && row < nb)
                *reinterpret_cast<v2i *>(A.col + 2 * (A.col_off[u] + row)) = v2i{v1, v2};
        }
    }
}

#endif  // IAMX_ABLATE

// ---------------------------------------------------------------------------------
// candidates.  Pass 1 (symcand_rows_kernel, one thread per query row) visits the rows in SORTED
// order (the order the sweep wrote its bounds in: coalesced reads) and drops a flag at the row's
// original position; pass 2 (symcand_kernel, one workgroup per ORDERED pair) lists the flagged rows
// in ascending original order at the pair's own slice of cand_q (first entry out_off[p]: a pair
// cannot have more candidates than rows, so no scan over the pairs is needed) and appends the
// pair's tasks to the exact stage's lists (pass 1 ran inside the per-pair workgroups until round 6:
// 21 MB of row bounds per pair of 37 k-row images through ONE workgroup).
// ---------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------
// NARROW exact stage (round 6).  The sweep already knows WHERE a candidate's two smallest
// distances can be: a query that was a B row has eight group minima over the streamed image
// (group = (tile & 3, lane half): `colmask` says which are <= v2), a query that was an A row has one
// (L, U1, U2) per 1024-row block of the register-resident image (`rowp`: block w can hold a row
// at or below the upper bound of the second distance only if L_w <= U2).  Every other train row is
// farther than the candidate's exact second distance.  So a candidate becomes ITEMS (candidate,
// class), the items of an ordered pair are bucketed by class, and a task scans ONE class -- 1/8 of
// the train image for the column direction, 1/nwg for the row direction -- for up to 256 items.
// Each item leaves (best, row, second) of its class; a finish pass merges a candidate's items.
// Results are those of the full scan: every row that can be the best, the second or tied with
// either lies in a class of the candidate's mask ("narrow exact stage").
//
// One device buffer `nar` (iamx_knn2sym_narrow_bytes), carved up by narrow_layout() below.
// ---------------------------------------------------------------------------------
struct NarTask { int p, cls, item0, n; };
struct NarLayout {
    int64_t ctl, pair_base, mask, slot, items, res, tasks, total;
    int64_t item_cap, task_cap;
};
// ctl[0] narrow tasks, ctl[1] items allocated, ctl[2] tasks taken (zero on entry, reset by the finish kernel)
__host__ __device__ inline NarLayout narrow_layout(int64_t rows, int64_t n_pairs)
{
    NarLayout L;
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    L.item_cap = rows;
    L.task_cap = rows / 64 + 80 * n_pairs + 64;
    int64_t o = 0;
    L.ctl = o;        o += 256;
    L.pair_base = o;  o += up(n_pairs * 4);
    L.mask = o;       o += up(rows * 8);
    L.slot = o;       o += up(rows * 8);
    L.items = o;      o += up(L.item_cap * 8);
    L.res = o;        o += up(L.item_cap * 16);
    L.tasks = o;      o += up(L.task_cap * 16);
    L.total = o;
    return L;
}
constexpr int NAR_MIN_TRAIN = 1024;      // narrow scans only for train images of at least this many rows
constexpr int NAR_ITEMS = 256;           // items of a narrow task

struct CandArgs {
    const int32_t *sn2, *sperm, *img_off, *img_n;
    const int32_t *pairs;        // [n_pairs][2] ordered (query image, train image)
    const int32_t *osrc;         // [n_pairs][2]: unordered pair u, role (0: query = B, 1: query = A)
    const int32_t *wg_off;       // [n_u+1]
    const int64_t *col_off, *rowp_off, *out_off;
    const int32_t *col, *rowp;
    double thresh;
    uint8_t *keep;
    int32_t *cand_cnt;
    int32_t *cand_q;
    int32_t *task_total;         // [2] wave tasks / workgroup tasks appended so far (zeroed by symcompact_kernel)
    int32_t *tasks;              // [..][2] (ordered pair, block): wave tasks from entry 0, workgroup tasks from entry n_pairs
    int32_t *d2;                 // [rows][2]: [.][1] of a candidate row = upper bound of its exact second distance
    int wg_shift;                // log2 of the candidates of a workgroup task (8, or 9 for the four-set form)
    // narrow exact stage (nar == NULL: off)
    const uint8_t *colmask;
    int8_t *nar;
    int64_t rows_total;
    int n_pairs, wgrows;
};

// pass 1, one thread per query row of every ordered pair (grid: 256-row chunks of the largest
// query image x ordered pairs): bounds -> candidate flag at the row's original position, the upper
// bound of its second distance, its class mask.
__device__ __forceinline__ void symcand_rows_pair(const CandArgs &A, int p, int pos)
{
    const int qimg = A.pairs[2 * p];
    const int u = A.osrc[2 * p], role = A.osrc[2 * p + 1];
    const int soff = A.img_off[qimg], n = A.img_n[qimg];
    if (pos >= n) return;
    const int cap = (n + CHUNK - 1) / CHUNK * CHUNK;
    const int nwg = A.wg_off[u + 1] - A.wg_off[u];
    const int64_t ob = A.out_off[p];
    const int32_t *rowq = A.rowp + 2 * A.rowp_off[u];
    // narrow exact stage: classes of this ordered pair's train image (8 groups of the streamed image
    // when the query was a B row, the nwg row blocks of the register-resident image otherwise)
    const int ncls = role == 0 ? 8 : nwg;
    const bool nar_ok = A.nar != nullptr && A.img_n[A.pairs[2 * p + 1]] >= NAR_MIN_TRAIN && ncls <= 64;
    const int n2 = A.sn2[soff + pos], par = n2 & 1;
    long long Lb, Ub;
    int U2 = 0x7FFFFFFF;
    if (role == 0) {
        const v2i v = *reinterpret_cast<const v2i *>(A.col + 2 * (A.col_off[u] + pos));
        const long long cq = n2 >> 1;
        Lb = 2 * (v.x + cq) + par;
        Ub = 2 * (v.y + cq) + par + 1;
    } else {
        int L = 0x7FFFFFFF, U1 = 0x7FFFFFFF;
        for (int w = 0; w < nwg; ++w) {
            int eL, e1, e2;
            unpack_row_bounds(*reinterpret_cast<const v2i *>(rowq + 2 * ((int64_t)w * cap + pos)), eL, e1, e2);
            L = min(L, eL);
            // merge the sorted pairs (U1, U2) and (e1, e2)
            const int n1 = min(U1, e1);
            U2 = min(max(U1, e1), min(U2, e2));
            U1 = n1;
        }
        Lb = 2ll * L + par;
        Ub = 2ll * U2 + par + 1;
    }
    if (Lb < 0) Lb = 0;
    const float f0 = (float)sqrt((double)Lb);
    const float f1 = (float)sqrt((double)Ub);
    const bool k = f1 == 0.0f || (double)f0 * ((double)f0 / (double)f1) < A.thresh;
    const int orig = A.sperm[soff + pos];
    // the flag at the row's ORIGINAL position, one bit per row in a map cleared by the launch (a byte
    // per row until round 6: 33 M scattered byte writes per launch of 4096 synthetic pairs were most of
    // this kernel's time; now only the candidates write)
    if (k) atomicOr(reinterpret_cast<unsigned *>(A.keep) + ((ob + orig) >> 5), 1u << ((ob + orig) & 31));
    // what the exact stage prunes its scan with: no row farther than this can be the best or
    // the second of the candidate (replaced by the exact pair of distances there)
    if (k) A.d2[2 * (ob + orig) + 1] = (int)(Ub < 0x7FFFFFFFll ? Ub : 0x7FFFFFFFll);
    if (k && nar_ok) {
        // the classes that can hold a row at or below Ub (every other row is farther than the
        // exact second distance): group minimum <= v2, block lower bound <= U2
        unsigned long long mk = 0ull;
        if (role == 0) {
            mk = A.colmask[A.col_off[u] + pos];
        } else {
            for (int w = 0; w < nwg; ++w)
                if (rowq[2 * ((int64_t)w * cap + pos)] <= U2) mk |= 1ull << w;
        }
        const NarLayout NL = narrow_layout(A.rows_total, A.n_pairs);
        reinterpret_cast<unsigned long long *>(A.nar + NL.mask)[ob + orig] = mk;
    }
}

__global__ __launch_bounds__(256) void symcand_rows_kernel(CandArgs A)
{
    // (1024 rows per workgroup: beside the next launch's sweep -- whose workgroups own their compute
    //  units -- every workgroup of this grid waits for a unit to come free; 131 k small ones per
    //  launch of 4096 synthetic pairs cost the sweep more than the kernel's own work)
    for (int p = blockIdx.y; p < A.n_pairs; p += gridDim.y)
#pragma unroll 1
        for (int k = 0; k < 4; ++k) symcand_rows_pair(A, p, (blockIdx.x * 4 + k) * 256 + threadIdx.x);
}

// passes 2 and 3, one workgroup per ordered pair
__global__ __launch_bounds__(256) void symcand_kernel(CandArgs A)
{
    __shared__ int wcnt[4];
    __shared__ int s_base;
    const int p = blockIdx.x;
    const int qimg = A.pairs[2 * p];
    const int u = A.osrc[2 * p], role = A.osrc[2 * p + 1];
    const int n = A.img_n[qimg];
    const int nwg = A.wg_off[u + 1] - A.wg_off[u];
    const int64_t ob = A.out_off[p];
    const NarLayout NL = narrow_layout(A.rows_total, A.n_pairs);
    const int ncls = role == 0 ? 8 : nwg;
    const bool nar_ok = A.nar != nullptr && A.img_n[A.pairs[2 * p + 1]] >= NAR_MIN_TRAIN && ncls <= 64;
    unsigned long long *maskv = reinterpret_cast<unsigned long long *>(A.nar + NL.mask);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int out = 0;
    for (int base = 0; base < n; base += 256) {
        const int i = base + threadIdx.x;
        const bool k = i < n && ((reinterpret_cast<const unsigned *>(A.keep)[(ob + i) >> 5] >> ((ob + i) & 31)) & 1u);
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
        if (k) A.cand_q[ob + out + woff + before] = i;
        out += tot;
        __syncthreads();
    }
    // ---- narrow exact stage: items (candidate, class) bucketed by class, tasks of one class
    bool narrow = nar_ok && out > 64;
    if (A.nar != nullptr) {
        __shared__ int cls_cnt[64], cls_off[65], cls_task[65], s_nb;
        int32_t *ctl = reinterpret_cast<int32_t *>(A.nar + NL.ctl);
        int32_t *pair_base = reinterpret_cast<int32_t *>(A.nar + NL.pair_base);
        v2i *slot = reinterpret_cast<v2i *>(A.nar + NL.slot);
        v2i *items = reinterpret_cast<v2i *>(A.nar + NL.items);
        NarTask *ntasks = reinterpret_cast<NarTask *>(A.nar + NL.tasks);
        if (narrow) {
            if (threadIdx.x < 64) cls_cnt[threadIdx.x] = 0;
            __syncthreads();
            int run = 0;                                   // result slots handed out so far
            for (int base = 0; base < out; base += 256) {
                const int i = base + threadIdx.x;
                unsigned long long mk = 0ull;
                if (i < out) mk = maskv[ob + A.cand_q[ob + i]];
                const int nk = __popcll(mk);
                int x = nk;
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) {
                    const int y = __shfl_up(x, sft);
                    if (lane >= sft) x += y;
                }
                if (lane == 63) wcnt[wave] = x;
                __syncthreads();
                int woff = 0, tot = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (w < wave) woff += wcnt[w];
                    tot += wcnt[w];
                }
                if (i < out) {
                    slot[ob + i] = v2i{run + woff + x - nk, nk};
                    for (unsigned long long b = mk; b; b &= b - 1) atomicAdd(&cls_cnt[__ffsll((long long)b) - 1], 1);
                }
                run += tot;
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                int o = 0, nt = 0;
                for (int cidx = 0; cidx < 64; ++cidx) {
                    cls_off[cidx] = o;
                    cls_task[cidx] = nt;
                    o += cls_cnt[cidx];
                    nt += (cls_cnt[cidx] + NAR_ITEMS - 1) / NAR_ITEMS;
                }
                cls_off[64] = o;
                cls_task[64] = nt;
                // (a batch whose items or tasks do not fit falls back to the full scan, pair by pair)
                const int b0 = atomicAdd(ctl + 1, run);
                bool ok = (int64_t)b0 + run <= NL.item_cap;
                int t0 = 0;
                if (ok) {
                    t0 = atomicAdd(ctl + 0, nt);
                    ok = (int64_t)t0 + nt <= NL.task_cap;
                }
                s_nb = ok ? b0 : -1;
                s_base = t0;
                pair_base[p] = ok ? b0 : -1;
            }
            __syncthreads();
            narrow = s_nb >= 0;
            if (narrow) {
                const int b0 = s_nb, t0 = s_base;
                if (threadIdx.x < 64) {
                    const int cidx = threadIdx.x, cnt_c = cls_cnt[cidx];
                    for (int t = 0; t * NAR_ITEMS < cnt_c; ++t)
                        ntasks[t0 + cls_task[cidx] + t] =
                            NarTask{p, cidx, b0 + cls_off[cidx] + t * NAR_ITEMS, min(NAR_ITEMS, cnt_c - t * NAR_ITEMS)};
                }
                __syncthreads();
                if (threadIdx.x < 64) cls_cnt[threadIdx.x] = 0;       // now the buckets' cursors
                __syncthreads();
                for (int i = threadIdx.x; i < out; i += 256) {
                    const unsigned long long mk = maskv[ob + A.cand_q[ob + i]];
                    int sl = b0 + slot[ob + i].x;
                    for (unsigned long long b = mk; b; b &= b - 1, ++sl) {
                        const int cidx = __ffsll((long long)b) - 1;
                        const int at = atomicAdd(&cls_cnt[cidx], 1);
                        items[b0 + cls_off[cidx] + at] = v2i{i, sl};
                    }
                }
            }
            __syncthreads();
        } else if (threadIdx.x == 0) {
            pair_base[p] = -1;
        }
    }
    if (narrow) {
        if (threadIdx.x == 0) A.cand_cnt[p] = out;
        return;
    }
    // two task lists in one buffer: a pair with <= 64 candidates is ONE wave's task (entries
    // [0, n_pairs): at most one per pair), a pair with more gets workgroup tasks of 256 or 512 candidates
    // whose four waves share every train tile through LDS (entries from n_pairs on)
    const bool small = out <= 64;
    const int ntask = small ? (out > 0 ? 1 : 0) : (out + (1 << A.wg_shift) - 1) >> A.wg_shift;
    if (threadIdx.x == 0) {
        A.cand_cnt[p] = out;
        s_base = ntask ? atomicAdd(A.task_total + (small ? 0 : 1), ntask) + (small ? 0 : (int)gridDim.x) : 0;
    }
    __syncthreads();
    // (a workgroup task carries the granularity it was cut to: the exact kernels check it)
    const int tag = small ? 0 : (A.wg_shift << 24);
    for (int t = threadIdx.x; t < ntask; t += 256)
        *reinterpret_cast<v2i *>(A.tasks + 2 * (s_base + t)) = v2i{p, t | tag};
}

// ---------------------------------------------------------------------------------
// exact top-2 of the candidate rows (original-order store), metric, final keep flag.
// A TASK = up to 32 candidate rows of one ordered pair against every row of its train image,
// on the MFMA like the general kernel (match_knn2.hip: packed key = distance | train row,
// v_med3 / v_min, lowest train row wins ties), run by ONE wave on its own: the candidates are
// the B operand, the train rows come straight from L2 (16 bytes per lane and 32-row tile; all
// tasks of a pair read the same 512 KiB).  Survivors cluster in the few overlapping image pairs
// of a launch, so the tasks of all pairs form one flat list (appended to by symcand_kernel) that
// a persistent grid walks.
// ---------------------------------------------------------------------------------
struct ExactArgs {
    const int8_t *desc;          // original-order store (iamx_desc_pack_*)
    const int32_t *norm_q;       // |s|^2 per row
    const int32_t *norm_t;       // |s|^2 + 2 sum(s) per row
    const int32_t *key_t;        // norm_t * 256 + (store row & 255) per row (symexact_keys_kernel)
    int n_pairs;                 // (the workgroup tasks start at entry n_pairs of `tasks`)
    const int32_t *img_off, *img_n;
    const int32_t *pairs;
    const int64_t *out_off;      // first row of a pair in d2 AND first entry of its candidate list
    const int32_t *cand_cnt;
    const int32_t *cand_q;
    const int32_t *task_total, *tasks;
    double thresh;
    int32_t *d2;                 // [rows][2]: exact (best, second) written for the candidates
    int32_t *cand_t;
    double *cand_metric;
    uint8_t *cand_keep;
    int32_t *zero_div;
};

__device__ __forceinline__ int med3_key(int a, int b, int c)
{
    // `c` is always a compiler-generated VALU result (the packed key), never a raw MFMA
    // accumulator: no MFMA -> VALU hazard hides inside the asm statement
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// key_t[g] = norm_t[g] * 256 + (g & 255) for every row g of the store: the per-row constant of
// the packed key, so that the scan below builds a key with ONE v_lshl_add_u32 per entry
__global__ __launch_bounds__(256) void symexact_keys_kernel(const int32_t *__restrict__ norm_t,
                                                            int64_t total_rows,
                                                            int32_t *__restrict__ key_t)
{
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total_rows;
         g += (int64_t)gridDim.x * 256)
        key_t[g] = norm_t[g] * 256 + (int)(g & 255);
}

__device__ __forceinline__ int lshl9_add(int a, int b)
{
    // (a << 9) + b in one instruction; `a` is a raw MFMA accumulator here: the compiler sees a
    // plain VALU use of it and inserts the MFMA -> VALU wait states itself
    return (int)(((unsigned)a << 9) + (unsigned)b);
}

// the end of a candidate's scan: merge the two lane halves (rows +4) lexicographically by
// (distance, row), write the exact squared distances, the train row, the metric and the keep flag
__device__ __forceinline__ void exact_finish(const ExactArgs &A, int p, int64_t cb, int cnt, int qoff,
                                             int q, int k, int g, int bd1, int bi1, int bd2, int bi2)
{
    const int od1 = __shfl_xor(bd1, 32), oi1 = __shfl_xor(bi1, 32);
    const int od2 = __shfl_xor(bd2, 32), oi2 = __shfl_xor(bi2, 32);
    auto less = [](int da, int ia, int db, int ib) { return da < db || (da == db && ia < ib); };
    int f_d, f_i, s_d;
    if (less(od1, oi1, bd1, bi1)) {
        f_d = od1; f_i = oi1;
        s_d = less(bd1, bi1, od2, oi2) ? bd1 : od2;
    } else {
        f_d = bd1; f_i = bi1;
        s_d = less(od1, oi1, bd2, bi2) ? od1 : bd2;
    }
    if (g == 0 && k < cnt) {
        const int na = A.norm_q[qoff + q];
        const int d1 = f_d + na, d2 = s_d + na;
        *reinterpret_cast<v2i *>(A.d2 + 2 * (A.out_off[p] + q)) = v2i{d1, d2};
        // cv2 L2 distance = float32 sqrt; the rest in float64 like python (matcher.py:253-263)
        const float f0 = (float)sqrt((double)d1);
        const float f1 = (float)sqrt((double)d2);
        double mt;
        bool ok = false;
        if (f1 == 0.0f) {
            mt = __longlong_as_double(0x7FF8000000000000LL);    // python raises ZeroDivisionError
            atomicAdd(A.zero_div, 1);
        } else {
            mt = (double)f0 * ((double)f0 / (double)f1);
            ok = mt < A.thresh;
        }
        A.cand_t[cb + k] = f_i;
        A.cand_metric[cb + k] = mt;
        A.cand_keep[cb + k] = ok ? 1 : 0;
    }
}

// A TASK = up to 64 candidate rows of one ordered pair (two B operands: every train tile that
// comes up from L2 serves both) against every row of its train image.  The scan is written for
// the VALU, which bounds it: per 32 x 32 tile and candidate set 4 MFMAs and 16 x (key, med3, min)
// = 48 VALU instructions -- the first version spent ~200 (key from three terms, a select for the
// ragged last tile in every tile, 32 register copies for the one-tile-ahead prefetch) and ran at
// 0.5 PFLOP/s, a third of the sweep's time on synthetic descriptors but THREE TIMES the sweep on
// real frames, where a third to two thirds of a pair's rows are candidates.
// Epochs of the packed keys follow the store's row numbers (images start on multiples of 128).
__global__ __launch_bounds__(256) void symexact_kernel(ExactArgs A)
{
    constexpr int KEY_INVALID = 0x7FFFFFFF;
    const int total = A.task_total[0];
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, g = lane >> 5;
    const int nwaves = gridDim.x * 4;
    for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < total; t += nwaves) {
        // (a task is the whole wave's: its numbers live in scalar registers, the tile loop's tests
        //  are scalar branches)
        const v2i task = *reinterpret_cast<const v2i *>(A.tasks + 2 * t);
        const int p = __builtin_amdgcn_readfirstlane(task.x);
        const int tblk = __builtin_amdgcn_readfirstlane(task.y);
        const int64_t cb = A.out_off[p];
        const int cnt = __builtin_amdgcn_readfirstlane(A.cand_cnt[p]);
        const int qimg = A.pairs[2 * p], timg = A.pairs[2 * p + 1];
        const int qoff = __builtin_amdgcn_readfirstlane(A.img_off[qimg]);
        const int toff = __builtin_amdgcn_readfirstlane(A.img_off[timg]);
        const int nt = __builtin_amdgcn_readfirstlane(A.img_n[timg]);
        int kq[2], q[2];
        v4i bq[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            kq[h] = tblk * 64 + h * 32 + c;
            q[h] = A.cand_q[cb + (kq[h] < cnt ? kq[h] : cnt - 1)];
            const v4i *src = reinterpret_cast<const v4i *>(A.desc + (int64_t)(qoff + q[h]) * D);
#pragma unroll
            for (int s = 0; s < 4; ++s) bq[h][s] = ~src[2 * s + g];
        }
        const int8_t *tbase = A.desc + (int64_t)toff * D;
        const int32_t *tkey = A.key_t + toff;
        const int ntiles = (nt + 31) / 32;
        auto load_tile = [&](int tile, v4i (&a)[4], v4i (&tk)[4]) {
            // rows past the end stay inside the image's 128-row padding; their keys are masked
            const v4i *src = reinterpret_cast<const v4i *>(tbase + (int64_t)(tile * 32 + c) * D);
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = src[2 * s + g];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                tk[kk] = *reinterpret_cast<const v4i *>(tkey + tile * 32 + 8 * kk + 4 * g);
        };
        int m1[2] = {KEY_INVALID, KEY_INVALID}, m2[2] = {KEY_INVALID, KEY_INVALID};
        int bd1[2] = {KEY_INVALID, KEY_INVALID}, bi1[2] = {0, 0};
        int bd2[2] = {KEY_INVALID, KEY_INVALID}, bi2[2] = {0, 0};
        // a task at the end of a pair's list (or the only one of a pair with a few candidates --
        // every pair of bench.py's synthetic surveys) fills one candidate set only
        const bool two_sets = cnt - tblk * 64 > 32;
        auto scan_tile = [&](int tile, const v4i (&a)[4], const v4i (&tk)[4]) {
            const bool ragged = tile * 32 + 32 > nt;              // (wave uniform: the last tile only)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 1 && !two_sets) break;
                v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[h][s], acc, 0, 0, 0);
                if (!ragged) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int key = lshl9_add(acc[reg], tk[reg >> 2][reg & 3]);
                        m2[h] = med3_key(m1[h], m2[h], key);
                        m1[h] = min(m1[h], key);
                    }
                } else {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const int row = tile * 32 + 8 * (reg >> 2) + 4 * g + (reg & 3);
                        int key = lshl9_add(acc[reg], tk[reg >> 2][reg & 3]);
                        key = row < nt ? key : KEY_INVALID;
                        m2[h] = med3_key(m1[h], m2[h], key);
                        m1[h] = min(m1[h], key);
                    }
                }
            }
            // fold the packed keys of this 256-row epoch of the STORE into (distance, row) pairs
            const int grow = toff + tile * 32;                    // first store row of the tile
            if (((grow + 32) & 255) == 0 || tile == ntiles - 1) {
                const int sbase = (grow & ~255) - toff;           // epoch start as a row of the image
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int d1k = m1[h] >> 8, i1k = sbase + (m1[h] & 255);
                    const int d2k = m2[h] >> 8, i2k = sbase + (m2[h] & 255);
                    if (d1k < bd1[h]) {
                        if (d2k < bd1[h]) { bd2[h] = d2k; bi2[h] = i2k; }
                        else              { bd2[h] = bd1[h]; bi2[h] = bi1[h]; }
                        bd1[h] = d1k; bi1[h] = i1k;
                    } else if (d1k < bd2[h]) {
                        bd2[h] = d1k; bi2[h] = i1k;
                    }
                    m1[h] = m2[h] = KEY_INVALID;
                }
            }
        };
        // two register sets, tiles alternate between them: the loads of tile + 1 are in flight
        // while tile is scanned, without copying a set into the other
        v4i a0[4], k0[4], a1[4], k1[4];
        load_tile(0, a0, k0);
        for (int tile = 0; tile < ntiles; tile += 2) {
            load_tile(tile + 1 < ntiles ? tile + 1 : tile, a1, k1);
            scan_tile(tile, a0, k0);
            if (tile + 1 < ntiles) {
                load_tile(tile + 2 < ntiles ? tile + 2 : tile + 1, a0, k0);
                scan_tile(tile + 1, a1, k1);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            exact_finish(A, p, cb, cnt, qoff, q[h], kq[h], g, bd1[h], bi1[h], bd2[h], bi2[h]);
    }
}

// Workgroup form of the exact stage for pairs with MANY candidates (real frames: a third to two
// thirds of a pair's rows): a task = 256 candidates of one ordered pair, 64 per wave as two B
// operands; the four waves share every 32-row train tile through LDS -- fetched once per
// workgroup, coalesced (thread = 16 bytes of a row, a wave = 8 whole rows) instead of once per wave
// as 64 scattered 16-byte pieces (what bounded the wave form: the CU's texture addresser, not the
// VALU).  LDS rows are XOR-swizzled by 16-byte chunk (chunk j of row r sits at j ^ ((r >> 1) & 7)):
// a ds_read_b128 is served in four groups of 16 lanes -- rows {0-3, 12-15, 20-27} and {4-11, 16-19,
// 28-31} of a lane half -- and two rows share the 64 banks, so the eight row PAIRS of a group must
// land on eight different chunks; r & 7 left them two-way conflicted (26 % of the LDS cycles).  Two tile buffers, one
// barrier per tile; tile T + 1 is in flight from L2 while tile T is scanned.
//
// PRUNE: the scan is bound by the VALU (SQ counters on overlapping frames: 126 VALU instructions
// per wave and tile, 78 % of the SIMD's issue slots; profiles/r5_exact_pmc_unpruned_*.txt), and 96 of the
// 126 keep a best and a second over rows that cannot be either: symcand_kernel knows an upper
// bound U of the candidate's exact second distance (d2[.][1]).  With c = norm_t >> 1 of the train
// row as the MFMA's C operand the accumulator IS the halved distance term h (distance - norm_q =
// 2 h + parity of norm_t), a row with h > (U - norm_q) >> 1 is out, and a tile is scanned the old
// way only if some lane's minimum over its 16 accumulators (8 x v_min3) is not: ~2 rows per
// candidate, ~60 of a wave's ~1150 tiles.  Keys, ties and the result are the unpruned scan's.
//
// SUB tiles are staged per barrier (a PHASE): the waves run that many tiles on their own between
// two workgroup rendezvous.
template <bool PRUNE, int SUB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void symexact_wg_kernel(ExactArgs A)
{
    constexpr int KEY_INVALID = 0x7FFFFFFF;
    constexpr int PR = 32 * SUB;                                           // rows per phase
    __shared__ __attribute__((aligned(16))) int8_t s_tile[2][PR * D];
    __shared__ __attribute__((aligned(16))) int32_t s_key[2][PR];      // PRUNE: key_t & 511
    __shared__ __attribute__((aligned(16))) int32_t s_half[2][PR];     // PRUNE: key_t >> 9 = norm_t >> 1
    const int total = A.task_total[1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, g = lane >> 5;
    const int lrow = threadIdx.x >> 3, lchunk = threadIdx.x & 7;      // staging: 16 bytes per thread
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const v2i task = *reinterpret_cast<const v2i *>(A.tasks + 2 * ((int64_t)A.n_pairs + t));
        const int p = __builtin_amdgcn_readfirstlane(task.x);
        const int ttag = __builtin_amdgcn_readfirstlane(task.y);
        if ((ttag >> 24) != 8) {        // cut for the other form of this stage: refuse loudly
            if (threadIdx.x == 0) atomicAdd(A.zero_div + 1, 1);
            continue;
        }
        const int tblk = ttag & 0xFFFFFF;
        const int64_t cb = A.out_off[p];
        const int cnt = __builtin_amdgcn_readfirstlane(A.cand_cnt[p]);
        const int qimg = A.pairs[2 * p], timg = A.pairs[2 * p + 1];
        const int qoff = __builtin_amdgcn_readfirstlane(A.img_off[qimg]);
        const int toff = __builtin_amdgcn_readfirstlane(A.img_off[timg]);
        const int nt = __builtin_amdgcn_readfirstlane(A.img_n[timg]);
        const int k_wave = tblk * 256 + wave * 64;                     // first candidate of this wave
        const int n_sets = cnt - k_wave > 32 ? 2 : (cnt - k_wave > 0 ? 1 : 0);
        int kq[2], q[2], hmax[2];
        v4i bq[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            kq[h] = k_wave + h * 32 + c;
            q[h] = A.cand_q[cb + (kq[h] < cnt ? kq[h] : cnt - 1)];
            const v4i *src = reinterpret_cast<const v4i *>(A.desc + (int64_t)(qoff + q[h]) * D);
#pragma unroll
            for (int s = 0; s < 4; ++s) bq[h][s] = ~src[2 * s + g];
            // (read before this task's exact_finish replaces it; a padding lane repeats the
            //  pair's last candidate, which is this task's)
            hmax[h] = PRUNE ? (A.d2[2 * (cb + q[h]) + 1] - A.norm_q[qoff + q[h]]) >> 1 : 0;
        }
        bool dirty = false;                            // (wave-uniform: keys since the last fold)
        const int8_t *tbase = A.desc + (int64_t)toff * D;
        const int32_t *tkey = A.key_t + toff;
        const int ntiles = (nt + 31) / 32;
        int m1[2] = {KEY_INVALID, KEY_INVALID}, m2[2] = {KEY_INVALID, KEY_INVALID};
        int bd1[2] = {KEY_INVALID, KEY_INVALID}, bi1[2] = {0, 0};
        int bd2[2] = {KEY_INVALID, KEY_INVALID}, bi2[2] = {0, 0};
        // (rows past the end stay inside the image's 128-row padding; their keys are masked)
        // (a phase may reach past the last tile: still inside the padding, never scanned)
        auto fetch = [&](int ph, v4i (&row16)[SUB], int &key) {
#pragma unroll
            for (int j = 0; j < SUB; ++j)
                row16[j] = *reinterpret_cast<const v4i *>(tbase + (int64_t)(ph * PR + j * 32 + lrow) * D + 16 * lchunk);
            key = threadIdx.x < PR ? tkey[ph * PR + threadIdx.x] : 0;
        };
        auto stage = [&](int buf, const v4i (&row16)[SUB], int key) {
#pragma unroll
            for (int j = 0; j < SUB; ++j)
                *reinterpret_cast<v4i *>(&s_tile[buf][(j * 32 + lrow) * D + 16 * (lchunk ^ ((lrow >> 1) & 7))]) = row16[j];
            if (threadIdx.x < PR) {
                s_key[buf][threadIdx.x] = PRUNE ? key & 511 : key;
                if (PRUNE) s_half[buf][threadIdx.x] = key >> 9;
            }
        };
        v4i pre[SUB];
        int pre_key;
        const int nph = (ntiles + SUB - 1) / SUB;
        __syncthreads();                               // (the previous task's last tile is consumed)
        fetch(0, pre, pre_key);
        stage(0, pre, pre_key);
        __syncthreads();
        for (int ph = 0; ph < nph; ++ph) {
            const int buf = ph & 1;
            if (ph + 1 < nph) fetch(ph + 1, pre, pre_key);
#pragma unroll
            for (int j = 0; j < SUB; ++j) {
            const int tile = ph * SUB + j;
            if (n_sets > 0 && tile < ntiles) {
                v4i a[4], tk[4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    a[s] = *reinterpret_cast<const v4i *>(&s_tile[buf][(j * 32 + c) * D + 16 * ((2 * s + g) ^ ((c >> 1) & 7))]);
                v16i cin = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (PRUNE) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const v4i t = *reinterpret_cast<const v4i *>(&s_half[buf][j * 32 + 8 * kk + 4 * g]);
                        cin[4 * kk] = t.x; cin[4 * kk + 1] = t.y; cin[4 * kk + 2] = t.z; cin[4 * kk + 3] = t.w;
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        tk[kk] = *reinterpret_cast<const v4i *>(&s_key[buf][j * 32 + 8 * kk + 4 * g]);
                }
                const bool ragged = tile * 32 + 32 > nt;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (h == 1 && n_sets < 2) break;
                    v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], bq[h][0], cin, 0, 0, 0);
#pragma unroll
                    for (int s = 1; s < 4; ++s)
                        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[h][s], acc, 0, 0, 0);
                    if (PRUNE) {
                        const int t0 = min(min(acc[0], acc[1]), acc[2]), t1 = min(min(acc[3], acc[4]), acc[5]);
                        const int t2 = min(min(acc[6], acc[7]), acc[8]), t3 = min(min(acc[9], acc[10]), acc[11]);
                        const int t4 = min(min(acc[12], acc[13]), acc[14]);
                        const int lo = min(min(min(t0, t1), t2), min(min(t3, t4), acc[15]));
                        if (__ballot(lo <= hmax[h]) == 0ull) continue;
                        dirty = true;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            tk[kk] = *reinterpret_cast<const v4i *>(&s_key[buf][j * 32 + 8 * kk + 4 * g]);
                    }
                    if (!ragged) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int key = lshl9_add(acc[reg], tk[reg >> 2][reg & 3]);
                            m2[h] = med3_key(m1[h], m2[h], key);
                            m1[h] = min(m1[h], key);
                        }
                    } else {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int row = tile * 32 + 8 * (reg >> 2) + 4 * g + (reg & 3);
                            int key = lshl9_add(acc[reg], tk[reg >> 2][reg & 3]);
                            key = row < nt ? key : KEY_INVALID;
                            m2[h] = med3_key(m1[h], m2[h], key);
                            m1[h] = min(m1[h], key);
                        }
                    }
                }
                const int grow = toff + tile * 32;
                if ((!PRUNE || dirty) && (((grow + 32) & 255) == 0 || tile == ntiles - 1)) {
                    dirty = false;
                    const int sbase = (grow & ~255) - toff;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int d1k = m1[h] >> 8, i1k = sbase + (m1[h] & 255);
                        const int d2k = m2[h] >> 8, i2k = sbase + (m2[h] & 255);
                        if (d1k < bd1[h]) {
                            if (d2k < bd1[h]) { bd2[h] = d2k; bi2[h] = i2k; }
                            else              { bd2[h] = bd1[h]; bi2[h] = bi1[h]; }
                            bd1[h] = d1k; bi1[h] = i1k;
                        } else if (d1k < bd2[h]) {
                            bd2[h] = d1k; bi2[h] = i1k;
                        }
                        m1[h] = m2[h] = KEY_INVALID;
                    }
                }
            }
            }
            if (ph + 1 < nph) stage(buf ^ 1, pre, pre_key);
            __syncthreads();
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            exact_finish(A, p, cb, cnt, qoff, q[h], kq[h], g, bd1[h], bi1[h], bd2[h], bi2[h]);
    }
}

// The same scan with FOUR candidate sets per wave (a task = 512 candidates) at two waves per SIMD:
// a train tile read from LDS feeds 16 MFMAs instead of 8, and the four accumulator chains are
// independent, so the matrix pipe is kept busy by one wave where the two-set form needs its
// neighbours (47 % MFMA busy there: waves parked on the tile barrier and LDS reads).
template <int SUB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void symexact_wg4_kernel(ExactArgs A)
{
    constexpr int KEY_INVALID = 0x7FFFFFFF;
    constexpr int NSET = 4;
    constexpr int PR = 32 * SUB;
    __shared__ __attribute__((aligned(16))) int8_t s_tile[2][PR * D];
    __shared__ __attribute__((aligned(16))) int32_t s_key[2][PR];      // key_t & 511
    __shared__ __attribute__((aligned(16))) int32_t s_half[2][PR];     // key_t >> 9 = norm_t >> 1
    const int total = A.task_total[1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, g = lane >> 5;
    const int lrow = threadIdx.x >> 3, lchunk = threadIdx.x & 7;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const v2i task = *reinterpret_cast<const v2i *>(A.tasks + 2 * ((int64_t)A.n_pairs + t));
        const int p = __builtin_amdgcn_readfirstlane(task.x);
        const int ttag = __builtin_amdgcn_readfirstlane(task.y);
        if ((ttag >> 24) != 9) {        // cut for the other form of this stage: refuse loudly
            if (threadIdx.x == 0) atomicAdd(A.zero_div + 1, 1);
            continue;
        }
        const int tblk = ttag & 0xFFFFFF;
        const int64_t cb = A.out_off[p];
        const int cnt = __builtin_amdgcn_readfirstlane(A.cand_cnt[p]);
        const int qimg = A.pairs[2 * p], timg = A.pairs[2 * p + 1];
        const int qoff = __builtin_amdgcn_readfirstlane(A.img_off[qimg]);
        const int toff = __builtin_amdgcn_readfirstlane(A.img_off[timg]);
        const int nt = __builtin_amdgcn_readfirstlane(A.img_n[timg]);
        const int k_wave = (tblk * 4 + wave) * (32 * NSET);      // 4 waves x NSET sets of 32
        const int left = cnt - k_wave;
        const int n_sets = left <= 0 ? 0 : (left >= 32 * NSET ? NSET : (left + 31) >> 5);
        int kq[NSET], q[NSET], hmax[NSET];
        v4i bq[NSET][4];
#pragma unroll
        for (int h = 0; h < NSET; ++h) {
            kq[h] = k_wave + h * 32 + c;
            q[h] = A.cand_q[cb + (kq[h] < cnt ? kq[h] : cnt - 1)];
            const v4i *src = reinterpret_cast<const v4i *>(A.desc + (int64_t)(qoff + q[h]) * D);
#pragma unroll
            for (int s = 0; s < 4; ++s) bq[h][s] = ~src[2 * s + g];
            hmax[h] = (A.d2[2 * (cb + q[h]) + 1] - A.norm_q[qoff + q[h]]) >> 1;
        }
        bool dirty = false;
        const int8_t *tbase = A.desc + (int64_t)toff * D;
        const int32_t *tkey = A.key_t + toff;
        const int ntiles = (nt + 31) / 32;
        int m1[NSET], m2[NSET], bd1[NSET], bi1[NSET], bd2[NSET], bi2[NSET];
#pragma unroll
        for (int h = 0; h < NSET; ++h) {
            m1[h] = m2[h] = bd1[h] = bd2[h] = KEY_INVALID;
            bi1[h] = bi2[h] = 0;
        }
        auto fetch = [&](int ph, v4i (&row16)[SUB], int &key) {
#pragma unroll
            for (int j = 0; j < SUB; ++j)
                row16[j] = *reinterpret_cast<const v4i *>(tbase + (int64_t)(ph * PR + j * 32 + lrow) * D + 16 * lchunk);
            key = threadIdx.x < PR ? tkey[ph * PR + threadIdx.x] : 0;
        };
        auto stage = [&](int buf, const v4i (&row16)[SUB], int key) {
#pragma unroll
            for (int j = 0; j < SUB; ++j)
                *reinterpret_cast<v4i *>(&s_tile[buf][(j * 32 + lrow) * D + 16 * (lchunk ^ ((lrow >> 1) & 7))]) = row16[j];
            if (threadIdx.x < PR) {
                s_key[buf][threadIdx.x] = key & 511;
                s_half[buf][threadIdx.x] = key >> 9;
            }
        };
        v4i pre[SUB];
        int pre_key;
        const int nph = (ntiles + SUB - 1) / SUB;
        __syncthreads();
        fetch(0, pre, pre_key);
        stage(0, pre, pre_key);
        __syncthreads();
        for (int ph = 0; ph < nph; ++ph) {
            const int buf = ph & 1;
            if (ph + 1 < nph) fetch(ph + 1, pre, pre_key);
#pragma unroll
            for (int j = 0; j < SUB; ++j) {
                const int tile = ph * SUB + j;
                if (n_sets == 0 || tile >= ntiles) continue;
                v4i a[4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    a[s] = *reinterpret_cast<const v4i *>(&s_tile[buf][(j * 32 + c) * D + 16 * ((2 * s + g) ^ ((c >> 1) & 7))]);
                v16i cin;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const v4i th = *reinterpret_cast<const v4i *>(&s_half[buf][j * 32 + 8 * kk + 4 * g]);
                    cin[4 * kk] = th.x; cin[4 * kk + 1] = th.y; cin[4 * kk + 2] = th.z; cin[4 * kk + 3] = th.w;
                }
                // (a set past n_sets repeats the pair's last candidate: computed, never used)
                v16i acc[NSET];
#pragma unroll
                for (int h = 0; h < NSET; ++h)
                    acc[h] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], bq[h][0], cin, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < 4; ++s)
#pragma unroll
                    for (int h = 0; h < NSET; ++h)
                        acc[h] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[h][s], acc[h], 0, 0, 0);
                const bool ragged = tile * 32 + 32 > nt;
#pragma unroll
                for (int h = 0; h < NSET; ++h) {
                    const v16i &x = acc[h];
                    const int t0 = min(min(x[0], x[1]), x[2]), t1 = min(min(x[3], x[4]), x[5]);
                    const int t2 = min(min(x[6], x[7]), x[8]), t3 = min(min(x[9], x[10]), x[11]);
                    const int t4 = min(min(x[12], x[13]), x[14]);
                    const int lo = min(min(min(t0, t1), t2), min(min(t3, t4), x[15]));
                    if (h >= n_sets || __ballot(lo <= hmax[h]) == 0ull) continue;
                    dirty = true;
                    v4i tk[4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        tk[kk] = *reinterpret_cast<const v4i *>(&s_key[buf][j * 32 + 8 * kk + 4 * g]);
                    if (!ragged) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int key = lshl9_add(x[reg], tk[reg >> 2][reg & 3]);
                            m2[h] = med3_key(m1[h], m2[h], key);
                            m1[h] = min(m1[h], key);
                        }
                    } else {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int row = tile * 32 + 8 * (reg >> 2) + 4 * g + (reg & 3);
                            int key = lshl9_add(x[reg], tk[reg >> 2][reg & 3]);
                            key = row < nt ? key : KEY_INVALID;
                            m2[h] = med3_key(m1[h], m2[h], key);
                            m1[h] = min(m1[h], key);
                        }
                    }
                }
                const int grow = toff + tile * 32;
                if (dirty && (((grow + 32) & 255) == 0 || tile == ntiles - 1)) {
                    dirty = false;
                    const int sbase = (grow & ~255) - toff;
#pragma unroll
                    for (int h = 0; h < NSET; ++h) {
                        const int d1k = m1[h] >> 8, i1k = sbase + (m1[h] & 255);
                        const int d2k = m2[h] >> 8, i2k = sbase + (m2[h] & 255);
                        if (d1k < bd1[h]) {
                            if (d2k < bd1[h]) { bd2[h] = d2k; bi2[h] = i2k; }
                            else              { bd2[h] = bd1[h]; bi2[h] = bi1[h]; }
                            bd1[h] = d1k; bi1[h] = i1k;
                        } else if (d1k < bd2[h]) {
                            bd2[h] = d1k; bi2[h] = i1k;
                        }
                        m1[h] = m2[h] = KEY_INVALID;
                    }
                }
            }
            if (ph + 1 < nph) stage(buf ^ 1, pre, pre_key);
            __syncthreads();
        }
#pragma unroll
        for (int h = 0; h < NSET; ++h)
            exact_finish(A, p, cb, cnt, qoff, q[h], kq[h], g, bd1[h], bi1[h], bd2[h], bi2[h]);
    }
}

// ---------------------------------------------------------------------------------
// narrow exact stage: a task = up to 256 items (candidate, class) of one ordered pair and ONE
// class, four waves x two sets of 32 as in symexact_wg_kernel; the class's rows come from the
// SORTED store (the order the sweep's classes are defined in) as 32-row tiles through LDS:
//   row direction  (query was an A row): class w = sorted rows [w * wgrows, (w + 1) * wgrows) of the
//                  register-resident image, consecutive tiles;
//   column direction (query was a B row): class (k, g) = the rows the sweep's lanes of half g saw
//                  in tiles with (tile & 3) == k: rows 8 i + 4 g + (0..3), i = 0..3, of such a tile
//                  -- a packed tile takes its 16 rows from two of them.
// Per row: C operand = sct (halved norm term; 2^25 on rows past the end: never below a bound,
// still no overflow in the key), parity and the row's position among the lane's 16 accumulators
// (`pk`), the original row number (`idx`: what the reference breaks ties with).  The pruning test
// is symexact_wg_kernel's; a tile that passes builds keys (acc << 5 | parity << 4 | position),
// takes best and second with v_med3 / v_min and folds them at once -- the row number of the best
// from LDS -- unless some lane holds two equal distances among its 16 rows: then every row of the
// tile goes through the full (distance, original row) comparison.
// ---------------------------------------------------------------------------------
struct NarArgs {
    const int8_t *desc;          // original-order store: the candidates' rows
    const int32_t *norm_q, *img_off;
    const int32_t *pairs, *osrc;
    const int64_t *out_off;
    const int32_t *cand_q;
    const int32_t *d2;
    const int8_t *sdesc;         // sorted store: the train rows
    const int32_t *sn2, *sct, *sperm, *img_off3, *img_n;
    int8_t *nar;
    int64_t rows_total;
    int n_pairs, wgrows;
    // finish
    const int32_t *cand_cnt;
    double thresh;
    int32_t *d2w, *cand_t;
    double *cand_metric;
    uint8_t *cand_keep;
    int32_t *zero_div;
};

constexpr int NAR_HALF_INVALID = 1 << 25;
constexpr int NAR_D_INVALID = 0x7FFFFFFF;

// ABL != 0: timing ablations (IAMX_NARROW_ABL; results are meaningless): 1 = no tile passes the
// pruning test, 2 = and no MFMA
template <int WPE, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void symnarrow_kernel(NarArgs A)
{
    constexpr int SUB = 2, PR = 32 * SUB;
    __shared__ __attribute__((aligned(16))) int8_t s_tile[2][PR * D];
    __shared__ __attribute__((aligned(16))) int32_t s_half[2][PR];
    __shared__ __attribute__((aligned(16))) int32_t s_pk[2][PR];
    __shared__ __attribute__((aligned(16))) int32_t s_idx[2][PR];
    const NarLayout NL = narrow_layout(A.rows_total, A.n_pairs);
    int32_t *ctl = reinterpret_cast<int32_t *>(A.nar + NL.ctl);
    const NarTask *ntasks = reinterpret_cast<const NarTask *>(A.nar + NL.tasks);
    const v2i *items = reinterpret_cast<const v2i *>(A.nar + NL.items);
    v4i *res = reinterpret_cast<v4i *>(A.nar + NL.res);
    const int total = ctl[0];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, g = lane >> 5;
    const int lrow = threadIdx.x >> 3, lchunk = threadIdx.x & 7;      // staging: 16 bytes per thread
    // tasks are taken from a counter (ctl[2]; reset by the finish kernel): a column-direction task
    // scans 4.5 x the rows of a row-direction one, and only part of the grid is resident at a time
    __shared__ int s_task;
    for (;;) {
        __syncthreads();                               // (the previous task's last tile is consumed)
        if (threadIdx.x == 0) s_task = atomicAdd(ctl + 2, 1);
        __syncthreads();
        const int t = s_task;
        if (t >= total) break;
        const NarTask task = ntasks[t];
        const int p = __builtin_amdgcn_readfirstlane(task.p);
        const int cls = __builtin_amdgcn_readfirstlane(task.cls);
        const int item0 = __builtin_amdgcn_readfirstlane(task.item0);
        const int nit = __builtin_amdgcn_readfirstlane(task.n);
        const int64_t cb = A.out_off[p];
        const int qimg = A.pairs[2 * p], timg = A.pairs[2 * p + 1];
        const int role = __builtin_amdgcn_readfirstlane(A.osrc[2 * p + 1]);
        const int qoff = __builtin_amdgcn_readfirstlane(A.img_off[qimg]);
        const int toff = __builtin_amdgcn_readfirstlane(A.img_off3[timg]);
        const int nt = __builtin_amdgcn_readfirstlane(A.img_n[timg]);
        // class geometry: packed tile j, packed row r -> sorted position (or -1)
        int ntiles, first = 0, ck = 0, cg = 0;
        if (role != 0) {
            first = cls * A.wgrows;
            const int last = min(nt, first + A.wgrows);
            ntiles = last > first ? (last - first + 31) / 32 : 0;
        } else {
            ck = cls & 3;
            cg = cls >> 2;
            const int tt = (nt + 31) / 32;
            const int nk = tt > ck ? (tt - ck + 3) / 4 : 0;
            ntiles = (nk + 1) / 2;
        }
        auto position = [&](int j, int r) -> int {
            int pos;
            if (role != 0) {
                pos = first + j * 32 + r;
                if (pos >= first + A.wgrows) pos = nt;
            } else {
                const int T = 4 * (2 * j + (r >> 4)) + ck, rr = r & 15;
                pos = T * 32 + 8 * (rr >> 2) + 4 * cg + (rr & 3);
            }
            return pos < nt ? pos : -1;
        };
        const int k_wave = wave * 64;                                   // first item of this wave
        const int n_sets = nit - k_wave > 32 ? 2 : (nit - k_wave > 0 ? 1 : 0);
        int slot[2], hmax[2];
        bool live[2];
        v4i bq[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int it = k_wave + h * 32 + c;
            live[h] = it < nit;
            const v2i item = items[item0 + (live[h] ? it : nit - 1)];
            slot[h] = item.y;
            const int q = A.cand_q[cb + item.x];
            const v4i *src = reinterpret_cast<const v4i *>(A.desc + (int64_t)(qoff + q) * D);
#pragma unroll
            for (int s = 0; s < 4; ++s) bq[h][s] = ~src[2 * s + g];
            const int ub = A.d2[2 * (cb + q) + 1];
            int hm = (ub - A.norm_q[qoff + q]) >> 1;
            hm = hm < (1 << 24) ? hm : (1 << 24);
            hmax[h] = live[h] ? hm : -0x40000000;
        }
        const int8_t *tbase = A.sdesc + (int64_t)toff * D;
        int bd1[2] = {NAR_D_INVALID, NAR_D_INVALID}, bi1[2] = {NAR_D_INVALID, NAR_D_INVALID};
        int bd2[2] = {NAR_D_INVALID, NAR_D_INVALID};
        auto fetch = [&](int ph, v4i (&row16)[SUB], v4i &meta) {
#pragma unroll
            for (int j = 0; j < SUB; ++j) {
                const int pos = position(ph * SUB + j, lrow);
                row16[j] = pos >= 0 ? *reinterpret_cast<const v4i *>(tbase + (int64_t)pos * D + 16 * lchunk)
                                    : v4i{0, 0, 0, 0};
            }
            meta = v4i{NAR_HALF_INVALID, 0, NAR_D_INVALID, 0};
            if (threadIdx.x < PR) {
                const int r = threadIdx.x & 31;
                const int pos = position(ph * SUB + (threadIdx.x >> 5), r);
                const int regpos = ((r >> 3) << 2) | (r & 3);
                meta.y = regpos;
                if (pos >= 0) {
                    meta.x = A.sct[toff + pos];
                    meta.y = ((A.sn2[toff + pos] & 1) << 4) | regpos;
                    meta.z = A.sperm[toff + pos];
                }
            }
        };
        auto stage = [&](int buf, const v4i (&row16)[SUB], const v4i &meta) {
#pragma unroll
            for (int j = 0; j < SUB; ++j)
                *reinterpret_cast<v4i *>(&s_tile[buf][(j * 32 + lrow) * D + 16 * (lchunk ^ ((lrow >> 1) & 7))]) = row16[j];
            if (threadIdx.x < PR) {
                s_half[buf][threadIdx.x] = meta.x;
                s_pk[buf][threadIdx.x] = meta.y;
                s_idx[buf][threadIdx.x] = meta.z;
            }
        };
        v4i pre[SUB], pre_meta;
        const int nph = (ntiles + SUB - 1) / SUB;
        if (nph > 0) {
            fetch(0, pre, pre_meta);
            stage(0, pre, pre_meta);
        }
        __syncthreads();
        for (int ph = 0; ph < nph; ++ph) {
            const int buf = ph & 1;
            if (ph + 1 < nph) fetch(ph + 1, pre, pre_meta);
#pragma unroll
            for (int j = 0; j < SUB; ++j) {
                const int tile = ph * SUB + j;
                if (n_sets > 0 && tile < ntiles) {
                    v4i a[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        a[s] = *reinterpret_cast<const v4i *>(&s_tile[buf][(j * 32 + c) * D + 16 * ((2 * s + g) ^ ((c >> 1) & 7))]);
                    v16i cin;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const v4i tv = *reinterpret_cast<const v4i *>(&s_half[buf][j * 32 + 8 * kk + 4 * g]);
                        cin[4 * kk] = tv.x; cin[4 * kk + 1] = tv.y; cin[4 * kk + 2] = tv.z; cin[4 * kk + 3] = tv.w;
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (h == 1 && n_sets < 2) break;
                        v16i acc;
                        if constexpr (ABL == 2) {
                            acc = cin;
#pragma unroll
                            for (int s = 0; s < 4; ++s) acc[s] += a[s][0] ^ bq[h][s][1];
                        } else {
                            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], bq[h][0], cin, 0, 0, 0);
#pragma unroll
                            for (int s = 1; s < 4; ++s)
                                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bq[h][s], acc, 0, 0, 0);
                        }
                        const int t0 = min(min(acc[0], acc[1]), acc[2]), t1 = min(min(acc[3], acc[4]), acc[5]);
                        const int t2 = min(min(acc[6], acc[7]), acc[8]), t3 = min(min(acc[9], acc[10]), acc[11]);
                        const int t4 = min(min(acc[12], acc[13]), acc[14]);
                        const int lo = min(min(min(t0, t1), t2), min(min(t3, t4), acc[15]));
                        if (__ballot(lo <= (ABL ? -0x7FFFFFF0 : hmax[h])) == 0ull) continue;
                        v4i tk[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            tk[kk] = *reinterpret_cast<const v4i *>(&s_pk[buf][j * 32 + 8 * kk + 4 * g]);
                        int m1 = 0x7FFFFFFF, m2 = 0x7FFFFFFF;
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            const int key = (int)(((unsigned)acc[reg] << 5) + (unsigned)tk[reg >> 2][reg & 3]);
                            m2 = med3_key(m1, m2, key);
                            m1 = min(m1, key);
                        }
                        const int d1k = m1 >> 4, d2k = m2 >> 4;
                        if (__ballot(d1k == d2k) == 0ull) {
                            const int r1 = m1 & 15;
                            const int i1k = s_idx[buf][j * 32 + 8 * (r1 >> 2) + 4 * g + (r1 & 3)];
                            if (d1k < bd1[h] || (d1k == bd1[h] && i1k < bi1[h])) {
                                bd2[h] = min(bd1[h], d2k);
                                bd1[h] = d1k;
                                bi1[h] = i1k;
                            } else {
                                bd2[h] = min(bd2[h], d1k);
                            }
                        } else {
                            // two equal distances among a lane's 16 rows: every row through the full
                            // (distance, original row) comparison
#pragma unroll
                            for (int reg = 0; reg < 16; ++reg) {
                                const int dk = 2 * acc[reg] + ((tk[reg >> 2][reg & 3] >> 4) & 1);
                                const int ik = s_idx[buf][j * 32 + 8 * (reg >> 2) + 4 * g + (reg & 3)];
                                if (dk < bd1[h] || (dk == bd1[h] && ik < bi1[h])) {
                                    bd2[h] = bd1[h];
                                    bd1[h] = dk;
                                    bi1[h] = ik;
                                } else {
                                    bd2[h] = min(bd2[h], dk);
                                }
                            }
                        }
                    }
                }
            }
            if (ph + 1 < nph) stage(buf ^ 1, pre, pre_meta);
            __syncthreads();
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // the two lane halves (rows +4 of every group of eight) -> one (best, row, second)
            const int od1 = __shfl_xor(bd1[h], 32), oi1 = __shfl_xor(bi1[h], 32), od2 = __shfl_xor(bd2[h], 32);
            int f_d, f_i, s_d;
            if (od1 < bd1[h] || (od1 == bd1[h] && oi1 < bi1[h])) {
                f_d = od1; f_i = oi1;
                s_d = min(bd1[h], od2);
            } else {
                f_d = bd1[h]; f_i = bi1[h];
                s_d = min(od1, bd2[h]);
            }
            if (g == 0 && live[h] && (h == 0 || n_sets == 2)) res[slot[h]] = v4i{f_d, f_i, s_d, 0};
        }
    }
}

// merge a candidate's items, then what exact_finish does: exact squared distances, train row,
// metric, keep flag.  The pair's workgroup of symcompact_kernel runs it in front of its compaction
// (a kernel of its own until the launch count of the filter stage showed in the sweep running beside
// it); pairs the narrow stage did not take return at once.
__device__ __forceinline__ void narrow_finish_pair(const NarArgs &A, int p)
{
    const NarLayout NL = narrow_layout(A.rows_total, A.n_pairs);
    const int32_t *pair_base = reinterpret_cast<const int32_t *>(A.nar + NL.pair_base);
    const v2i *slot = reinterpret_cast<const v2i *>(A.nar + NL.slot);
    const v4i *res = reinterpret_cast<const v4i *>(A.nar + NL.res);
    if (p == 0 && threadIdx.x == 0) {                   // the scan has consumed tasks and items
        int32_t *ctl = reinterpret_cast<int32_t *>(A.nar + NL.ctl);
        ctl[0] = ctl[1] = ctl[2] = 0;
    }
    const int b0 = pair_base[p];
    if (b0 < 0) return;
    const int64_t cb = A.out_off[p];
    const int cnt = A.cand_cnt[p];
    const int qoff = A.img_off[A.pairs[2 * p]];
    for (int k = threadIdx.x; k < cnt; k += 256) {
        const v2i sl = slot[cb + k];
        int f_d = NAR_D_INVALID, f_i = NAR_D_INVALID, s_d = NAR_D_INVALID;
        for (int j = 0; j < sl.y; ++j) {
            const v4i e = res[b0 + sl.x + j];
            if (e.x < f_d || (e.x == f_d && e.y < f_i)) {
                s_d = min(f_d, e.z);
                f_d = e.x;
                f_i = e.y;
            } else {
                s_d = min(s_d, e.x);
            }
        }
        const int q = A.cand_q[cb + k];
        // (no row at all, or no second one, below 2^26: the classes of the mask hold every row that
        //  can matter, so this is a broken invariant -- flagged like an unresolved row)
        if (f_d >= (1 << 26) || s_d >= (1 << 26)) {
            atomicAdd(A.zero_div + 1, 1);
            A.cand_t[cb + k] = 0;
            A.cand_metric[cb + k] = 0.0;
            A.cand_keep[cb + k] = 0;
            continue;
        }
        const int na = A.norm_q[qoff + q];
        const int d1 = f_d + na, d2 = s_d + na;
        *reinterpret_cast<v2i *>(A.d2w + 2 * (cb + q)) = v2i{d1, d2};
        const float f0 = (float)sqrt((double)d1);
        const float f1 = (float)sqrt((double)d2);
        double mt;
        bool ok = false;
        if (f1 == 0.0f) {
            mt = __longlong_as_double(0x7FF8000000000000LL);    // python raises ZeroDivisionError
            atomicAdd(A.zero_div, 1);
        } else {
            mt = (double)f0 * ((double)f0 / (double)f1);
            ok = mt < A.thresh;
        }
        A.cand_t[cb + k] = f_i;
        A.cand_metric[cb + k] = mt;
        A.cand_keep[cb + k] = ok ? 1 : 0;
    }
}

// in-place, order-preserving compaction of every pair's candidate list to its survivors
// (no __restrict__: narrow_finish_pair writes cand_t / cand_metric / cand_keep through N first)
__global__ __launch_bounds__(256) void symcompact_kernel(const int64_t *cand_off,
                                                         const int32_t *cand_cnt,
                                                         const uint8_t *cand_keep,
                                                         int32_t *q, int32_t *t,
                                                         double *metric,
                                                         int32_t *surv_cnt,
                                                         int32_t *task_total, NarArgs N)
{
    __shared__ int wcnt[4];
    const int p = blockIdx.x;
    if (N.nar != nullptr) {
        narrow_finish_pair(N, p);
        __threadfence_block();
        __syncthreads();
    }
    const int64_t b = cand_off[p];
    const int cnt = cand_cnt[p];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int out = 0;
    for (int base = 0; base < cnt; base += 256) {
        const int i = base + threadIdx.x;
        const bool k = i < cnt && cand_keep[b + i];
        int vq = 0, vt = 0;
        double vm = 0.0;
        if (k) { vq = q[b + i]; vt = t[b + i]; vm = metric[b + i]; }
        const unsigned long long mask = __ballot(k);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave] = __popcll(mask);
        __syncthreads();                       // every read of this block precedes its writes
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) woff += wcnt[w];
            tot += wcnt[w];
        }
        if (k) {
            const int64_t o = b + out + woff + before;
            q[o] = vq; t[o] = vt; metric[o] = vm;
        }
        out += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        surv_cnt[p] = out;
        if (p == 0) task_total[0] = task_total[1] = 0;   // the exact stage has consumed the lists
    }
}

}  // namespace

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" int64_t iamx_desc3_rows_cap(int64_t n_rows)
{
    if (n_rows <= 0) return 0;
    return (n_rows + CHUNK - 1) / CHUNK * CHUNK;
}

extern "C" int iamx_knn2sym_rows_per_wg(int form)
{
    return form == 2 ? 1024 : (form == 1 ? 512 : (form == 0 ? 256 : 0));
}

template <typename SRC>
static int pack3_launch(Pack3Args P, int64_t total_rows, int max_rows, void *stream, const char *what)
{
    hipStream_t st = iamx::as_stream(stream);
    if (total_rows <= 0 || max_rows <= 0) return IAMX_OK;
    const int cap = (max_rows + CHUNK - 1) / CHUNK * CHUNK;
    hipLaunchKernelGGL(pack3_rows_kernel<SRC>, dim3((unsigned)((total_rows * 8 + 255) / 256)),
                       dim3(256), 0, st, P, total_rows);
    // (few, large images: split every row group's comparisons so that the launch fills the chip)
    const int64_t groups = (int64_t)((max_rows + 255) / 256) * P.n_img;
    int split = 1;
    while (split < 32 && groups * split < 2048 && max_rows > 1024 * split) split *= 2;
    if (split > 1) (void)hipMemsetAsync(P.pos, 0, (size_t)total_rows * sizeof(int32_t), st);
    hipLaunchKernelGGL(pack3_rank_bins_kernel, dim3((unsigned)P.n_img), dim3(1024), 0, st, P);
    hipLaunchKernelGGL(pack3_rank_kernel,
                       dim3((unsigned)((max_rows + 255) / 256), (unsigned)P.n_img, (unsigned)split),
                       dim3(256), 0, st, P);
    hipLaunchKernelGGL(pack3_scatter_kernel<SRC>,
                       dim3((unsigned)(((int64_t)cap * 8 + 255) / 256), (unsigned)P.n_img),
                       dim3(256), 0, st, P);
    return iamx::check_launch(what);
}

template <typename SRC>
static int pack3_single(const SRC *src, int64_t n_rows, int8_t *dst, int32_t *sn2, int32_t *sct,
                        int32_t *sperm, int32_t *sinv, int32_t *scratch, void *stream,
                        const char *what)
{
    if (n_rows < 0 || n_rows > (1 << 24) || (n_rows > 0 && !src) || !dst || !sn2 || !sct || !sperm ||
        !sinv || !scratch)
        return iamx::fail(IAMX_EINVAL, "%s: null pointer or bad row count", what);
    Pack3Args P{src, nullptr, n_rows, nullptr, dst, sn2, sct, sperm, sinv,
                scratch, scratch + n_rows, scratch + 2 * n_rows, 1};
    return pack3_launch<SRC>(P, n_rows, (int)n_rows, stream, what);
}

extern "C" int iamx_desc3_pack_u8(const uint8_t *src, int64_t n_rows, int8_t *dst, int32_t *sn2,
                                  int32_t *sct, int32_t *sperm, int32_t *sinv, int32_t *scratch,
                                  void *stream)
{
    return pack3_single(src, n_rows, dst, sn2, sct, sperm, sinv, scratch, stream, "iamx_desc3_pack_u8");
}

extern "C" int iamx_desc3_pack_f32(const float *src, int64_t n_rows, int8_t *dst, int32_t *sn2,
                                   int32_t *sct, int32_t *sperm, int32_t *sinv, int32_t *scratch,
                                   void *stream)
{
    return pack3_single(src, n_rows, dst, sn2, sct, sperm, sinv, scratch, stream, "iamx_desc3_pack_f32");
}

extern "C" int iamx_desc3_pack_batch_u8(const uint8_t *src, const int64_t *src_off,
                                        const int32_t *dst_off, int n_img, int64_t total_rows,
                                        int max_rows_per_image, int8_t *dst, int32_t *sn2,
                                        int32_t *sct, int32_t *sperm, int32_t *sinv,
                                        int32_t *scratch, void *stream)
{
    IAMX_REQUIRE(src && src_off && dst_off && dst && sn2 && sct && sperm && sinv && scratch,
                 "null pointer");
    IAMX_REQUIRE(n_img > 0 && total_rows >= 0 && max_rows_per_image >= 0, "bad count");
    Pack3Args P{src, src_off, 0, dst_off, dst, sn2, sct, sperm, sinv,
                scratch, scratch + total_rows, scratch + 2 * total_rows, n_img};
    return pack3_launch<uint8_t>(P, total_rows, max_rows_per_image, stream, "iamx_desc3_pack_batch_u8");
}

// template arguments of the shipped sweep, one place for the launch and for iamx_knn2sym_kernel_id()
// (form 2: eight query blocks per wave, ONE wave per SIMD, B operand in AGPRs -- see the kernel and
//  profiles/r3_knn2sym_onewave.txt; compiled with -mllvm -amdgpu-mfma-vgpr-form, csrc/build.sh)
#define SWEEP_FORM2 8, 4, 0, 5, 2, 0, true, 128, 1
#define SWEEP_FORM1 4, 4, 0, 6, 2, 0, true, 128, 2
#define SWEEP_FORM0 2, 4, 0, 6, 2, 0, true, 128, 2
#define SWEEP_STR2(...) #__VA_ARGS__
#define SWEEP_STR(...) SWEEP_STR2(__VA_ARGS__)

extern "C" const char *iamx_knn2sym_kernel_id(int form)
{
    switch (form) {
    case 2: return "knn2sym_kernel<" SWEEP_STR(SWEEP_FORM2) ">";
    case 1: return "knn2sym_kernel<" SWEEP_STR(SWEEP_FORM1) ">";
    case 0: return "knn2sym_kernel<" SWEEP_STR(SWEEP_FORM0) ">";
    default: return "";
    }
}

extern "C" int iamx_knn2sym_sweep(const int8_t *sdesc, const int32_t *sn2, const int32_t *sct,
                                  const int32_t *img_off, const int32_t *img_n,
                                  const int32_t *upairs, const int32_t *wg_off,
                                  const int64_t *col_off, const int64_t *rowp_off, int n_u,
                                  int total_wg, int form, int32_t *col, int32_t *rowp,
                                  uint8_t *colmask, void *stream)
{
    IAMX_REQUIRE(sdesc && sn2 && sct && img_off && img_n && upairs && wg_off && col_off && rowp_off &&
                     col && rowp,
                 "null pointer");
    IAMX_REQUIRE(n_u >= 0 && total_wg >= 0, "negative count");
    IAMX_REQUIRE(form >= 0 && form <= 2, "form must be 0 (256 rows), 1 (512) or 2 (1024)");
    if (n_u == 0 || total_wg == 0) return IAMX_OK;
    SymArgs a{sdesc, sn2, sct, img_off, img_n, upairs, wg_off, col_off, rowp_off, col, rowp, n_u, total_wg,
              colmask};
    const dim3 g((unsigned)total_wg);
    hipStream_t st = iamx::as_stream(stream);
    // PIPE = 6: six epilogue VALU instructions beside every MFMA of the next pair of query
    // blocks (profiles/r2_knn2sym_ablate.txt: 1.83 -> 1.78 us per image pair)
    if (form == 2) hipLaunchKernelGGL((knn2sym_kernel<SWEEP_FORM2>), g, dim3(256), 0, st, a);
    else if (form == 1) hipLaunchKernelGGL((knn2sym_kernel<SWEEP_FORM1>), g, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((knn2sym_kernel<SWEEP_FORM0>), g, dim3(256), 0, st, a);
    return iamx::check_launch("iamx_knn2sym_sweep");
}

// the workgroup form of the exact stage the two entry points below agree on
// (default: two candidate sets per wave, tasks of 256; IAMX_EXACT_SETS=4: four, tasks of 512)
// Both entry points read the environment when they are called (tests and A/B tools switch forms
// inside one process).  What ties them together is the task list itself: symcand_kernel writes
// the granularity it cut the tasks to into every workgroup task (bits 24.. of the block index),
// and the exact kernels refuse a task cut for the other form -- they raise the "unresolved rows"
// flag, which find_matches turns into an exception, instead of mapping tasks to the wrong
// candidates.
static bool exact_four_sets()
{
    const char *e = getenv("IAMX_EXACT_SETS"), *pr = getenv("IAMX_EXACT_PRUNE");
    if (pr && pr[0] == '0') return false;
    return e && e[0] == '4';
}

// IAMX_EXACT_NARROW=0: every candidate through the full scan (A/B, tests)
static bool narrow_enabled()
{
    const char *e = getenv("IAMX_EXACT_NARROW");
    return !(e && e[0] == '0');
}

extern "C" int64_t iamx_knn2sym_narrow_bytes(int64_t rows, int n_pairs)
{
    if (rows <= 0 || n_pairs <= 0) return 0;
    return narrow_layout(rows, n_pairs).total;
}

extern "C" int iamx_knn2sym_candidates(const int32_t *sn2, const int32_t *sperm,
                                       const int32_t *img_off, const int32_t *img_n,
                                       const int32_t *pairs, const int32_t *osrc,
                                       const int32_t *wg_off, const int64_t *col_off,
                                       const int64_t *rowp_off, const int64_t *out_off,
                                       const int32_t *col, const int32_t *rowp, int n_pairs,
                                       double thresh, uint8_t *keep, int32_t *cand_cnt,
                                       int32_t *cand_q, int32_t *task_total, int32_t *tasks,
                                       int32_t *d2, const uint8_t *colmask, void *nar,
                                       int64_t rows_total, int max_query_rows, int form, void *stream)
{
    IAMX_REQUIRE(sn2 && sperm && img_off && img_n && pairs && osrc && wg_off && col_off && rowp_off &&
                     out_off && col && rowp && keep && cand_cnt && cand_q && task_total && tasks && d2,
                 "null pointer");
    IAMX_REQUIRE(!nar || colmask, "narrow workspace without colmask");
    IAMX_REQUIRE(form >= 0 && form <= 2, "form must be 0, 1 or 2");
    if (n_pairs <= 0) return IAMX_OK;
    CandArgs a{sn2, sperm, img_off, img_n, pairs, osrc, wg_off, col_off, rowp_off, out_off, col, rowp,
               thresh, keep, cand_cnt, cand_q, task_total, tasks, d2, exact_four_sets() ? 9 : 8,
               colmask, narrow_enabled() ? static_cast<int8_t *>(nar) : nullptr, rows_total, n_pairs,
               iamx_knn2sym_rows_per_wg(form)};
    IAMX_REQUIRE(rows_total > 0 && rows_total < (1ll << 31), "rows_total = rows of all ordered pairs");
    IAMX_REQUIRE(max_query_rows > 0, "max_query_rows = rows of the largest query image");
    (void)hipMemsetAsync(keep, 0, (size_t)((rows_total + 31) / 32 * 4), iamx::as_stream(stream));
    hipLaunchKernelGGL(symcand_rows_kernel,
                       dim3((unsigned)((max_query_rows + 1023) / 1024), (unsigned)(n_pairs < 65535 ? n_pairs : 65535)),
                       dim3(256), 0, iamx::as_stream(stream), a);
    hipLaunchKernelGGL(symcand_kernel, dim3((unsigned)n_pairs), dim3(256), 0, iamx::as_stream(stream), a);
    return iamx::check_launch("iamx_knn2sym_candidates");
}

extern "C" int iamx_knn2sym_exact(const int8_t *desc, const int32_t *norm_q, const int32_t *norm_t,
                                  int32_t *key_t, int64_t total_rows,
                                  const int32_t *img_off, const int32_t *img_n, const int32_t *pairs,
                                  const int64_t *out_off, const int32_t *cand_cnt,
                                  int32_t *task_total, const int32_t *tasks, int32_t *cand_q,
                                  int n_pairs, double thresh, int32_t *d2, int32_t *cand_t,
                                  double *cand_metric, uint8_t *cand_keep, int32_t *surv_cnt,
                                  int32_t *zero_div, const int8_t *sdesc, const int32_t *sn2,
                                  const int32_t *sct, const int32_t *sperm, const int32_t *img_off3,
                                  const int32_t *osrc, void *nar, int64_t rows_total, int form,
                                  void *stream)
{
    IAMX_REQUIRE(desc && norm_q && norm_t && key_t && img_off && img_n && pairs && out_off && cand_cnt &&
                     task_total && tasks && cand_q && d2 && cand_t && cand_metric && cand_keep &&
                     surv_cnt && zero_div,
                 "null pointer");
    IAMX_REQUIRE(total_rows >= 0, "negative row count");
    if (n_pairs <= 0) return IAMX_OK;
    hipStream_t st = iamx::as_stream(stream);
    // (46 MB each way for the 2812-image store of configs[2]: ~20 us beside a 7 ms round; rebuilt
    //  every call, so the keys can never be stale against a re-packed store)
    if (total_rows > 0) {
        int64_t gk = (total_rows + 255) / 256;
        hipLaunchKernelGGL(symexact_keys_kernel, dim3((unsigned)(gk > 4096 ? 4096 : gk)), dim3(256), 0, st,
                           norm_t, total_rows, key_t);
    }
    ExactArgs a{desc, norm_q, norm_t, key_t, n_pairs, img_off, img_n, pairs, out_off, cand_cnt, cand_q,
                task_total, tasks, thresh, d2, cand_t, cand_metric, cand_keep, zero_div};
    hipLaunchKernelGGL(symexact_kernel, dim3(1024), dim3(256), 0, st, a);       // pairs with <= 64 candidates
    // the others, 256 per workgroup (IAMX_EXACT_PRUNE=0: the unpruned scan, for A/B and tests)
    const char *prune = getenv("IAMX_EXACT_PRUNE"), *sub = getenv("IAMX_EXACT_SUB");
    const int nsub = sub ? atoi(sub) : 2;
    if (exact_four_sets()) {
        if (nsub == 1)
            hipLaunchKernelGGL((symexact_wg4_kernel<1>), dim3(2048), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((symexact_wg4_kernel<2>), dim3(2048), dim3(256), 0, st, a);
    } else if (prune && prune[0] == '0')
        hipLaunchKernelGGL((symexact_wg_kernel<false, 1>), dim3(2048), dim3(256), 0, st, a);
    else if (nsub == 1)
        hipLaunchKernelGGL((symexact_wg_kernel<true, 1>), dim3(2048), dim3(256), 0, st, a);
    else if (nsub == 4)
        hipLaunchKernelGGL((symexact_wg_kernel<true, 4>), dim3(2048), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((symexact_wg_kernel<true, 2>), dim3(2048), dim3(256), 0, st, a);
    NarArgs nfin{};                                      // (nar == NULL: symcompact_kernel has nothing to merge)
    if (nar && narrow_enabled()) {
        // (the same switch and the same buffer as iamx_knn2sym_candidates: pairs it bucketed have
        //  no task in the two lists above)
        if (!(sdesc && sn2 && sct && sperm && img_off3 && osrc && rows_total > 0 && form >= 0 && form <= 2))
            return iamx::fail(IAMX_EINVAL, "iamx_knn2sym_exact: narrow workspace without the sorted store");
        NarArgs na{desc, norm_q, img_off, pairs, osrc, out_off, cand_q, d2, sdesc, sn2, sct, sperm, img_off3,
                   img_n, static_cast<int8_t *>(nar), rows_total, n_pairs, iamx_knn2sym_rows_per_wg(form),
                   cand_cnt, thresh, d2, cand_t, cand_metric, cand_keep, zero_div};
        const char *wpe = getenv("IAMX_NARROW_WPE");
        const char *abl = getenv("IAMX_NARROW_ABL");
        if (abl && abl[0] == '1')
            hipLaunchKernelGGL((symnarrow_kernel<3, 1>), dim3(768), dim3(256), 0, st, na);
        else if (abl && abl[0] == '2')
            hipLaunchKernelGGL((symnarrow_kernel<3, 2>), dim3(768), dim3(256), 0, st, na);
        else if (wpe && wpe[0] == '2')
            hipLaunchKernelGGL(symnarrow_kernel<2>, dim3(512), dim3(256), 0, st, na);
        else
            hipLaunchKernelGGL(symnarrow_kernel<3>, dim3(768), dim3(256), 0, st, na);
        nfin = na;
    }
    hipLaunchKernelGGL(symcompact_kernel, dim3((unsigned)n_pairs), dim3(256), 0, st, out_off,
                       cand_cnt, cand_keep, cand_q, cand_t, cand_metric, surv_cnt, task_total, nfin);
    return iamx::check_launch("iamx_knn2sym_exact");
}

#ifdef IAMX_ABLATE
// timing ablations of the 1024-row form (not part of the C ABI; tools/knn2sym_ablate.py)
extern "C" int iamxdbg_knn2sym_variant(int variant, const int8_t *sdesc, const int32_t *sn2,
                                       const int32_t *sct, const int32_t *img_off,
                                       const int32_t *img_n, const int32_t *upairs,
                                       const int32_t *wg_off, const int64_t *col_off,
                                       const int64_t *rowp_off, int n_u, int total_wg, int32_t *col,
                                       int32_t *rowp, void *stream)
{
    SymArgs a{sdesc, sn2, sct, img_off, img_n, upairs, wg_off, col_off, rowp_off, col, rowp, n_u, total_wg};
    const dim3 g((unsigned)total_wg);
    hipStream_t st = iamx::as_stream(stream);
    switch (variant) {
#define V(id) case id: hipLaunchKernelGGL((knn2sym_kernel<4, 8, id>), g, dim3(512), 0, st, a); break;
        V(0) V(1) V(2) V(3) V(4) V(5) V(8) V(11)
#undef V
    case 330: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 256 + 3, 0, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 331: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 256 + 64 + 3, 0, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 332: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 256 + 64 + 32 + 16 + 3, 0, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 320: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 128, 0, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 400: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 64, 6, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 401: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 64 + 32, 6, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 402: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 64 + 32 + 16, 6, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 403: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 64 + 3, 0, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 404: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 3, 0, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 310: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 5, 4, 0, true, 256>), g, dim3(512), 0, st, a); break;
    case 311: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 5, 4, 0, true, 128>), g, dim3(512), 0, st, a); break;
    case 300: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 6, 2, 0, true>), g, dim3(512), 0, st, a); break;
    // 8 query blocks per wave, ONE wave per SIMD (512 registers): the butterfly and the LDS operand
    // reads are shared by twice the MFMAs (5.3 instead of 6.3 VALU per MFMA), no second wave to
    // hide stalls.  Same 1024 B rows per workgroup, same tables.  (round 3: compiled, not measured)
    case 600: hipLaunchKernelGGL((knn2sym_kernel<8, 4, 0, 6, 2, 0, true, 128, 1>), g, dim3(256), 0, st, a); break;
    case 601: hipLaunchKernelGGL((knn2sym_kernel<8, 4, 0, 5, 2, 0, true, 128, 1>), g, dim3(256), 0, st, a); break;
    case 602: hipLaunchKernelGGL((knn2sym_kernel<8, 4, 0, 0, 2, 0, true, 128, 1>), g, dim3(256), 0, st, a); break;
    case 603: hipLaunchKernelGGL((knn2sym_kernel<8, 4, 0, 4, 2, 0, true, 128, 1>), g, dim3(256), 0, st, a); break;
    case 604: hipLaunchKernelGGL((knn2sym_kernel<8, 4, 0, 7, 2, 0, true, 128, 1>), g, dim3(256), 0, st, a); break;
    case 605: hipLaunchKernelGGL((knn2sym_kernel<8, 4, 0, 5, 4, 0, true, 256, 1>), g, dim3(256), 0, st, a); break;
    case 606: hipLaunchKernelGGL((knn2sym_kernel<8, 4, 0, 5, 4, 0, true, 128, 1>), g, dim3(256), 0, st, a); break;
#define X(id, pipe, lo, hi) case id: hipLaunchKernelGGL((knn2sym_x_kernel<4, 8, pipe, lo, hi>), g, dim3(512), 0, st, a); break;
        X(500, 6, 4, 4) X(501, 6, 4, 0) X(502, 6, 0, 0) X(503, 6, 5, 1) X(504, 6, 2, 2) X(505, 6, 4, 1)
        X(506, 5, 4, 0) X(507, 7, 4, 0) X(508, 6, 5, 2) X(509, 6, 3, 0)
#undef X
    case 301: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 0, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 302: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 4, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 303: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 5, 2, 0, true>), g, dim3(512), 0, st, a); break;
    case 200: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 6, 4, 0>), g, dim3(512), 0, st, a); break;
    case 201: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 6, 2, 2>), g, dim3(512), 0, st, a); break;
    case 202: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 6, 2, 5>), g, dim3(512), 0, st, a); break;
    case 203: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 6, 4, 3>), g, dim3(512), 0, st, a); break;
    case 204: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 6, 8, 0>), g, dim3(512), 0, st, a); break;
    case 16: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 16, 6>), g, dim3(512), 0, st, a); break;
    case 48: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 48, 6>), g, dim3(512), 0, st, a); break;
    case 100: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 4>), g, dim3(512), 0, st, a); break;
    case 101: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 6>), g, dim3(512), 0, st, a); break;
    case 102: hipLaunchKernelGGL((knn2sym_kernel<4, 8, 0, 8>), g, dim3(512), 0, st, a); break;
    default: return iamx::fail(IAMX_EINVAL, "unknown variant");
    }
    return iamx::check_launch("iamxdbg_knn2sym_variant");
}
#endif
