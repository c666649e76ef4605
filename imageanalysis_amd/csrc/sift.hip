// K1 -- SIFT detector + descriptor for gfx950 (MI355X).
//
// Replaces cv2.SIFT_create().detectAndCompute(scaled, None) on the reference's path
// (scripts/lib/image.py:235-237,324).  OpenCV itself is a third-party native that is not part
// of the reference tree; this is the published algorithm (Lowe 2004) with OpenCV 4.x defaults:
// 3 layers/octave, contrast 0.04, edge 10, sigma 1.6, image doubled (first octave -1), border 5,
// 36-bin orientation histogram (peak ratio 0.8), 4x4x8 descriptor, clip 0.2, x512 -> u8.
//
// Stages (all enqueued on one stream, no host round trip inside):
//   gray_up2x      BGR u8 -> gray u8 (fixed point) -> x2 bilinear -> f32                HBM
//   blur_strip     separable Gaussian, BORDER_REFLECT_101, f32, fused multiply-add taps in  HBM
//                  fixed order; one launch per level: column strips walked top to bottom with
//                  a ring of horizontally blurred rows in LDS; also writes the DoG
//   downsample                                                                          HBM
//   extrema        26-neighbour test on the DoG stack -> candidate list (atomic append)
//   refine         one thread per candidate: adjustLocalExtrema in float32 (<=5 steps, Cramer
//                  solve, contrast/edge tests) -> refined list (atomic append)
//   orient         one wave per refined candidate: calcOrientationHist (float32 terms, exact
//                  f64 LDS-atomic sums), smoothing, peaks -> keypoints (atomic append)
//   descriptor     one wave per keypoint: calcSIFTDescriptor, rotated 4x4x8 trilinear histogram
//                  in LDS (float32 terms summed exactly in float64: a lane combines the run of
//                  samples that fall into one cell in registers, f64 atomics per run), the pass
//                  ordered by (XCD stripe, level, row band), clip/normalise/quantise
//   sort           removeDuplicatedSorted: OpenCV's output order, duplicates dropped
// The Gaussian taps are explicit fused multiply-adds (one rounding per tap, `fmaf` in the CPU
// oracle) in the oracle's tap order; the other per-pixel arithmetic of the pyramid (grey scale,
// x2 resize, DoG, derivatives) uses separately rounded mul / add / sub under `#pragma clang fp
// contract(off)` -- nothing is left to the compiler's contraction choices, so the pyramids are
// bit-identical to the oracle and the keypoint sets can be compared one to one; keypoints are
// appended in nondeterministic order; iamx_sift_sort removes duplicates and puts them into
// OpenCV's output order (KeyPointsFilter::removeDuplicatedSorted).
// Pyramid traffic: 6 Gaussian + 5 DoG f32 levels per octave written once, read once
// (~469 B per detect-resolution pixel).
#include "iamx_common.h"
#include <algorithm>
#include <mutex>
#include <stdlib.h>

namespace {

// separately rounded float mul / add / sub for the pyramid (bit parity with the oracle): hipcc's
// default -ffp-contract=fast fuses a * b + c -- and __fmul_rn + __fadd_rn, which are plain
// operators underneath -- into v_fma_f32
#pragma clang fp contract(off)
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
#pragma clang fp contract(fast)

constexpr int NL = 3;                 // nOctaveLayers
constexpr int BORDER = 5;
constexpr int MAX_STEPS = 5;
constexpr int ORI_BINS = 36;
constexpr int MAX_TAPS = 33;

struct Taps {
    int r;
    float k[MAX_TAPS];
};

// Workgroup b is observed to run on XCD b % 8, each XCD with its own 4 MB L2 (MI355X_MICROARCH.md,
// workgroup dispatch): the bijective remap that hands every XCD a CONTIGUOUS range of tile numbers,
// so that tiles side by side -- which share halo columns and the 128-byte lines a tile edge cuts
// through -- are fetched into one L2 instead of two.  A placement guess: slower if wrong, never
// incorrect.  IAMX_SIFT_NO_XCD=1 (read once) switches it off for A/B measurements.
__device__ __forceinline__ int xcd_remap(int orig, int nwg, int enabled)
{
    if (!enabled) return orig;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    p = p < 0 ? -p : p;
    p %= period;
    return p >= n ? period - p : p;
}

// pixel (y, x) of the doubled gray image (cv::resize INTER_LINEAR x2 of cvtColor BGR2GRAY), from
// the h x w x ch 8-bit source
__device__ __forceinline__ float gray_up2x_at(const uint8_t *__restrict__ src, int h, int w, int ch,
                                              int y, int x)
{
    auto tap = [](int d, int n, int &i0, int &i1, float &t) {
        const float f = sub_rn(mul_rn(add_rn((float)d, 0.5f), 0.5f), 0.5f);
        i0 = (int)floorf(f);
        t = sub_rn(f, (float)i0);
        if (i0 < 0) { i0 = 0; t = 0.f; }
        if (i0 >= n - 1) { i0 = n - 1; t = 0.f; }
        i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    };
    int x0, x1, y0, y1;
    float tx, ty;
    tap(x, w, x0, x1, tx);
    tap(y, h, y0, y1, ty);
    auto gray = [&](int yy, int xx) -> float {
        const uint8_t *p = src + ((int64_t)yy * w + xx) * ch;
        if (ch == 1) return (float)p[0];
        const int v = ((int)p[0] * 1868 + (int)p[1] * 9617 + (int)p[2] * 4899 + 8192) >> 14;
        return (float)v;
    };
    const float omtx = sub_rn(1.f, tx), omty = sub_rn(1.f, ty);
    const float top = add_rn(mul_rn(gray(y0, x0), omtx), mul_rn(gray(y0, x1), tx));
    const float bot = add_rn(mul_rn(gray(y1, x0), omtx), mul_rn(gray(y1, x1), tx));
    return add_rn(mul_rn(top, omty), mul_rn(bot, ty));
}

// (Doubling inside the base blur's loads -- the doubled image never stored -- was measured in
// round 4: 126 MB less traffic per 2189x1459 frame but 190 us instead of 60 + 38 us, four byte
// gathers per source element in the blur's prefetch registers.  Not kept.)
__global__ __launch_bounds__(256) void gray_up2x_kernel(const uint8_t *__restrict__ src, int h,
                                                        int w, int ch, float *__restrict__ dst)
{
    const int W2 = 2 * w, H2 = 2 * h;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)W2 * H2) return;
    dst[i] = gray_up2x_at(src, h, w, ch, (int)(i / W2), (int)(i % W2));
}

// Separable Gaussian, BORDER_REFLECT_101.  Both passes start from 0 and add the taps in ascending
// order with ONE rounding per tap: acc = fma(v, k[t], acc), t = -r..r -- what a filter engine
// compiled for FMA hardware computes, and what the CPU oracle computes (fmaf), so the pyramid is
// bit-identical to it.  (Rounds 1-2 rounded multiply and add separately to match a numpy
// expression: twice the VALU instructions for no reference-derived reason.)
//
// One launch per level.  A workgroup owns a strip of SB_TW columns and `seg_rows` rows and walks
// down it in blocks of SB_TH rows:
//   H block k : SB_TH source rows (+ R columns on each side, reflect-101 in x and y) -> LDS,
//               horizontal pass in strips of 8 outputs with the 8 + 2R inputs in registers
//               -> a ring of SB_RING = 64 horizontally blurred rows in LDS
//   V block b : a thread owns one column and 8 rows, its 8 + 2R ring rows in registers; writes
//               the level
// Every source row of a segment is blurred horizontally once (the tile form of round 2 redid
// 2R rows per 32-row tile: 1.8x at R = 13), the source is read once (+ 2R / 64 columns of halo),
// the level is written once: 8 B per pixel and level + halo (the DoG levels are not stored:
// extrema_kernel and refine_one subtract two levels where they need one).  2R <= 32 keeps the
// window of a V block inside the ring while the next H block is already in it.
constexpr int SB_TW = 64, SB_TH = 32, SB_RING = 64, SB_STRIP = 8;

template <int R>
__global__ __launch_bounds__(256) void blur_strip_kernel(const float *__restrict__ src, int h, int w,
                                                         Taps T, float *__restrict__ dst,
                                                         int seg_rows, int xcd)
{
    static_assert(2 * R <= SB_RING - SB_TH, "the V window must fit the ring");
    const int tile = xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y, xcd);
    const int tile_x = tile % (int)gridDim.x, tile_y = tile / (int)gridDim.x;
    constexpr int ROW = SB_TW + 2 * R;                    // source elements per row of an H block
    constexpr int SW = ROW + 1;                           // (+1: odd row pitch)
    constexpr int HW = SB_TW + 1;
    // the ring holds its first SB_MIRROR rows a second time behind row SB_RING - 1, so that the
    // 8 + 2R rows a thread of the V pass reads never wrap: constant LDS offsets from one base
    constexpr int SB_MIRROR = SB_STRIP + 2 * R;
    __shared__ float S[SB_TH * SW];
    __shared__ float Hb[(SB_RING + SB_MIRROR) * HW];
    const int x0 = tile_x * SB_TW;
    const int y_begin = tile_y * seg_rows;
    const int y_end = min(y_begin + seg_rows, h);
    // Fetch map of an H block: thread t takes rows (t >> 4) and (t >> 4) + 16, columns (t & 15) +
    // 16 c -- one per-thread offset, everything else constants of the unrolled loops (round 4 dealt
    // the elements out linearly, e = t + 256 i, and redid a division by ROW and two reflect-101
    // tests per element and block: about as many VALU instructions as the taps themselves).
    constexpr int NC = (ROW + 15) / 16;                   // column steps (the last one partial)
    constexpr int PER_ = 2 * NC;
    static_assert(SB_TH == 32, "two rows per thread");
    const int f_ry = threadIdx.x >> 4, f_l = threadIdx.x & 15;
    const bool cols_inside = x0 - R >= 0 && x0 + SB_TW + R <= w;        // (uniform)
    // The source rows of H block k + 1 are fetched into registers while block k is computed
    // (global -> register -> LDS: the load latency sits behind a block's ~430 FMAs per thread).
    float pre[PER_];
    auto fetch = [&](int k) {                             // source rows [y_begin - R + SB_TH k, + SB_TH)
        const int ybase = y_begin - R + SB_TH * k;
        if (cols_inside && ybase >= 0 && ybase + SB_TH <= h) {
            // the whole rectangle lies inside the image
            const float *base = src + (int64_t)(ybase + f_ry) * w + (x0 - R + f_l);
#pragma unroll
            for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    pre[ri * NC + c] = (16 * c + 15 < ROW || f_l + 16 * c < ROW)
                                           ? base[(int64_t)(16 * ri) * w + 16 * c] : 0.f;
            return;
        }
#pragma unroll
        for (int ri = 0; ri < 2; ++ri) {
            int yy = ybase + f_ry + 16 * ri;
            yy = (yy >= 0 && yy < h) ? yy : reflect101(yy, h);
            const float *row = src + (int64_t)yy * w;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                int xx = x0 - R + f_l + 16 * c;
                xx = (xx >= 0 && xx < w) ? xx : reflect101(xx, w);
                pre[ri * NC + c] = (16 * c + 15 < ROW || f_l + 16 * c < ROW) ? row[xx] : 0.f;
            }
        }
    };
    auto stage = [&]() {                                  // registers -> S
        float *sp = S + f_ry * SW + f_l;
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (16 * c + 15 < ROW || f_l + 16 * c < ROW) sp[16 * ri * SW + 16 * c] = pre[ri * NC + c];
    };
    const int h_ry = threadIdx.x / (SB_TW / SB_STRIP), h_sx = (threadIdx.x % (SB_TW / SB_STRIP)) * SB_STRIP;
    auto hpass = [&](int k) {                             // S -> ring rows (SB_TH k + ry) & 63
        float win[SB_STRIP + 2 * R];
#pragma unroll
        for (int i = 0; i < SB_STRIP + 2 * R; ++i) win[i] = S[h_ry * SW + h_sx + i];
        const int ring_row = (SB_TH * k + h_ry) & (SB_RING - 1);
        float *hrow = Hb + ring_row * HW + h_sx;
        const bool mirror = ring_row < SB_MIRROR;
        float acc[SB_STRIP];
#pragma unroll
        for (int j = 0; j < SB_STRIP; ++j) {
            acc[j] = 0.f;
#pragma unroll
            for (int t = 0; t <= 2 * R; ++t) acc[j] = __builtin_fmaf(win[j + t], T.k[t], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < SB_STRIP; ++j) hrow[j] = acc[j];
        if (mirror) {
#pragma unroll
            for (int j = 0; j < SB_STRIP; ++j) hrow[SB_RING * HW + j] = acc[j];
        }
    };
    const int cx = threadIdx.x & (SB_TW - 1), g = threadIdx.x / SB_TW;
    const int x = x0 + cx;
    const float *vbase = Hb + (g * SB_STRIP) * HW + cx;   // ring row g * 8 of this thread's column
    const int n_blocks = (y_end - y_begin + SB_TH - 1) / SB_TH;
    fetch(0);
    stage();
    fetch(1);
    __syncthreads();
    hpass(0);
    __syncthreads();
    auto vpass = [&](int b, const float *vb) {
        // V block b: output rows y_begin + SB_TH b + 8 g + j; source row y' sits in ring row
        // (y' - (y_begin - R)) & 63, so the window of output row y starts at (y - y_begin) & 63
        // = vb's row (SB_TH b & 63, a constant of the unrolled body) + 8 g
        const int y0 = y_begin + SB_TH * b + g * SB_STRIP;
        if (x < w && y0 < y_end) {
            float win[SB_STRIP + 2 * R];
#pragma unroll
            for (int i = 0; i < SB_STRIP + 2 * R; ++i) win[i] = vb[i * HW];
            float *out = dst + (int64_t)y0 * w + x;
            float acc[SB_STRIP];
#pragma unroll
            for (int j = 0; j < SB_STRIP; ++j) {
                acc[j] = 0.f;
#pragma unroll
                for (int t = 0; t <= 2 * R; ++t) acc[j] = __builtin_fmaf(win[j + t], T.k[t], acc[j]);
            }
#pragma unroll
            for (int j = 0; j < SB_STRIP; ++j)
                if (y0 + j < y_end) out[(int64_t)j * w] = acc[j];
        }
    };
    static_assert(SB_RING == 2 * SB_TH, "the ring start of a V block alternates between two rows");
    for (int b = 0; b < n_blocks; ++b) {
        stage();                                          // source rows of H block b + 1
        if (b + 1 < n_blocks) fetch(b + 2);               // (in flight during the two passes below)
        __syncthreads();
        hpass(b + 1);
        __syncthreads();
        if (b & 1) vpass(b, vbase + SB_TH * HW); else vpass(b, vbase);
    }
}

// IAMX_DESC_FORM (read once; A/B measurements): descriptor_kernel<FORM, WAVES> -- 0 = <0, 8> (rounds 2-5),
// 1 = <1, 6>, 25 = <2, 5>, 9 = <9, 4> (TIMING ONLY: form 2 with every gather from the window centre: 429 us,
// what the kernel would take with its loads cache resident), anything else <2, 4> (shipped).  Same box, 2189 x 1459 detect image, 37 k
// keypoints (profiles/r6h_sift_desc_ab.txt): 791-824 / 653-667 / 637-646 / 614-628 us; forms 1 and 2 at
// eight waves per SIMD spill (1.8 / 2.1 ms), form 2 at six keeps 36 B of scratch in the loop (763 us).
inline int desc_form()
{
    static const int f = [] { const char *e = getenv("IAMX_DESC_FORM"); return e && e[0] ? atoi(e) : 24; }();
    return f;
}

// IAMX_DESC_SORT=0 (read once; A/B): the descriptor pass walks the keypoint list as orient_kernel left it
inline bool desc_sorted()
{
    static const bool on = [] { const char *e = getenv("IAMX_DESC_SORT"); return !(e && e[0] == '0'); }();
    return on;
}

inline int xcd_enabled()
{
    static const int on = [] { const char *e = getenv("IAMX_SIFT_NO_XCD"); return (e && e[0] == '1') ? 0 : 1; }();
    return on;
}

template <int R>
void launch_blur_strip(hipStream_t st, const float *src, int h, int w, const Taps &tp, float *dst)
{
    static_assert(256 / SB_TW * SB_STRIP == SB_TH && SB_TH * (SB_TW / SB_STRIP) == 256,
                  "thread maps of the two passes cover a block");
    const int strips = (w + SB_TW - 1) / SB_TW;
    // rows per workgroup: long segments amortise the one extra H block per segment, short ones
    // fill the chip (>= ~768 workgroups when the level is large enough)
    int seg = (int)(((int64_t)h * strips / 768 + SB_TH - 1) / SB_TH) * SB_TH;
    seg = seg < SB_TH ? SB_TH : (seg > 256 ? 256 : seg);
    hipLaunchKernelGGL(blur_strip_kernel<R>, dim3(strips, (h + seg - 1) / seg), dim3(256), 0, st, src, h,
                       w, tp, dst, seg, xcd_enabled());
}

// every radius gaussian_taps() can produce (r <= 16)
void blur_level(hipStream_t st, const float *src, int h, int w, const Taps &tp, float *dst)
{
    switch (tp.r) {
#define IAMX_BS(r) case r: launch_blur_strip<r>(st, src, h, w, tp, dst); break;
        IAMX_BS(1) IAMX_BS(2) IAMX_BS(3) IAMX_BS(4) IAMX_BS(5) IAMX_BS(6) IAMX_BS(7) IAMX_BS(8)
        IAMX_BS(9) IAMX_BS(10) IAMX_BS(11) IAMX_BS(12) IAMX_BS(13) IAMX_BS(14) IAMX_BS(15)
#undef IAMX_BS
    default: launch_blur_strip<16>(st, src, h, w, tp, dst); break;
    }
}

__global__ __launch_bounds__(256) void downsample_kernel(const float *__restrict__ src, int sw,
                                                         int dh, int dw, float *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)dw * dh) return;
    const int x = (int)(i % dw), y = (int)(i / dw);
    dst[i] = src[(int64_t)(2 * y) * sw + 2 * x];
}

struct Cand {
    int o, layer, r, c;
};

// one launch per octave: every interior pixel is tested on the NL middle DoG layers.  The DoG
// images are never stored: D_l = G_(l+1) - G_l is one float subtraction of two levels the scan
// reads anyway (six Gaussian levels once, instead of five DoG levels that the blur passes would
// have had to write first: 4 B per pixel and level less traffic, and 5 of 11 pyramid buffers less)
struct GaussStack {
    const float *g[NL + 3];
};

// One wave per strip of 62 columns (+1 halo column each side) and EXT_ROWS rows: every lane keeps
// its own column of the five DoG images on three consecutive rows in registers and loads ONE new
// row per step (5 coalesced loads per row and lane serve all NL layer tests; a wave per row
// issued 15), takes the max / min of its 9 values per layer triple, and gets the neighbouring
// columns' with two lane shifts.  val >= (<=) every one of the 26 neighbours  <=>  val >= (<=)
// the max (min) over the 3x3x3 block, which contains val.
constexpr int EXT_ROWS = 16;      // rows per wave (4 waves of a workgroup: 64 rows)
constexpr int EXT_LIST = 512;     // candidates a workgroup collects before its one global atomic

__device__ __forceinline__ void extrema_tile(const GaussStack &G, int h, int w, int o, float threshold,
                                             Cand *__restrict__ cand, int cap, int *__restrict__ count,
                                             int tile_x, int tile_y)
{
    // Candidates are ~1 % of the pixels; one global atomicAdd each on the single counter was the
    // whole cost of this kernel (~150 k same-address device-scope atomics per image = 0.75 ms).
    // A workgroup collects its candidates in LDS and reserves their slots with ONE atomic.
    __shared__ Cand s_list[EXT_LIST];
    __shared__ int s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = BORDER + tile_x * 62 - 1 + lane;
    const int cl = c < w - 1 ? c : w - 1;              // clamped: only feeds masked-out lanes
    const bool out = lane >= 1 && lane <= 62 && c < w - BORDER;
    const int r0 = BORDER + (tile_y * 4 + wave) * EXT_ROWS;
    const int r1 = min(r0 + EXT_ROWS, h - BORDER);
    if (r0 < r1) {                                     // (whole wave)
        float v[NL + 2][3];                            // rows r - 1, r, r + 1 (rolling)
        float gl[NL + 3];
        // the NL + 2 DoG values of one pixel from its NL + 3 Gaussian values
        auto dog_row = [&](int row, int slot) {
#pragma unroll
            for (int L = 0; L < NL + 3; ++L) gl[L] = G.g[L][(int64_t)row * w + cl];
#pragma unroll
            for (int L = 0; L < NL + 2; ++L) v[L][slot] = sub_rn(gl[L + 1], gl[L]);
        };
        dog_row(r0 - 1, 1);
        dog_row(r0, 2);
        for (int r = r0; r < r1; ++r) {
#pragma unroll
            for (int L = 0; L < NL + 2; ++L) {
                v[L][0] = v[L][1];
                v[L][1] = v[L][2];
            }
            dog_row(r + 1, 2);
            float cmax[NL + 2], cmin[NL + 2];
#pragma unroll
            for (int L = 0; L < NL + 2; ++L) {
                cmax[L] = fmaxf(fmaxf(v[L][0], v[L][1]), v[L][2]);
                cmin[L] = fminf(fminf(v[L][0], v[L][1]), v[L][2]);
            }
#pragma unroll
            for (int layer = 1; layer <= NL; ++layer) {
                const float val = v[layer][1];
                float mx = fmaxf(fmaxf(cmax[layer - 1], cmax[layer]), cmax[layer + 1]);
                float mn = fminf(fminf(cmin[layer - 1], cmin[layer]), cmin[layer + 1]);
                mx = fmaxf(mx, fmaxf(__shfl_up(mx, 1), __shfl_down(mx, 1)));
                mn = fminf(mn, fminf(__shfl_up(mn, 1), __shfl_down(mn, 1)));
                if (!out || !(fabsf(val) > threshold)) continue;
                const bool is_max = val > 0.f && val >= mx, is_min = val < 0.f && val <= mn;
                if (is_max || is_min) {
                    const int k = atomicAdd(&s_n, 1);
                    if (k < EXT_LIST) {
                        s_list[k] = Cand{o, layer, r, c};
                    } else {                           // (a block this dense: straight to global)
                        const int g = atomicAdd(count, 1);
                        if (g < cap) cand[g] = Cand{o, layer, r, c};
                    }
                }
            }
        }
    }
    __syncthreads();
    const int n = s_n < EXT_LIST ? s_n : EXT_LIST;
    if (threadIdx.x == 0 && n > 0) s_base = atomicAdd(count, n);
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 256) {
        const int g = s_base + k;
        if (g < cap) cand[g] = s_list[k];
    }
}

__global__ __launch_bounds__(256) void extrema_kernel(GaussStack G, int h, int w, int o,
                                                      float threshold, Cand *__restrict__ cand,
                                                      int cap, int *__restrict__ count, int xcd)
{
    const int tile = xcd_remap(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y, xcd);
    extrema_tile(G, h, w, o, threshold, cand, cap, count, tile % (int)gridDim.x, tile / (int)gridDim.x);
}

struct Pyr {
    // per octave: pointers of the 6 Gaussian levels, dims
    float *g[6];
    int h, w;
    int diag;            // (int)sqrt(w^2 + h^2) in float64 (calcSIFTDescriptor's radius clip), filled by the host
};
constexpr int MAX_OCT = 16;
struct PyrTable {
    Pyr oct[MAX_OCT];
    int n_oct;
};

// the extrema scans of SEVERAL small octaves in one launch (blockIdx.z = octave - o_first; a block
// beyond its octave's tile range exits): the four octaves behind the one-workgroup tail of the
// pyramid were four dependent launches of a few microseconds of work each, on the critical path
__global__ __launch_bounds__(256) void extrema_multi_kernel(PyrTable T, int o_first, float threshold,
                                                            Cand *__restrict__ cand, int cap,
                                                            int *__restrict__ count)
{
    const int o = o_first + (int)blockIdx.z;
    const Pyr &P = T.oct[o];
    const int h = P.h, w = P.w;
    if (!(h > 2 * BORDER && w > 2 * BORDER)) return;
    const int tiles_x = (w - 2 * BORDER + 61) / 62, tiles_y = (h - 2 * BORDER + 4 * EXT_ROWS - 1) / (4 * EXT_ROWS);
    if ((int)blockIdx.x >= tiles_x || (int)blockIdx.y >= tiles_y) return;
    GaussStack G;
#pragma unroll
    for (int i = 0; i < NL + 3; ++i) G.g[i] = P.g[i];
    extrema_tile(G, h, w, o, threshold, cand, cap, count, (int)blockIdx.x, (int)blockIdx.y);
}

// The smallest octaves (<= TAIL_PIXELS pixels per level) are pure launch latency as separate
// kernels (~12 us per level for microseconds of work, ~35 launches): ONE workgroup builds all of
// them in one launch with the current level and its horizontal pass in LDS.  Same taps, same
// ascending fused multiply-add chains as blur_strip_kernel: bit-identical levels.  Both passes
// work in strips of 8 outputs with the 8 + 2R inputs in registers (reflect-101 is applied when
// the window is loaded, not per tap -- the round-2 tail kernel spent ~12 instructions per tap on
// it and took 0.58 ms).
constexpr int TAIL_PIXELS = 12800;

struct TapSet {
    Taps t[NL + 2];              // layers 1 .. NL+2 (sigma of the incremental blurs)
};

template <int R>
__device__ __forceinline__ void tail_level(float *__restrict__ cur, float *__restrict__ hor,
                                           float *__restrict__ nxt_base, int H, int W, int P, int P2,
                                           const float *__restrict__ k /* LDS */,
                                           float *__restrict__ dst, bool make_base)
{
    // horizontal: strips of 8 consecutive outputs of a row.  Consecutive lanes take the same
    // strip of consecutive ROWS: their LDS addresses differ by the odd pitch P (conflict free;
    // lanes side by side in a row would be 8 dwords apart = 8 lanes per bank)
    const int spr = (W + 7) >> 3;
    for (int s = threadIdx.x; s < H * spr; s += 1024) {
        const int sx = (s / H) * 8, row = s - (s / H) * H;
        float win[8 + 2 * R];
#pragma unroll
        for (int i = 0; i < 8 + 2 * R; ++i) {
            int xx = sx - R + i;
            xx = (xx >= 0 && xx < W) ? xx : reflect101(xx, W);
            win[i] = cur[row * P + xx];
        }
        // (eight chains side by side, each in ascending tap order; one LDS read per tap)
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t <= 2 * R; ++t) {
            const float kt = k[t];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(win[j + t], kt, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (sx + j < W) hor[row * P + sx + j] = acc[j];
    }
    __syncthreads();
    // vertical: a thread owns one column and 8 rows (consecutive threads = consecutive columns)
    const int gpr = (H + 7) >> 3;
    for (int s = threadIdx.x; s < W * gpr; s += 1024) {
        const int gy = (s / W) * 8, col = s - (s / W) * W;
        float win[8 + 2 * R];
#pragma unroll
        for (int i = 0; i < 8 + 2 * R; ++i) {
            int yy = gy - R + i;
            yy = (yy >= 0 && yy < H) ? yy : reflect101(yy, H);
            win[i] = hor[yy * P + col];
        }
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t <= 2 * R; ++t) {
            const float kt = k[t];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(win[j + t], kt, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int y = gy + j;
            if (y < H) {
                const int i = y * W + col, il = y * P + col;
                dst[i] = acc[j];
                cur[il] = acc[j];                  // (this position is read by nobody else any more)
                // level NL is the source of the next octave (INTER_NEAREST to half the size)
                if (make_base && !(y & 1) && !(col & 1) && (y >> 1) < H / 2 && (col >> 1) < W / 2)
                    nxt_base[(y >> 1) * P2 + (col >> 1)] = acc[j];
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void pyramid_tail_kernel(PyrTable T, int o_first, TapSet TS)
{
    __shared__ float cur[TAIL_PIXELS];
    __shared__ float hor[TAIL_PIXELS];
    __shared__ float nxt[TAIL_PIXELS / 4 + 128];
    __shared__ float sk[NL + 2][MAX_TAPS];
    __shared__ int sr[NL + 2];
    // the taps go from the kernel argument to LDS with compile-time indices (indexing the
    // argument with a runtime level or tap number would put a copy of it in scratch memory)
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < NL + 2; ++q) {
            sr[q] = TS.t[q].r;
#pragma unroll
            for (int t = 0; t < MAX_TAPS; ++t) sk[q][t] = TS.t[q].k[t];
        }
    }
    __syncthreads();
    for (int o = o_first; o < T.n_oct; ++o) {
        const int H = T.oct[o].h, W = T.oct[o].w, npx = H * W;
        const int P = W | 1, P2 = (W / 2) | 1;            // odd LDS pitches
        float *g0 = T.oct[o].g[0];
        if (o == o_first) {
            const float *src = T.oct[o - 1].g[NL];
            const int sw = T.oct[o - 1].w;
            for (int i = threadIdx.x; i < npx; i += 1024) {
                const int x = i % W, y = i / W;
                const float v = src[(int64_t)(2 * y) * sw + 2 * x];
                g0[i] = v;
                cur[y * P + x] = v;
            }
        } else {
            for (int i = threadIdx.x; i < npx; i += 1024) {
                const int x = i % W, y = i / W;
                const float v = nxt[y * P + x];
                g0[i] = v;
                cur[y * P + x] = v;
            }
        }
        __syncthreads();
        for (int l = 1; l < NL + 3; ++l) {
            const float *tp = sk[l - 1];
            float *dst = T.oct[o].g[l];
            const bool mb = l == NL && o + 1 < T.n_oct;
            switch (sr[l - 1]) {
#define IAMX_TL(r) case r: tail_level<r>(cur, hor, nxt, H, W, P, P2, tp, dst, mb); break;
                IAMX_TL(1) IAMX_TL(2) IAMX_TL(3) IAMX_TL(4) IAMX_TL(5) IAMX_TL(6) IAMX_TL(7) IAMX_TL(8)
                IAMX_TL(9) IAMX_TL(10) IAMX_TL(11) IAMX_TL(12) IAMX_TL(13) IAMX_TL(14) IAMX_TL(15)
#undef IAMX_TL
            default: tail_level<16>(cur, hor, nxt, H, W, P, P2, tp, dst, mb); break;
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------
// Behind the pyramid everything follows OpenCV's float32 scalar code (sift.simd.hpp
// adjustLocalExtrema / calcOrientationHist / calcSIFTDescriptor, mathfuncs' fastAtan2) operation
// by operation, exactly as oracle/sift_ref.c restates it: the functions below are compiled with
// contraction OFF, every float operator is one IEEE float32 operation (division and square root
// correctly rounded: hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt).  The stated
// departures from OpenCV: exp32() instead of hal::exp32f's table, correctly
// rounded cosf / sinf / powf, and histogram bins that are the exact sum of OpenCV's float32 terms
// (float64 LDS atomics, order independent) rounded to float32 once.
// -----------------------------------------------------------------------------------------------
#pragma clang fp contract(off)

__device__ __forceinline__ int cv_round(float v) { return __float2int_rn(v); }

// cv::fastAtan2, scalar form: degrees in [0, 360], 7th-order polynomial (~0.3 deg)
__device__ __forceinline__ float fast_atan2_cv(float y, float x)
{
    constexpr float P1 = 0.9997878412794807f * 57.29577951308232f, P3 = -0.3258083974640975f * 57.29577951308232f,
                    P5 = 0.1555786518463281f * 57.29577951308232f, P7 = -0.04432655554792128f * 57.29577951308232f;
    const float eps = 2.220446049250313e-16f;
    const float ax = fabsf(x), ay = fabsf(y);
    const bool steep = !(ax >= ay);
    const float num = steep ? ax : ay, den = (steep ? ay : ax) + eps;
    const float c = num / den;
    const float c2 = c * c;
    float a = (((P7 * c2 + P5) * c2 + P3) * c2 + P1) * c;
    if (steep) a = 90.f - a;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// oracle exp32(): float64 range reduction, degree-7 float32 Horner polynomial for 2^f, x <= 0
// CHECKED = false: the caller guarantees x >= -87 (the same value, without the test and its branch)
template <bool CHECKED = true>
__device__ __forceinline__ float exp32(float x)
{
    if (CHECKED && x < -87.f) return 0.f;
    const double t = (double)x * 1.4426950408889634;
    const double n = rint(t);
    const float f = (float)(t - n);
    float p = 1.5252733804059841e-05f;
    p = p * f + 0.00015403530393381608f;
    p = p * f + 0.0013333558146428443f;
    p = p * f + 0.009618129107628477f;
    p = p * f + 0.05550410866482158f;
    p = p * f + 0.2402265069591007f;
    p = p * f + 0.6931471805599453f;
    p = p * f + 1.0f;
    return ldexpf(p, (int)n);
}

// Matx33f::solve(Vec3f, DECOMP_LU) = Cramer's rule in float32 (Matx_FastSolveOp<float, 3, 3, 1>);
// determinant exactly 0 -> the zero vector
__device__ __forceinline__ void solve3_cramer(const float a[3][3], const float b[3], float x[3])
{
    const float det = a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2]) -
                      a[0][1] * (a[1][0] * a[2][2] - a[2][0] * a[1][2]) +
                      a[0][2] * (a[1][0] * a[2][1] - a[2][0] * a[1][1]);
    if (det == 0) { x[0] = x[1] = x[2] = 0.f; return; }
    const float d = 1.f / det;
    x[0] = d * (b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) -
                a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2]) +
                a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
    x[1] = d * (a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) -
                b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
    x[2] = d * (a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) -
                a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0]) +
                b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
}

// a candidate that passed the sub-pixel fit and the contrast / edge tests
struct Refined {
    int o, layer, r, c;
    float xi, xr, xc, contr;
};

// one thread per candidate: adjustLocalExtrema (<= 5 steps, contrast and edge tests), float32
__device__ __forceinline__ bool refine_one(const PyrTable &T, const Cand cd,
                                           float contrast_threshold, float edge_threshold,
                                           Refined &R)
{
    const Pyr &P = T.oct[cd.o];
    const int h = P.h, w = P.w;
    int layer = cd.layer, r = cd.r, c = cd.c;
    const float img_scale = 1.f / 255.f;
    const float deriv_scale = img_scale * 0.5f, second_scale = img_scale, cross_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int it = 0;
    // DoG layer `im` at (rr, cc): the difference of two Gaussian levels, as the pyramid defines it
    // (img / prv / nxt are the DoG layer indices layer, layer - 1, layer + 1)
#define AT(im, rr, cc) sub_rn(P.g[(im) + 1][(int64_t)(rr) * w + (cc)], P.g[(im)][(int64_t)(rr) * w + (cc)])
    for (; it < MAX_STEPS; ++it) {
        const int img = layer, prv = layer - 1, nxt = layer + 1;
        const float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale,
                             (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                             (AT(nxt, r, c) - AT(prv, r, c)) * deriv_scale};
        const float v2 = AT(img, r, c) * 2.f;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_scale;
        const float dss = (AT(nxt, r, c) + AT(prv, r, c) - v2) * second_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_scale;
        const float dxs = (AT(nxt, r, c + 1) - AT(nxt, r, c - 1) - AT(prv, r, c + 1) + AT(prv, r, c - 1)) * cross_scale;
        const float dys = (AT(nxt, r + 1, c) - AT(nxt, r - 1, c) - AT(prv, r + 1, c) + AT(prv, r - 1, c)) * cross_scale;
        const float H[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        float X[3];
        solve3_cramer(H, dD, X);
        xc = -X[0]; xr = -X[1]; xi = -X[2];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        const float big = (float)(2147483647 / 3);
        if (fabsf(xi) > big || fabsf(xr) > big || fabsf(xc) > big) return false;
        c += cv_round(xc);
        r += cv_round(xr);
        layer += cv_round(xi);
        if (layer < 1 || layer > NL || c < BORDER || c >= w - BORDER || r < BORDER || r >= h - BORDER)
            return false;
    }
    if (it >= MAX_STEPS) return false;
    float contr;
    {
        const int img = layer, prv = layer - 1, nxt = layer + 1;
        const float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale,
                             (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                             (AT(nxt, r, c) - AT(prv, r, c)) * deriv_scale};
        float t = 0.f;                                    // Matx::dot
        t += dD[0] * xc; t += dD[1] * xr; t += dD[2] * xi;
        contr = AT(img, r, c) * img_scale + t * 0.5f;
        if (fabsf(contr) * (float)NL < contrast_threshold) return false;
        const float v2 = AT(img, r, c) * 2.f;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        const float e = edge_threshold;
        if (det <= 0 || tr * tr * e >= (e + 1) * (e + 1) * det) return false;
    }
#undef AT
    R.o = cd.o; R.layer = layer; R.r = r; R.c = c;
    R.xi = xi; R.xr = xr; R.xc = xc; R.contr = contr;
    return true;
}

// one thread per candidate; the survivors of a wave reserve their output slots with one
// atomicAdd (a per-thread atomic on the single counter serialises ~10^5 device-scope atomics)
__global__ __launch_bounds__(256) void refine_kernel(PyrTable T, const Cand *__restrict__ cand,
                                                     const int *__restrict__ n_cand, int cap_c,
                                                     float contrast_threshold, float edge_threshold,
                                                     Refined *__restrict__ refined,
                                                     int *__restrict__ n_refined)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = *n_cand < cap_c ? *n_cand : cap_c;
    Refined R;
    const bool ok = idx < total && refine_one(T, cand[idx < total ? idx : 0], contrast_threshold,
                                              edge_threshold, R);
    const unsigned long long m = __ballot(ok);
    if (m == 0) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(n_refined, __popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (ok) {
        const int k = base + __popcll(m & ((1ull << lane) - 1));
        if (k < cap_c) refined[k] = R;
    }
}

constexpr int ORI_BUF = 72;       // keypoints a wave collects before it touches the global counter

// wave-wide: copy `n` buffered keypoints (8 floats each) behind one atomicAdd
__device__ __forceinline__ void flush_keypoints(const float *__restrict__ kbuf, int n,
                                                float *__restrict__ kp, int cap_k,
                                                int *__restrict__ n_kp, int lane)
{
    if (n == 0) return;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);               // the LDS writes of the list have landed
    int base = 0;
    if (lane == 0) base = atomicAdd(n_kp, n);
    base = __shfl(base, 0);
    for (int e = lane; e < n * 8; e += 64) {
        const int k = base + (e >> 3);
        if (k < cap_k) kp[(int64_t)k * 8 + (e & 7)] = kbuf[e];
    }
    __builtin_amdgcn_wave_barrier();
}

// one wave per refined candidate: the KeyPoint fields of adjustLocalExtrema, then
// calcOrientationHist over the (2*radius+1)^2 window (lanes stride over the pixels; float32
// terms W * Mag summed exactly by float64 LDS atomics, each bin rounded to float32 once),
// float32 smoothing and peak interpolation
// (96 VGPRs, 5 waves per SIMD: forcing 6 / 8 costs 64 / 132 B of scratch and measured 0 / +2 %)
__global__ __launch_bounds__(256) void orient_kernel(PyrTable T, const Refined *__restrict__ refined,
                                                     const int *__restrict__ n_refined, int cap_c,
                                                     float sigma, float *__restrict__ kp, int cap_k,
                                                     int *__restrict__ n_kp, int xcd)
{
    __shared__ double hist_s[4][ORI_BINS];
    __shared__ float sm_s[4][ORI_BINS];
    __shared__ float kbuf_s[4][ORI_BUF * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int total = *n_refined < cap_c ? *n_refined : cap_c;
    float *kbuf = kbuf_s[wave];
    int n_buf = 0;
    // persistent waves (wave-uniform control flow; no block-level barriers below).  A wave takes
    // runs of ORI_RUN consecutive candidates (the list is in the extrema scan's tile order:
    // neighbours in the list are neighbours in the image, and so are the keypoints a wave appends
    // together); the runs of one XCD's workgroups form a contiguous part of the list.
    constexpr int ORI_RUN = 4;
    const int n_runs = (total + ORI_RUN - 1) / ORI_RUN;
    const int xcds = xcd ? 8 : 1;
    const int my_xcd = xcd ? (int)(blockIdx.x & 7) : 0;
    const int runs_per = (n_runs + xcds - 1) / xcds;
    const int run_end = min((my_xcd + 1) * runs_per, n_runs);
    const int wave_here = (int)(blockIdx.x / xcds) * 4 + wave, waves_here = (int)(gridDim.x / xcds) * 4;
    for (int run = my_xcd * runs_per + wave_here; run < run_end; run += waves_here)
    for (int idx = run * ORI_RUN; idx < min(run * ORI_RUN + ORI_RUN, total); ++idx) {
    const Refined R = refined[idx];
    const Pyr &P = T.oct[R.o];
    const int h = P.h, w = P.w, o = R.o, layer = R.layer, r = R.r, c = R.c;
    const float xi = R.xi, xr = R.xr, xc = R.xc, contr = R.contr;
    const float oscale = (float)(1 << o);
    const float e3 = ((float)layer + xi) / (float)NL;
    const float size = sigma * (float)exp2((double)e3) * oscale * 2.f;       // powf(2.f, e3)
    const float px = ((float)c + xc) * oscale, py = ((float)r + xr) * oscale;
    const int octave = o + (layer << 8) + ((int)rint(((double)xi + 0.5) * 255) << 16);
    const float scl_octv = size * 0.5f / oscale;

    const float *g = P.g[layer];
    const int radius = cv_round(4.5f * scl_octv);
    const float osig = 1.5f * scl_octv;
    const float expf_scale = -1.f / (2.f * osig * osig);
    double *hist = hist_s[wave];
    float *sm = sm_s[wave];
    if (lane < ORI_BINS) hist[lane] = 0.0;
    __builtin_amdgcn_wave_barrier();
    const int side = 2 * radius + 1;
    // e / side without the integer division (~30 instructions per sample): (e + 0.5) / side is at
    // least 0.5 / side away from an integer and the float32 product is off by < 1.2e-7 x the quotient:
    // a sixteenth of that distance at side = 511 (the radius formula gives side <= 35)
    const float inv_side = 1.f / (float)side;
    const bool small_side = side < 512;
    for (int e = lane; e < side * side; e += 64) {
        const int i0 = small_side ? (int)(((float)e + 0.5f) * inv_side) : e / side;
        const int i = i0 - radius, j = e - i0 * side - radius;
        const int y = r + i, x = c + j;
        if (y <= 0 || y >= h - 1 || x <= 0 || x >= w - 1) continue;
        const float dx = g[(int64_t)y * w + x + 1] - g[(int64_t)y * w + x - 1];
        const float dy = g[(int64_t)(y - 1) * w + x] - g[(int64_t)(y + 1) * w + x];
        const float wgt = exp32((float)(i * i + j * j) * expf_scale);
        const float ori = fast_atan2_cv(dy, dx);
        const float mag = sqrtf(dx * dx + dy * dy);
        int b = cv_round((float)(ORI_BINS / 360.f) * ori);
        if (b >= ORI_BINS) b -= ORI_BINS;
        if (b < 0) b += ORI_BINS;
        atomicAdd(&hist[b], (double)(wgt * mag));
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): LDS atomics landed
    if (lane < ORI_BINS) {
        const int k = lane;
        const float m2 = (float)hist[(k + ORI_BINS - 2) % ORI_BINS], p2 = (float)hist[(k + 2) % ORI_BINS];
        const float m1 = (float)hist[(k + ORI_BINS - 1) % ORI_BINS], p1 = (float)hist[(k + 1) % ORI_BINS];
        sm[k] = (m2 + p2) * (1.f / 16.f) + (m1 + p1) * (4.f / 16.f) + (float)hist[k] * (6.f / 16.f);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float omax = lane < ORI_BINS ? sm[lane] : 0.f;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) omax = fmaxf(omax, __shfl_xor(omax, m));
    const float mag_thr = omax * 0.8f;
    bool peak = false;
    float angle = 0.f;
    if (lane < ORI_BINS) {
        const int j = lane;
        const float lft = sm[(j + ORI_BINS - 1) % ORI_BINS], rgt = sm[(j + 1) % ORI_BINS];
        if (sm[j] > lft && sm[j] > rgt && sm[j] >= mag_thr) {
            float bin = (float)j + 0.5f * (lft - rgt) / (lft - 2 * sm[j] + rgt);
            bin = bin < 0 ? (float)ORI_BINS + bin : (bin >= ORI_BINS ? bin - (float)ORI_BINS : bin);
            angle = 360.f - (360.f / ORI_BINS) * bin;
            if (fabsf(angle - 360.f) < 1.1920929e-07f) angle = 0.f;
            peak = true;
        }
    }
    // keypoints go to a per-wave LDS list first; the wave reserves output slots with one
    // atomicAdd when the list fills up or the wave is done (not one atomic per keypoint)
    const unsigned long long pm = __ballot(peak);
    const int npk = __popcll(pm);
    if (n_buf + npk > ORI_BUF) {
        flush_keypoints(kbuf, n_buf, kp, cap_k, n_kp, lane);
        n_buf = 0;
    }
    if (peak) {
        float *q = kbuf + (n_buf + __popcll(pm & ((1ull << lane) - 1))) * 8;
        // first octave is -1: report in input-image pixels (detectAndCompute; x 0.5 is exact)
        q[0] = px * 0.5f;
        q[1] = py * 0.5f;
        q[2] = size * 0.5f;
        q[3] = angle;
        q[4] = fabsf(contr);
        const int oct_out = (octave & ~255) | ((octave - 1) & 255);
        q[5] = __int_as_float(oct_out);
        q[6] = __int_as_float(o * 256 + layer);        // pyramid address for the descriptor
        q[7] = 0.f;
    }
    n_buf += npk;
    __builtin_amdgcn_wave_barrier();
    }
    flush_keypoints(kbuf, n_buf, kp, cap_k, n_kp, lane);
}

// ---------------------------------------------------------------------------------
// Order of the descriptor pass (round 6, third session).  A keypoint's window is 43-85 pixels
// square, 17-50 KB of cache lines, and neighbouring keypoints' windows overlap five-fold -- but
// orient_kernel appends keypoints in whatever order its waves finish, the workgroups of the
// descriptor kernel are dealt round-robin to the eight XCDs, and 512 waves per XCD then hold
// ~15 MB of windows from all over the pyramid against 4 MB of L2: the kernel fetched 1.97 GB per
// frame (53 KB per keypoint: no reuse at all) and, once the LDS atomics were out of the way,
// waited for those gathers (SQ_WAIT_INST_ANY 0.42 of the wave cycles).  So the pass runs over a
// permutation: keypoints are bucketed by (vertical stripe of the image = XCD, pyramid level,
// 64-pixel band of rows) with one counting sort (count + rank, one-workgroup scan, scatter); XCD x
// walks the keypoints of stripe x -- an eighth of every level, so the XCDs carry equal work --
// level by level, top to bottom, and the ~512 keypoints it has in flight are neighbours in one
// level: a region of ~2 MB.  The output rows do not move (kp[k] / desc[k] keep their index).
// ---------------------------------------------------------------------------------
constexpr int DB_ROWS = 64;                       // 64-pixel bands per level (levels taller than 4096 rows share the last)
constexpr int DB_SEGS = 32;                       // (3 - layer) * 10 + octave index (octaves past the tenth share a segment)
constexpr int DB_BUCKETS = 8 * DB_SEGS * DB_ROWS; // 16 384: the one-workgroup scan is 30 us for 65 536

__global__ __launch_bounds__(256) void desc_bucket_count_kernel(PyrTable T, const float *__restrict__ kp,
                                                                const int *__restrict__ n_kp, int cap_k,
                                                                int *__restrict__ cnt, int *__restrict__ kb,
                                                                int *__restrict__ kr, float2 *__restrict__ cs)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int total = *n_kp < cap_k ? *n_kp : cap_k;
    if (k >= total) return;
    const float *q = kp + (int64_t)k * 8;
    const int addr = __float_as_int(q[6]);
    const int o = addr >> 8, layer = addr & 255;
    const float scale = o >= 1 ? 1.f / (float)(1 << (o - 1)) : 2.f;
    const int px = max((int)(q[0] * scale), 0), py = max((int)(q[1] * scale), 0);
    const int stripe = min((int)(((int64_t)px * 8) / T.oct[o].w), 7);
    const int row = min(py >> 6, DB_ROWS - 1);
    // level order: the third layers of all octaves first, the first layers last -- a window of the
    // third layer holds up to four times the samples of one of the first, and the launch ends with
    // the workgroups that started last: those should be the short ones
    const int seg = min(max((3 - layer) * 10 + min(o, 9), 0), DB_SEGS - 1);
    const int b = (stripe * DB_SEGS + seg) * DB_ROWS + row;
    kb[k] = b;
    kr[k] = atomicAdd(&cnt[b], 1);
    // cosf / sinf of the descriptor's rotation (correctly rounded: float64 cos / sin, rounded once), one
    // LANE per keypoint here instead of one WAVE per keypoint in descriptor_kernel: ~300 float64
    // instructions that every wave of that kernel spent before its first sample
    float ori = 360.f - q[3];
    if (fabsf(ori - 360.f) < 1.1920929e-07f) ori = 0.f;
    const float ang = ori * (float)(3.141592653589793 / 180.0);
    cs[k] = make_float2((float)cos((double)ang), (float)sin((double)ang));
}

// cnt[b] -> first position of bucket b (exclusive prefix, in place); xcd_start[0..8] = the stripes' ranges
__global__ __launch_bounds__(1024) void desc_bucket_scan_kernel(int *__restrict__ cnt, int *__restrict__ xcd_start)
{
    __shared__ int part[1024];
    constexpr int PER = DB_BUCKETS / 1024;
    const int lo = threadIdx.x * PER;
    int sum = 0;
    for (int j = lo; j < lo + PER; ++j) sum += cnt[j];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int sft = 1; sft < 1024; sft <<= 1) {
        const int v = threadIdx.x >= sft ? part[threadIdx.x - sft] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    if ((lo % (DB_SEGS * DB_ROWS)) == 0) xcd_start[lo / (DB_SEGS * DB_ROWS)] = run;
    if (threadIdx.x == 1023) xcd_start[8] = part[1023];
    for (int j = lo; j < lo + PER; ++j) {
        const int c = cnt[j];
        cnt[j] = run;
        run += c;
    }
}

__global__ __launch_bounds__(256) void desc_bucket_scatter_kernel(const int *__restrict__ n_kp, int cap_k,
                                                                  const int *__restrict__ start,
                                                                  const int *__restrict__ kb,
                                                                  const int *__restrict__ kr, int *__restrict__ perm)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int total = *n_kp < cap_k ? *n_kp : cap_k;
    if (k >= total) return;
    perm[start[kb[k]] + kr[k]] = k;
}

// one wave per keypoint: calcSIFTDescriptor
// (form 0 ran 8 waves per SIMD -- 64 VGPRs and 32 B of scratch instead of 86 VGPRs / 5 waves: it waited
//  on LDS atomics and image gathers more than it issued; 1.645 / 1.626 / 1.596 ms per detection at
//  5 / 6 / 8 waves on one box, round 5.  Forms 1 and 2 issue: the counters of the third session say
//  SQ_WAIT_INST_LDS 1.06e9 -> 2.3e7 wave cycles per launch, LDS-active cycles 3.1e8 -> 1.5e8, VALU
//  instructions unchanged (2.96e8 -> 2.89e8: eight f64 adds replace the address arithmetic of
//  eight atomics), VALU issue ~0.78 of the slots -- four waves per SIMD with 96 VGPRs are enough.)
//
// FORM (round 6, third session):
//   0  every sample sends its eight terms to the histogram with eight f64 LDS atomics (rounds 2-5)
//   1  RUN COMBINING: a lane walks consecutive samples of a window row, and consecutive samples mostly
//      fall into the same (row bin, column bin, orientation bin) cell -- 0.61 of them on the 2189 x
//      1459 detect image (gradient orientation is smooth at the keypoint's scale, a bin is ~6 pixels
//      wide).  The lane keeps the eight f64 sums of its current cell in registers and sends them to
//      LDS when the cell changes: 8 atomics per RUN instead of per sample.  Sums of float32 terms in
//      float64 are exact in either grouping (the convention of the oracle: each bin = the exact sum
//      of its terms, rounded to float32 once), so the result is the same bit for bit.
//   2  form 1 + the row tables of the interval walk in registers: the lane reads the ends of its
//      first three rows once, before the loop; the loop itself touches LDS only for the histogram
//      (forms 0/1 read rowpre / rowlo for every sample, and each of those reads waits for the
//      atomics queued before it); a lane that needs a fourth row or meets an empty one takes the
//      LDS path for that step.
template <int FORM, int WAVES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void descriptor_kernel(PyrTable T, const float *__restrict__ kp,
                                                         const int *__restrict__ n_kp, int cap_k,
                                                         uint8_t *__restrict__ desc, int xcd,
                                                         const int *__restrict__ perm,
                                                         const int *__restrict__ xcd_start,
                                                         const float2 *__restrict__ cs)
{
    constexpr int d = 4, n = 8;
    constexpr int HB = (d + 2) * (d + 2) * (n + 2);       // 360
    __shared__ double hist_s[4][HB];
    __shared__ float raw_s[4][d * d * n + 2];
    __shared__ float sq_s[4][d * d * n];
    constexpr int DESC_ROWS = 160;                        // window rows handled by the interval walk
    __shared__ int rowlo_s[4][DESC_ROWS];
    __shared__ int rowpre_s[4][DESC_ROWS + 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int total = *n_kp < cap_k ? *n_kp : cap_k;
    double *hist = hist_s[wave];
    float *raw = raw_s[wave];
    // one keypoint per wave (private LDS slice, no block barriers).  A workgroup per 4 keypoints
    // rather than a few persistent waves: the window size varies 10x between keypoints and the
    // hardware's dynamic workgroup dispatch balances that (persistent waves measured 15 % slower).
    // The grid is capped at 16 384 workgroups (the keypoint count is only known on the device:
    // a grid for the whole capacity dispatched up to 375 k workgroups that exit at once); images
    // with more than 65 536 keypoints give some waves a second one.
    // The workgroups of one XCD walk a contiguous part of the list (appended by orient_kernel in
    // runs of image neighbours): the windows an XCD's L2 sees overlap instead of being spread over
    // the whole pyramid.
    // With `perm` (the counting sort above): XCD x takes the positions xcd_start[x] .. xcd_start[x + 1]
    // of the permutation, four consecutive ones (neighbours in one pyramid level) per workgroup.
    const int xcds = (xcd || perm) ? 8 : 1;
    const int my_xcd = xcds == 8 ? (int)(blockIdx.x & 7) : 0;
    const int slab = ((total + xcds - 1) / xcds + 3) & ~3;
    const int k_begin = perm ? xcd_start[my_xcd] : my_xcd * slab;
    const int k_end = perm ? min(xcd_start[my_xcd + 1], total) : min((my_xcd + 1) * slab, total);
    for (int kpos = k_begin + (int)(blockIdx.x / xcds) * 4 + wave; kpos < k_end; kpos += (int)(gridDim.x / xcds) * 4) {
    const int k = perm ? perm[kpos] : kpos;
    for (int i = lane; i < HB; i += 64) hist[i] = 0.0;
    __builtin_amdgcn_wave_barrier();
    {
        const float *q = kp + (int64_t)k * 8;
        const int addr = __float_as_int(q[6]);
        const int o = addr >> 8, layer = addr & 255;
        const Pyr &P = T.oct[o];
        const float *img = P.g[layer];
        const int h = P.h, w = P.w;
        // calcDescriptors: keypoint in the coordinates of its octave, scale = 2^-(o-1) (exact)
        const float scale = o >= 1 ? 1.f / (float)(1 << (o - 1)) : 2.f;
        const float ptx = q[0] * scale, pty = q[1] * scale;
        float ori = 360.f - q[3];
        if (fabsf(ori - 360.f) < 1.1920929e-07f) ori = 0.f;
        const float scl = q[2] * scale * 0.5f;
        const int px = cv_round(ptx), py = cv_round(pty);
        const float ang = ori * (float)(3.141592653589793 / 180.0);
        float cos_t, sin_t;                                                       // cosf / sinf, correctly rounded
        if (cs) {
            const float2 t = cs[k];                // (desc_bucket_count_kernel: the same expressions)
            cos_t = t.x; sin_t = t.y;
        } else {
            cos_t = (float)cos((double)ang); sin_t = (float)sin((double)ang);
        }
        const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f);
        const float hist_width = 3.f * scl;
        int radius = cv_round(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
        const int diag = FORM == 0 ? (int)sqrt((double)w * w + (double)h * h) : P.diag;
        radius = radius < diag ? radius : diag;
        cos_t /= hist_width;
        sin_t /= hist_width;
        // radius <= diag of the octave image (< 2^14 for any image that fits the workspace),
        // so the window index fits 32 bits: no 64-bit division in the sample loop
        const int side = 2 * radius + 1;
        // the four gradient neighbours of window position (i, j); callers guarantee that the
        // position lies inside the image (0 < r < h - 1, 0 < c < w - 1)
        struct Grad { float xl, xr, yu, yd; };
        auto fetch = [&](int i, int j) -> Grad {
            const float *pc = img + (int64_t)(py + i) * w + (px + j);
            return Grad{pc[-1], pc[1], pc[-w], pc[w]};
        };
        // (forms 1, 2) the lane's current cell and its eight partial sums
        double run[8];
        int run_base = -1;
        auto flush_run = [&]() {
            if (run_base >= 0) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int b = run_base + ((q4 >> 1) * (d + 2) + (q4 & 1)) * (n + 2);
                    atomicAdd(&hist[b], run[2 * q4]);
                    atomicAdd(&hist[b + 1], run[2 * q4 + 1]);
                }
            }
        };
        auto accumulate = [&](int i, int j, const Grad g) {
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            const float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d)) return;
            const float dx = g.xr - g.xl;
            const float dy = g.yu - g.yd;
            // (-1 < rbin, cbin < d: |r_rot|, |c_rot| < 2.5, the argument of exp32 is > -1.6)
            const float wgt = exp32<FORM == 0>((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            const float og = fast_atan2_cv(dy, dx);
            const float mag = sqrtf(dx * dx + dy * dy) * wgt;
            const float obin = (og - ori) * bins_per_rad;
            const float fr0 = floorf(rbin), fc0 = floorf(cbin), fo0 = floorf(obin);
            const int r0 = (int)fr0, c0 = (int)fc0;
            int o0 = (int)fo0;
            const float fr = rbin - fr0, fc = cbin - fc0, fo = obin - fo0;
            if (FORM == 0) {
                if (o0 < 0) o0 += n;
                if (o0 >= n) o0 -= n;
            } else {
                o0 &= n - 1;           // (og, ori in [0, 360]: -8 <= o0 <= 8, the same wrap in one instruction)
            }
            const float v_r1 = mag * fr, v_r0 = mag - v_r1;
            float vv[4], v1s[4];
            if (FORM == 0) {
                const float v_rc11 = v_r1 * fc, v_rc10 = v_r1 - v_rc11;
                const float v_rc01 = v_r0 * fc, v_rc00 = v_r0 - v_rc01;
                vv[0] = v_rc00; vv[1] = v_rc01; vv[2] = v_rc10; vv[3] = v_rc11;
            } else {
                // the same products and differences two at a time (v_pk_mul_f32 / v_pk_add_f32:
                // separately rounded, like the scalar form)
                typedef float f2 __attribute__((ext_vector_type(2)));
                const f2 vr = {v_r0, v_r1};
                const f2 c1 = vr * fc;                  // v_rc01, v_rc11
                const f2 c0v = vr - c1;                 // v_rc00, v_rc10
                const f2 c1o = c1 * fo, c0o = c0v * fo;
                const f2 c1r = c1 - c1o, c0r = c0v - c0o;
                vv[0] = c0r.x; vv[1] = c1r.x; vv[2] = c0r.y; vv[3] = c1r.y;        // vv[q] - v1
                v1s[0] = c0o.x; v1s[1] = c1o.x; v1s[2] = c0o.y; v1s[3] = c1o.y;    // v1 = vv[q] * fo
            }
            if (FORM == 0) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int rr = r0 + 1 + (q4 >> 1), cc = c0 + 1 + (q4 & 1);
                    const int base = (rr * (d + 2) + cc) * (n + 2) + o0;
                    const float v1 = vv[q4] * fo;
                    atomicAdd(&hist[base], (double)(vv[q4] - v1));
                    atomicAdd(&hist[base + 1], (double)v1);
                }
            } else {
                // (24-bit multiplies: full rate, the 32-bit v_mul_lo_u32 is a quarter of that)
                const int base = __mul24(r0 + 1, (d + 2) * (n + 2)) + __mul24(c0 + 1, n + 2) + o0;
                if (base != run_base) {
                    flush_run();
                    run_base = base;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        run[2 * q4] = (double)vv[q4];
                        run[2 * q4 + 1] = (double)v1s[q4];
                    }
                } else {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        run[2 * q4] += (double)vv[q4];
                        run[2 * q4 + 1] += (double)v1s[q4];
                    }
                }
            }
        };
        auto sample = [&](int i, int j) {
            const int r = py + i, c = px + j;
            if (r > 0 && r < h - 1 && c > 0 && c < w - 1) accumulate(i, j, fetch(i, j));
        };
        if (side <= DESC_ROWS) {
            // The samples that pass the test above fill a ROTATED square, half of the upright
            // window at best.  Per window row: a conservative column interval (one column of
            // slack on each side; the exact test still runs per sample), then the lanes walk the
            // concatenated intervals instead of the whole window.
            int *rowlo = rowlo_s[wave], *rowpre = rowpre_s[wave];
            int carry = 0;
            const double cos_d = (double)cos_t, sin_d = (double)sin_t;
            // (FORM >= 1: the bounds through one reciprocal per constraint instead of four f64
            //  divisions per row -- the interval only has to CONTAIN the samples that pass the float32
            //  test in accumulate(); floor / ceil leave a column of slack, an error of 1e-13 in a
            //  bound moves it by a column only where the bound is that close to an integer, and then
            //  both neighbours are inside the slack)
            const double inv_a[2] = {1.0 / sin_d, 1.0 / cos_d};
            for (int base = 0; base < side; base += 64) {
                const int row = base + lane;
                int jl = 0, cnt = 0;
                if (row < side) {
                    const int i = row - radius, r = py + i;
                    if (r > 0 && r < h - 1) {
                        double lo = fmax((double)-radius, (double)(1 - px));
                        double hi = fmin((double)radius, (double)(w - 2 - px));
                        // -1 < j*a + b < d  for (a, b) = (sin_t, i*cos_t + 1.5) and (cos_t, -i*sin_t + 1.5)
                        const double aa[2] = {sin_d, cos_d};
                        const double bb[2] = {i * cos_d + d / 2 - 0.5, -i * sin_d + d / 2 - 0.5};
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            if (fabs(aa[e]) < 1e-9) continue;          // no usable bound: keep all
                            double x1, x2;
                            if (FORM == 0) {
                                x1 = (-1.0 - bb[e]) / aa[e];
                                x2 = ((double)d - bb[e]) / aa[e];
                            } else {
                                x1 = (-1.0 - bb[e]) * inv_a[e];
                                x2 = ((double)d - bb[e]) * inv_a[e];
                            }
                            if (x1 > x2) { const double t = x1; x1 = x2; x2 = t; }
                            lo = fmax(lo, floor(x1));
                            hi = fmin(hi, ceil(x2));
                        }
                        if (hi >= lo) { jl = (int)lo; cnt = (int)hi - jl + 1; }
                    }
                }
                int x = cnt;
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) {
                    const int y = __shfl_up(x, sft);
                    if (lane >= sft) x += y;
                }
                if (row < side) { rowlo[row] = jl; rowpre[row] = carry + x - cnt; }
                carry += __shfl(x, 63);
            }
            if (lane == 0) rowpre[side] = carry;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            // blocked, not cyclic: lane L walks samples [L*chunk, (L+1)*chunk), so at any moment
            // the 64 lanes sit in different parts of the window and their LDS atomics mostly hit
            // different histogram cells (neighbouring samples share a cell: 8-way conflicts)
            const int chunk = (carry + 63) >> 6;
            const int s0 = lane * chunk, s1 = min(s0 + chunk, carry);
            // Software pipeline, one sample deep: the image reads of sample s + 1 are in flight
            // while sample s goes through its ~150 dependent instructions and 8 LDS atomics (the
            // lanes of a wave sit in 64 different cache lines; with the loads at the head of each
            // iteration the kernel was bound by their latency, not by any throughput).  Every
            // listed position lies inside the image (the intervals are clipped), so the loads
            // need no test.
            int row = 0;
            if (FORM < 2) {
            if (s0 < s1) {
                while (s0 >= rowpre[row + 1]) ++row;
                int ci = row - radius, cj = rowlo[row] + (s0 - rowpre[row]);
                Grad cg = fetch(ci, cj);
                for (int s = s0; s < s1; ++s) {
                    int ni = ci, nj = cj;
                    Grad ng = cg;
                    if (s + 1 < s1) {
                        while (s + 1 >= rowpre[row + 1]) ++row;
                        ni = row - radius;
                        nj = rowlo[row] + (s + 1 - rowpre[row]);
                        ng = fetch(ni, nj);
                    }
                    accumulate(ci, cj, cg);
                    ci = ni; cj = nj; cg = ng;
                }
            }
            } else if (s0 < s1) {
                // first row with rowpre[row + 1] > s0 (rows may be empty: rowpre is non-decreasing)
                int lo = 0, hi = side - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (rowpre[mid + 1] > s0) hi = mid; else lo = mid + 1;
                }
                row = lo;
                // the ends of this row and the next two, the first columns of the next two: all the
                // loop needs unless the lane's chunk spans more than three rows or meets an empty one
                const int rmax = side - 1;
                const int r1 = min(row + 1, rmax), r2 = min(row + 2, rmax);
                int end0 = rowpre[row + 1];
                const int end1 = rowpre[r1 + 1], end2 = rowpre[r2 + 1];
                const int lo1 = rowlo[r1], lo2 = rowlo[r2];
                int ahead = (row + 1 <= rmax && end1 > end0) ? ((row + 2 <= rmax && end2 > end1) ? 2 : 1) : 0;
                int ci = row - radius, cj = rowlo[row] + (s0 - rowpre[row]);
                const float *pc = img + (int64_t)(py + ci) * w + (px + cj);
                Grad cg = Grad{pc[-1], pc[1], pc[-w], pc[w]};
                int used = 0;                              // register rows consumed so far
                for (int s = s0; s < s1; ++s) {
                    int ni = ci, nj = cj;
                    const float *pn = pc;
                    Grad ng = cg;
                    if (s + 1 < s1) {
                        bool slow = false;
                        if (s + 1 < end0) {
                            nj = cj + 1;
                            pn = pc + 1;
                        } else if (used < ahead) {
                            // (s + 1 == the first sample of the next row: rows are contiguous in the list)
                            ++row;
                            ni = row - radius;
                            nj = used == 0 ? lo1 : lo2;
                            end0 = used == 0 ? end1 : end2;
                            ++used;
                            pn = img + (int64_t)(py + ni) * w + (px + nj);
                        } else {
                            slow = true;
                        }
                        if (__builtin_amdgcn_ballot_w64(slow) != 0) {
                            if (slow) {
                                while (s + 1 >= rowpre[row + 1]) ++row;
                                ni = row - radius;
                                nj = rowlo[row] + (s + 1 - rowpre[row]);
                                end0 = rowpre[row + 1];
                                pn = img + (int64_t)(py + ni) * w + (px + nj);
                            }
                        }
                        if (FORM == 9) {      // timing only: every gather from the window centre (cache resident)
                            const float *pz = img + (int64_t)py * w + px;
                            ng = Grad{pz[-1], pz[1], pz[-w], pz[w]};
                        } else
                        ng = Grad{pn[-1], pn[1], pn[-w], pn[w]};
                    }
                    accumulate(ci, cj, cg);
                    ci = ni; cj = nj; cg = ng; pc = pn;
                }
            }
        } else {
            const int nsamp = side * side;
            for (int s = lane; s < nsamp; s += 64) {
                const int i0 = s / side;
                sample(i0 - radius, s - i0 * side - radius);
            }
        }
        if (FORM >= 1) flush_run();
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): the LDS atomics have landed
    {
        // bins -> float32 (one rounding each), circular orientation bins folded in float32, then
        // the 128 values: lanes 0..63 hold 2 each
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = lane * 2 + e;             // (i*d + j)*n + kk
            const int cell = t / n, kk = t % n;
            const int i = cell / d, j = cell % d;
            const int base = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            float x = (float)hist[base + kk];
            if (kk == 0) x = x + (float)hist[base + n];
            if (kk == 1) x = x + (float)hist[base + n + 1];
            v[e] = x;
            raw[t] = x;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        // the two norms are OpenCV's scalar loops: sequential float32 sums over k = 0 .. 127
        if (FORM == 0) {
        if (lane == 0) {
            float nrm2 = 0.f;
            for (int t = 0; t < d * d * n; ++t) nrm2 += raw[t] * raw[t];
            const float thr = sqrtf(nrm2) * 0.2f;
            nrm2 = 0.f;
            for (int t = 0; t < d * d * n; ++t) {
                const float val = raw[t] < thr ? raw[t] : thr;
                nrm2 += val * val;
            }
            const float s2 = sqrtf(nrm2);
            raw[d * d * n] = thr;
            raw[d * d * n + 1] = 512.f / (s2 > 1.1920929e-07f ? s2 : 1.1920929e-07f);
        }
        } else {
            // ... of which only the ADDITIONS are sequential: every lane squares its own two values
            // (the same float32 products), lane 0 adds the 128 squares in order -- 2 x 128 dependent
            // additions instead of 2 x 128 x (compare, multiply, add) on one lane while 63 wait (a
            // sixth of the kernel's instruction slots for a keypoint of the first layer)
            float *sq = sq_s[wave];
            sq[lane * 2] = v[0] * v[0];
            sq[lane * 2 + 1] = v[1] * v[1];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane == 0) {
                float nrm2 = 0.f;
                for (int t = 0; t < d * d * n; ++t) nrm2 += sq[t];
                raw[d * d * n] = sqrtf(nrm2) * 0.2f;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const float thr0 = raw[d * d * n];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float val = v[e] < thr0 ? v[e] : thr0;
                sq[lane * 2 + e] = val * val;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane == 0) {
                float nrm2 = 0.f;
                for (int t = 0; t < d * d * n; ++t) nrm2 += sq[t];
                const float s2 = sqrtf(nrm2);
                raw[d * d * n + 1] = 512.f / (s2 > 1.1920929e-07f ? s2 : 1.1920929e-07f);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const float thr = raw[d * d * n], nrm = raw[d * d * n + 1];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float val = v[e] < thr ? v[e] : thr;
            float x = rintf(val * nrm);
            x = x < 0 ? 0 : (x > 255 ? 255 : x);
            desc[(int64_t)k * 128 + lane * 2 + e] = (uint8_t)x;
        }
    }
    __builtin_amdgcn_wave_barrier();
    }   // keypoint loop
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------
// Output order and duplicate removal (detectAndCompute -> KeyPointsFilter::removeDuplicatedSorted,
// features2d/keypoint.cpp): the detector kernels append in a nondeterministic order.
//   order 1 (OpenCV's): x, y ascending, size DESCENDING, angle ascending, response DESCENDING,
//            packed octave (before the first-octave adjustment) DESCENDING
//   order 0 (pyramid-local, rounds 1-3): octave, layer, y, x, angle ascending, then size /
//            response / octave descending
// then every keypoint equal to its predecessor in (x, y, size, angle) is dropped -- the first of
// such a run (the highest response) stays, as in OpenCV's loop.
// All fields are non-negative floats, so their bit patterns order like the values; a key is six
// 32-bit words compared lexicographically (descending fields complemented) + the row as a tie
// break.  Keys are dealt to SORT_BUCKETS buckets by a monotone function of the leading field(s)
// (the x range, or (octave, layer) x the y range), so bucket order is key order; inside its
// bucket every key is ranked by counting the smaller ones (a few dozen per bucket for an
// evenly textured image; correct, only slower, if they all fall into one), duplicates are
// flagged in the same pass, an exclusive scan of the flags closes the gaps.
// ---------------------------------------------------------------------------------
constexpr int SORT_BUCKETS = 4096;
constexpr int SORT_SEGS = 128;            // (octave index + 1) * 4 + layer < 128  (order 0)

struct SortKey {
    unsigned w[6];
    int bucket, orig;
};

struct SortParams {
    int order;
    float ext_x, ext_y;                  // image width / height in pixels (bucket scale)
};

__device__ __forceinline__ SortKey make_key(const float *row, int i, const SortParams P)
{
    SortKey k;
    const unsigned x = (unsigned)__float_as_int(row[0]), y = (unsigned)__float_as_int(row[1]);
    const unsigned size = ~(unsigned)__float_as_int(row[2]), angle = (unsigned)__float_as_int(row[3]);
    const unsigned resp = ~(unsigned)__float_as_int(row[4]);
    const int oct = __float_as_int(row[5]);
    const unsigned oct_pre = ~(unsigned)((oct & ~255) | ((oct + 1) & 255));
    if (P.order == 1) {
        int b = (int)(row[0] / P.ext_x * (float)SORT_BUCKETS);
        k.bucket = b < 0 ? 0 : (b >= SORT_BUCKETS ? SORT_BUCKETS - 1 : b);
        k.w[0] = x; k.w[1] = y; k.w[2] = size; k.w[3] = angle;
    } else {
        const int seg = ((((oct & 255) + 1) & 255) << 2 | ((oct >> 8) & 3)) & (SORT_SEGS - 1);
        constexpr int PER = SORT_BUCKETS / SORT_SEGS;
        int b = (int)(row[1] / P.ext_y * (float)PER);
        b = b < 0 ? 0 : (b >= PER ? PER - 1 : b);
        k.bucket = seg * PER + b;
        k.w[0] = y; k.w[1] = x; k.w[2] = angle; k.w[3] = size;
    }
    k.w[4] = resp; k.w[5] = oct_pre;
    k.orig = i;
    return k;
}

__global__ __launch_bounds__(256) void sort_count_kernel(const float *__restrict__ kp,
                                                         const int32_t *__restrict__ n_out, int cap,
                                                         SortParams P, int32_t *__restrict__ bucket_cnt)
{
    const int n = min(*n_out, cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&bucket_cnt[make_key(kp + (int64_t)i * 8, i, P).bucket], 1);
}

// exclusive scan of the bucket counts (one workgroup); also clears the fill counters
__global__ __launch_bounds__(1024) void sort_offsets_kernel(const int32_t *__restrict__ bucket_cnt,
                                                            int32_t *__restrict__ bucket_off,
                                                            int32_t *__restrict__ bucket_fill)
{
    constexpr int PER = SORT_BUCKETS / 1024;
    __shared__ int part[1024];
    int c[PER], sum = 0;
#pragma unroll
    for (int e = 0; e < PER; ++e) { c[e] = bucket_cnt[threadIdx.x * PER + e]; sum += c[e]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int sft = 1; sft < 1024; sft <<= 1) {
        const int v = threadIdx.x >= sft ? part[threadIdx.x - sft] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int off = part[threadIdx.x] - sum;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        bucket_off[threadIdx.x * PER + e] = off;
        bucket_fill[threadIdx.x * PER + e] = 0;
        off += c[e];
    }
    if (threadIdx.x == 1023) bucket_off[SORT_BUCKETS] = off;
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(const float *__restrict__ kp,
                                                           const int32_t *__restrict__ n_out, int cap,
                                                           SortParams P,
                                                           const int32_t *__restrict__ bucket_off,
                                                           int32_t *__restrict__ bucket_fill,
                                                           SortKey *__restrict__ keys)
{
    const int n = min(*n_out, cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const SortKey k = make_key(kp + (int64_t)i * 8, i, P);
    keys[bucket_off[k.bucket] + atomicAdd(&bucket_fill[k.bucket], 1)] = k;
}

// one thread per key: rank inside its bucket = number of smaller keys there; dup = one of them is
// equal in the four leading words (x, y, size, angle in either order)
__global__ __launch_bounds__(256) void sort_rank_kernel(const SortKey *__restrict__ keys,
                                                        const int32_t *__restrict__ bucket_off,
                                                        int32_t *__restrict__ rank,
                                                        unsigned *__restrict__ dup_bits,
                                                        int32_t *__restrict__ tile_cnt)
{
    const int n = bucket_off[SORT_BUCKETS];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const SortKey me = keys[i];
    const int lo = bucket_off[me.bucket], hi = bucket_off[me.bucket + 1];
    int cnt = 0;
    bool dup = false;
    for (int j = lo; j < hi; ++j) {
        const SortKey o = keys[j];
        bool less = false, eq = true, eq4 = true;
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            if (eq && o.w[e] != me.w[e]) { less = o.w[e] < me.w[e]; eq = false; }
            if (e == 3) eq4 = eq;
        }
        if (eq) less = o.orig < me.orig;
        cnt += less;
        dup |= less && eq4;
    }
    const int r = lo + cnt;
    rank[i] = r;
    if (dup) {                                           // (rare: a handful per frame)
        atomicOr(&dup_bits[r >> 5], 1u << (r & 31));
        atomicAdd(&tile_cnt[r >> 8], 1);
    }
}

// duplicates are flagged as bits at their sorted position (dup_bits) and counted per tile of 256
// positions (tile_cnt): exclusive scan of the tile counts (one workgroup); n_sorted = rows kept
__global__ __launch_bounds__(1024) void sort_compact_kernel(const int32_t *__restrict__ bucket_off,
                                                            int32_t *__restrict__ tile_cnt,
                                                            int32_t *__restrict__ n_sorted)
{
    __shared__ int part[1024];
    const int n = bucket_off[SORT_BUCKETS];
    const int tiles = (n + 255) >> 8;
    const int chunk = (tiles + 1023) / 1024;
    const int lo = min((int)threadIdx.x * chunk, tiles), hi = min(lo + chunk, tiles);
    int sum = 0;
    for (int j = lo; j < hi; ++j) sum += tile_cnt[j];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int sft = 1; sft < 1024; sft <<= 1) {
        const int v = threadIdx.x >= sft ? part[threadIdx.x - sft] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int j = lo; j < hi; ++j) {
        const int c = tile_cnt[j];
        tile_cnt[j] = run;                             // drops before the tile
        run += c;
    }
    if (threadIdx.x == 1023) *n_sorted = n - part[1023];
}

__global__ __launch_bounds__(256) void sort_gather_kernel(const SortKey *__restrict__ keys,
                                                          const int32_t *__restrict__ rank,
                                                          const unsigned *__restrict__ dup_bits,
                                                          const int32_t *__restrict__ tile_off,
                                                          const int32_t *__restrict__ bucket_off,
                                                          const float *__restrict__ kp,
                                                          const uint8_t *__restrict__ desc,
                                                          float *__restrict__ out_kp,
                                                          uint8_t *__restrict__ out_desc)
{
    const int n = bucket_off[SORT_BUCKETS];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 3, part = t & 7;
    if (i >= n) return;
    const int r = rank[i];
    const unsigned word = dup_bits[r >> 5];
    if ((word >> (r & 31)) & 1u) return;               // a duplicate: dropped
    int before = tile_off[r >> 8] + __popc(word & ((1u << (r & 31)) - 1u));
    for (int w = (r >> 8) << 3; w < (r >> 5); ++w) before += __popc(dup_bits[w]);
    const int src = keys[i].orig, dst = r - before;
    out_kp[(int64_t)dst * 8 + part] = kp[(int64_t)src * 8 + part];
    reinterpret_cast<uint4 *>(out_desc + (int64_t)dst * 128)[part] =
        reinterpret_cast<const uint4 *>(desc + (int64_t)src * 128)[part];
}

inline unsigned blocks(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

void gaussian_taps(double sigma, Taps &T)
{
    int ksize = (int)lrint(sigma * 8 + 1) | 1;
    int r = ksize / 2;
    if (r > (MAX_TAPS - 1) / 2) r = (MAX_TAPS - 1) / 2;
    double k[MAX_TAPS], sum = 0;
    for (int i = -r; i <= r; ++i) {
        k[i + r] = exp(-(double)(i * i) / (2.0 * sigma * sigma));
        sum += k[i + r];
    }
    T.r = r;
    for (int i = 0; i < 2 * r + 1; ++i) T.k[i] = (float)(k[i] / sum);
}

struct Layout {
    int n_oct;
    int h[MAX_OCT], w[MAX_OCT];
    int64_t g_off[MAX_OCT][6];
    int64_t up_off, cand_off, refined_off, count_off, bucket_off, total;
};

Layout make_layout(int height, int width, int cap_c)
{
    Layout L;
    int H = 2 * height, W = 2 * width;
    int n_oct = (int)lrint(log((double)(H < W ? H : W)) / log(2.0) - 2) + 1;
    if (n_oct < 1) n_oct = 1;
    if (n_oct > MAX_OCT) n_oct = MAX_OCT;
    L.n_oct = n_oct;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { int64_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    for (int o = 0; o < n_oct; ++o) {
        L.h[o] = H; L.w[o] = W;
        for (int i = 0; i < 6; ++i) L.g_off[o][i] = take((int64_t)H * W * 4);
        H /= 2; W /= 2;
        if (H < 1 || W < 1) { L.n_oct = o + 1; break; }
    }
    L.up_off = take((int64_t)L.h[0] * L.w[0] * 4);          // the 2x image before the base blur
    L.cand_off = take((int64_t)cap_c * sizeof(Cand));
    L.refined_off = take((int64_t)cap_c * sizeof(Refined));
    L.count_off = take(256);
    L.bucket_off = take((int64_t)(DB_BUCKETS + 16) * 4);     // descriptor-pass buckets + the stripes' ranges
    L.total = off;
    return L;
}

constexpr int CAP_CAND = 1 << 21;

}  // namespace

extern "C" int64_t iamx_sift_workspace_bytes(int height, int width)
{
    if (height < 1 || width < 1) return 0;
    return make_layout(height, width, CAP_CAND).total;
}

// second stream + fork / join events of iamx_sift_detect, one set per device AND calling thread
// (concurrent detections from several host threads must not share fork / join events), created on
// first use and kept for the life of the thread (nullptr if the runtime refuses: single-stream order)
namespace {
struct SideStream {
    hipStream_t stream;
    hipEvent_t fork, join;
    hipStream_t stream2;          // the one-workgroup tail of the pyramid beside the last strip octave
    hipEvent_t fork2, join2;
};

SideStream *side_stream()
{
    static thread_local SideStream slots[64];
    static thread_local bool ready[64], failed[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (failed[dev]) return nullptr;
    if (!ready[dev]) {
        SideStream &S = slots[dev];
        // highest priority: the side chain is ~40 small dependent launches that must not queue
        // behind the main stream's chip-filling levels
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        if (hipStreamCreateWithPriority(&S.stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
            hipStreamCreateWithPriority(&S.stream2, hipStreamNonBlocking, prio_hi) != hipSuccess ||
            hipEventCreateWithFlags(&S.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&S.join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&S.fork2, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&S.join2, hipEventDisableTiming) != hipSuccess) {
            failed[dev] = true;
            return nullptr;
        }
        ready[dev] = true;
    }
    return &slots[dev];
}
}  // namespace

// Everything of a detection behind the first kernel (which reads the caller's image), enqueued on
// `st` and, for the small octaves, on this thread's side stream -- ~65 launches for a 3 MP frame.
static int sift_enqueue_pyramid(const Layout &L, float contrast_threshold, float edge_threshold,
                                float sigma, char *ws, float *kp, uint8_t *desc, int cap,
                                int32_t *n_out, hipStream_t st);

// The launch sequence of a frame size never changes (grids are sized by capacity, counts live
// on the device) and every pointer in it belongs to the caller's per-detector workspace / output
// buffers, so it CAN be captured once into a HIP graph per (frame size, parameters, buffers) and
// replayed: one submission instead of ~65.  The first kernel (gray_up2x: the only reader of the
// image, whose address changes from frame to frame) stays a plain launch in front of the graph.
// OPT-IN (IAMX_SIFT_GRAPH=1), because it does not pay on MI355X / ROCm 7 (round 5,
// profiles/r5_sift_graph_ab.txt): in steady state the ~65 plain launches cost the host 0.12 ms per
// frame and the stream is kernel bound either way -- 1.773 ms per detection replayed against
// 1.781 ms launched --, and with eight detector threads in flight the replayed graphs serialise
// where plain launches of different frames interleave: 2.02 against 1.51 ms per detection.  (The
// "0.6 ms of launch gaps per frame" of round 4 was an artefact of timing four cold detections.)
namespace {
struct GraphKey {
    int height, width, cap, xcd;
    float ct, et, sigma;
    void *ws, *kp, *desc, *n_out;
    int device;
};
struct GraphSlot {
    GraphKey key;
    hipGraphExec_t exec;
    uint64_t stamp;
    bool used;
};
constexpr int GRAPH_SLOTS = 12;

inline bool graph_enabled()
{
    static const bool on = []() {
        const char *e = getenv("IAMX_SIFT_GRAPH");
        return e && e[0] == '1';
    }();
    return on;
}
}  // namespace

extern "C" int iamx_sift_detect(const uint8_t *image, int height, int width, int channels,
                                float contrast_threshold, float edge_threshold, float sigma,
                                void *workspace, int64_t workspace_bytes, float *kp, uint8_t *desc,
                                int cap, int32_t *n_out, void *stream)
{
    IAMX_REQUIRE(image && workspace && kp && desc && n_out, "null pointer");
    IAMX_REQUIRE(height >= 2 && width >= 2 && (channels == 1 || channels == 3) && cap > 0,
                 "bad image size / channels / capacity");
    const Layout L = make_layout(height, width, CAP_CAND);
    IAMX_REQUIRE(workspace_bytes >= L.total, "workspace too small (iamx_sift_workspace_bytes)");
    hipStream_t st = iamx::as_stream(stream);
    char *ws = static_cast<char *>(workspace);
    // base image, part 1: gray -> x2 (reads the caller's image: outside the graph)
    hipLaunchKernelGGL(gray_up2x_kernel, dim3(blocks((int64_t)L.h[0] * L.w[0], 256)), dim3(256), 0, st,
                       image, height, width, channels, reinterpret_cast<float *>(ws + L.up_off));
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    if (!graph_enabled() || hipStreamIsCapturing(st, &cap_status) != hipSuccess ||
        cap_status != hipStreamCaptureStatusNone)
        return sift_enqueue_pyramid(L, contrast_threshold, edge_threshold, sigma, ws, kp, desc, cap,
                                    n_out, st);
    static thread_local GraphSlot slots[GRAPH_SLOTS];
    static thread_local uint64_t clock_ = 0;
    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.height = height; key.width = width; key.cap = cap; key.xcd = xcd_enabled();
    key.ct = contrast_threshold; key.et = edge_threshold; key.sigma = sigma;
    key.ws = workspace; key.kp = kp; key.desc = desc; key.n_out = n_out;
    (void)hipGetDevice(&key.device);
    GraphSlot *hit = nullptr, *victim = &slots[0];
    for (GraphSlot &g : slots) {
        if (g.used && memcmp(&g.key, &key, sizeof(key)) == 0) { hit = &g; break; }
        if (!g.used || (victim->used && g.stamp < victim->stamp)) victim = &g;
    }
    if (!hit) {
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            return sift_enqueue_pyramid(L, contrast_threshold, edge_threshold, sigma, ws, kp, desc,
                                        cap, n_out, st);
        }
        const int rc = sift_enqueue_pyramid(L, contrast_threshold, edge_threshold, sigma, ws, kp, desc,
                                            cap, n_out, st);
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        const hipError_t e1 = hipStreamEndCapture(st, &graph);
        if (rc != IAMX_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            return rc;
        }
        if (e1 != hipSuccess || !graph ||
            hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            // the runtime refused (the capture left nothing on the stream): plain launches
            (void)hipGetLastError();
            if (graph) (void)hipGraphDestroy(graph);
            return sift_enqueue_pyramid(L, contrast_threshold, edge_threshold, sigma, ws, kp, desc,
                                        cap, n_out, st);
        }
        (void)hipGraphDestroy(graph);
        if (victim->used) (void)hipGraphExecDestroy(victim->exec);
        victim->key = key;
        victim->exec = exec;
        victim->used = true;
        hit = victim;
    }
    hit->stamp = ++clock_;
    if (hipGraphLaunch(hit->exec, st) != hipSuccess)
        return iamx::fail(IAMX_ELAUNCH, "iamx_sift_detect: hipGraphLaunch: %s",
                          hipGetErrorString(hipGetLastError()));
    return iamx::check_launch("iamx_sift_detect");
}

static int sift_enqueue_pyramid(const Layout &L, float contrast_threshold, float edge_threshold,
                                float sigma, char *ws, float *kp, uint8_t *desc, int cap,
                                int32_t *n_out, hipStream_t st)
{
    PyrTable T;
    T.n_oct = L.n_oct;
    for (int o = 0; o < L.n_oct; ++o) {
        T.oct[o].h = L.h[o];
        T.oct[o].w = L.w[o];
        T.oct[o].diag = (int)sqrt((double)L.w[o] * L.w[o] + (double)L.h[o] * L.h[o]);
        for (int i = 0; i < 6; ++i) T.oct[o].g[i] = reinterpret_cast<float *>(ws + L.g_off[o][i]);
    }
    Cand *cand = reinterpret_cast<Cand *>(ws + L.cand_off);
    Refined *refined = reinterpret_cast<Refined *>(ws + L.refined_off);
    int *n_cand = reinterpret_cast<int *>(ws + L.count_off);
    int *n_refined = n_cand + 1;
    (void)hipMemsetAsync(n_cand, 0, 8, st);
    (void)hipMemsetAsync(n_out, 0, 4, st);
    // descriptor-pass order: bucket counters, and kb / kr / perm / (cos, sin) in the candidate list's memory (dead
    // once refine_kernel has run); a capacity that does not fit there keeps the list order
    int *bucket_cnt = reinterpret_cast<int *>(ws + L.bucket_off);
    int *xcd_start = bucket_cnt + DB_BUCKETS;
    const bool sorted_pass = desc_sorted() && (int64_t)cap * 20 <= (int64_t)CAP_CAND * (int64_t)sizeof(Cand);
    if (sorted_pass) (void)hipMemsetAsync(bucket_cnt, 0, (size_t)(DB_BUCKETS + 16) * 4, st);

    // OpenCV's sigma is a double (1.6); the ABI carries a float, whose widening (1.60000002...)
    // would move every Gaussian tap in its last bits: take the parameter to 6 decimals
    const double sigma_d = round((double)sigma * 1e6) / 1e6;
    // layer sigmas (Lowe / OpenCV buildGaussianPyramid)
    double sig[6];
    sig[0] = sigma_d;
    const double kf = pow(2.0, 1.0 / NL);
    for (int i = 1; i < NL + 3; ++i) {
        const double sp = pow(kf, (double)(i - 1)) * sigma_d, stt = sp * kf;
        sig[i] = sqrt(stt * stt - sp * sp);
    }
    // dst = G_sigma * src
    Taps taps[NL + 3];
    for (int i = 1; i < NL + 3; ++i) gaussian_taps(sig[i], taps[i]);
    auto blur = [&](hipStream_t q, const float *src, float *dst, int h, int w, const Taps &tp) {
        blur_level(q, src, h, w, tp, dst);
    };
    // base image, part 2: blur(sqrt(sigma^2 - 1)) of the doubled image
    {
        const int H = L.h[0], W = L.w[0];
        float *up = reinterpret_cast<float *>(ws + L.up_off);
        Taps tb;
        gaussian_taps(sqrt(fmax(sigma_d * sigma_d - 1.0, 0.01)), tb);
        blur(st, up, T.oct[0].g[0], H, W, tb);
    }
    const float threshold = floorf(0.5f * contrast_threshold / NL * 255.f);
    auto extrema = [&](hipStream_t q, int o) {
        const int H = L.h[o], W = L.w[o];
        if (H > 2 * BORDER && W > 2 * BORDER) {
            GaussStack G;
            for (int i = 0; i < NL + 3; ++i) G.g[i] = T.oct[o].g[i];
            hipLaunchKernelGGL(extrema_kernel,
                               dim3((unsigned)((W - 2 * BORDER + 61) / 62),
                                    (unsigned)((H - 2 * BORDER + 4 * EXT_ROWS - 1) / (4 * EXT_ROWS))),
                               dim3(256), 0, q, G, H, W, o, threshold, cand, CAP_CAND, n_cand, xcd_enabled());
        }
    };
    auto downsample = [&](hipStream_t q, int o) {
        hipLaunchKernelGGL(downsample_kernel, dim3(blocks((int64_t)L.h[o] * L.w[o], 256)), dim3(256), 0, q,
                           T.oct[o - 1].g[NL], L.w[o - 1], L.h[o], L.w[o], T.oct[o].g[0]);
    };
    // The levels of an octave depend on each other and octave o + 1 starts from level NL of octave
    // o, so the pyramid is one dependency chain whose links get small quickly: from octave
    // SIDE_OCTAVE on (<= 1/64 of the pixels of octave 0) a level is a few workgroups and its cost
    // is the launch.  Issue order: levels 1..NL of the big octaves (the chain that leads to the
    // small ones) first, then the small octaves -- all their levels and extrema scans -- on a second
    // stream beside the remaining two levels and the extrema scans of the big octaves.
    constexpr int SIDE_OCTAVE = 4;
    const int o_side = L.n_oct < SIDE_OCTAVE ? L.n_oct : SIDE_OCTAVE;
    for (int o = 0; o < o_side; ++o) {
        if (o > 0) downsample(st, o);
        for (int i = 1; i <= NL; ++i)
            blur(st, T.oct[o].g[i - 1], T.oct[o].g[i], L.h[o], L.w[o], taps[i]);
    }
    SideStream *side = nullptr;
    hipStream_t ts = st;
    if (o_side < L.n_oct) {
        side = side_stream();
        if (side) {
            ts = side->stream;
            (void)hipEventRecord(side->fork, st);
            (void)hipStreamWaitEvent(ts, side->fork, 0);
        }
    }
    // the host needs ~8 us per launch: the few big launches go in first so that the GPU works on
    // them while the ~50 small ones of the side chain are still being enqueued (the other order
    // left the main stream idle for 0.45 ms per frame)
    for (int o = 0; o < o_side; ++o) {
        for (int i = NL + 1; i < NL + 3; ++i)
            blur(st, T.oct[o].g[i - 1], T.oct[o].g[i], L.h[o], L.w[o], taps[i]);
        extrema(st, o);
    }
    // small octaves: strip kernels down to o_tail, then ONE workgroup for the rest
    int o_tail = L.n_oct;
    for (int o = L.n_oct - 1; o >= (o_side > 1 ? o_side : 1) && (int64_t)L.h[o] * (L.w[o] | 1) <= TAIL_PIXELS; --o)
        o_tail = o;
    // The one-workgroup tail (0.25 ms) only needs level NL of the last strip octave: it starts on a
    // THIRD stream as soon as that level exists, beside the octave's two remaining levels and its
    // extrema scan, and the scans of its own octaves are one launch (round 5: the chain tail ->
    // four small scans used to END 0.1 ms after the main stream had run out of work,
    // profiles/r5_sift_single_timeline.txt).
    auto launch_tail = [&](hipStream_t q) {
        TapSet TS;
        for (int i = 1; i < NL + 3; ++i) TS.t[i - 1] = taps[i];
        hipLaunchKernelGGL(pyramid_tail_kernel, dim3(1), dim3(1024), 0, q, T, o_tail, TS);
        int mx = 1, my = 1;
        for (int o = o_tail; o < L.n_oct; ++o) {
            if (!(L.h[o] > 2 * BORDER && L.w[o] > 2 * BORDER)) continue;
            mx = std::max(mx, (L.w[o] - 2 * BORDER + 61) / 62);
            my = std::max(my, (L.h[o] - 2 * BORDER + 4 * EXT_ROWS - 1) / (4 * EXT_ROWS));
        }
        hipLaunchKernelGGL(extrema_multi_kernel, dim3((unsigned)mx, (unsigned)my, (unsigned)(L.n_oct - o_tail)),
                           dim3(256), 0, q, T, o_tail, threshold, cand, CAP_CAND, n_cand);
    };
    bool tail_forked = false;
    for (int o = o_side; o < o_tail; ++o) {
        downsample(ts, o);
        for (int i = 1; i < NL + 3; ++i) {
            blur(ts, T.oct[o].g[i - 1], T.oct[o].g[i], L.h[o], L.w[o], taps[i]);
            if (side && o == o_tail - 1 && i == NL && o_tail < L.n_oct) {
                (void)hipEventRecord(side->fork2, ts);
                (void)hipStreamWaitEvent(side->stream2, side->fork2, 0);
                launch_tail(side->stream2);
                (void)hipEventRecord(side->join2, side->stream2);
                tail_forked = true;
            }
        }
        extrema(ts, o);
    }
    if (o_tail < L.n_oct && !tail_forked) launch_tail(ts);
    if (tail_forked) (void)hipStreamWaitEvent(st, side->join2, 0);
    if (side) (void)hipEventRecord(side->join, ts);
    if (side) (void)hipStreamWaitEvent(st, side->join, 0);
    // the number of candidates is only known on the device: launch for the capacity in slabs
    // sized by the largest plausible count (threads beyond *n_cand exit immediately)
    hipLaunchKernelGGL(refine_kernel, dim3(blocks(CAP_CAND, 256)), dim3(256), 0, st, T, cand, n_cand,
                       CAP_CAND, contrast_threshold, edge_threshold, refined, n_refined);
    hipLaunchKernelGGL(orient_kernel, dim3(512 * 8), dim3(256), 0, st, T, refined, n_refined,
                       CAP_CAND, (float)sigma_d, kp, cap, n_out, xcd_enabled());
    {
        const unsigned g = (blocks(cap, 4) + 7u) & ~7u;     // (a multiple of 8: the XCD slabs)
        const dim3 dg(g < 16384u ? g : 16384u);     // (XCD slabs of the unsorted list: -21 % traffic but +6 % time, profiles/r4_sift_ab.txt)
        const int *perm = nullptr;
        float2 *csp = nullptr;
        if (sorted_pass) {
            int *kb = reinterpret_cast<int *>(cand), *kr = kb + cap, *pm = kr + cap;
            csp = reinterpret_cast<float2 *>(pm + cap);
            hipLaunchKernelGGL(desc_bucket_count_kernel, dim3(blocks(cap, 256)), dim3(256), 0, st, T, kp, n_out, cap,
                               bucket_cnt, kb, kr, csp);
            hipLaunchKernelGGL(desc_bucket_scan_kernel, dim3(1), dim3(1024), 0, st, bucket_cnt, xcd_start);
            hipLaunchKernelGGL(desc_bucket_scatter_kernel, dim3(blocks(cap, 256)), dim3(256), 0, st, n_out, cap,
                               bucket_cnt, kb, kr, pm);
            perm = pm;
        }
        switch (desc_form()) {
        case 0: hipLaunchKernelGGL((descriptor_kernel<0, 8>), dg, dim3(256), 0, st, T, kp, n_out, cap, desc, 0, perm, xcd_start, csp); break;
        case 1: hipLaunchKernelGGL((descriptor_kernel<1, 6>), dg, dim3(256), 0, st, T, kp, n_out, cap, desc, 0, perm, xcd_start, csp); break;
        case 9: hipLaunchKernelGGL((descriptor_kernel<9, 4>), dg, dim3(256), 0, st, T, kp, n_out, cap, desc, 0, perm, xcd_start, csp); break;
        case 25: hipLaunchKernelGGL((descriptor_kernel<2, 5>), dg, dim3(256), 0, st, T, kp, n_out, cap, desc, 0, perm, xcd_start, csp); break;
        default: hipLaunchKernelGGL((descriptor_kernel<2, 4>), dg, dim3(256), 0, st, T, kp, n_out, cap, desc, 0, perm, xcd_start, csp); break;
        }
    }
    return iamx::check_launch("iamx_sift_detect");
}

extern "C" int64_t iamx_sift_sort_workspace_bytes(int cap)
{
    if (cap < 1) return 0;
    return (int64_t)cap * (int64_t)(sizeof(SortKey) + 4) + ((int64_t)cap / 32 + cap / 256 + 4) * 4 +
           (3 * SORT_BUCKETS + 4) * 4 + 256;
}

// detectAndCompute's removeDuplicatedSorted on the lists iamx_sift_detect appended (n_out DEV [1]
// as written by it; rows beyond cap were not stored and are ignored): duplicates dropped, rows in
// OpenCV's output order (order 1) or the pyramid-local order (order 0).  width / height: the
// detect image's size in pixels (bucket scale only).  out_kp DEV [cap][8], out_desc DEV
// [cap][128], n_sorted DEV [1] = rows kept; workspace DEV iamx_sift_sort_workspace_bytes(cap).
extern "C" int iamx_sift_sort(const float *kp, const uint8_t *desc, const int32_t *n_out, int cap,
                              int order, int width, int height, void *workspace,
                              int64_t workspace_bytes, float *out_kp, uint8_t *out_desc,
                              int32_t *n_sorted, void *stream)
{
    IAMX_REQUIRE(kp && desc && n_out && workspace && out_kp && out_desc && n_sorted, "null pointer");
    IAMX_REQUIRE(cap > 0 && workspace_bytes >= iamx_sift_sort_workspace_bytes(cap), "workspace too small");
    IAMX_REQUIRE((order == 0 || order == 1) && width > 0 && height > 0, "bad order / image size");
    hipStream_t st = iamx::as_stream(stream);
    char *ws = static_cast<char *>(workspace);
    SortKey *keys = reinterpret_cast<SortKey *>(ws);
    int32_t *rank = reinterpret_cast<int32_t *>(ws + (int64_t)cap * sizeof(SortKey));
    // [dup_bits | tile_cnt | bucket_cnt] are zeroed together
    unsigned *dup_bits = reinterpret_cast<unsigned *>(rank + cap);
    int32_t *tile_cnt = reinterpret_cast<int32_t *>(dup_bits + cap / 32 + 1);
    int32_t *bucket_cnt = tile_cnt + cap / 256 + 1;
    int32_t *bucket_off = bucket_cnt + SORT_BUCKETS;
    int32_t *bucket_fill = bucket_off + SORT_BUCKETS + 1;
    const SortParams P{order, (float)width, (float)height};
    (void)hipMemsetAsync(dup_bits, 0, ((size_t)cap / 32 + 1 + cap / 256 + 1 + SORT_BUCKETS) * 4, st);
    const unsigned g = blocks(cap, 256);
    hipLaunchKernelGGL(sort_count_kernel, dim3(g), dim3(256), 0, st, kp, n_out, cap, P, bucket_cnt);
    hipLaunchKernelGGL(sort_offsets_kernel, dim3(1), dim3(1024), 0, st, bucket_cnt, bucket_off, bucket_fill);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(g), dim3(256), 0, st, kp, n_out, cap, P, bucket_off,
                       bucket_fill, keys);
    hipLaunchKernelGGL(sort_rank_kernel, dim3(g), dim3(256), 0, st, keys, bucket_off, rank, dup_bits, tile_cnt);
    hipLaunchKernelGGL(sort_compact_kernel, dim3(1), dim3(1024), 0, st, bucket_off, tile_cnt, n_sorted);
    hipLaunchKernelGGL(sort_gather_kernel, dim3(blocks((int64_t)cap * 8, 256)), dim3(256), 0, st, keys,
                       rank, dup_bits, tile_cnt, bucket_off, kp, desc, out_kp, out_desc);
    return iamx::check_launch("iamx_sift_sort");
}

// Where a pyramid level lives inside the workspace of iamx_sift_detect (tests / diagnosis):
// kind 0 = Gaussian level `index` (0..5) of `octave`.  (kind 1, the DoG levels of rounds 1-3, is
// gone with the buffers: DoG level i is level i + 1 minus level i, taken where it is needed.)
extern "C" int iamx_sift_pyramid_level(int height, int width, int octave, int kind, int index,
                                       int64_t *byte_offset, int *level_h, int *level_w,
                                       int *n_octaves)
{
    IAMX_REQUIRE(byte_offset && level_h && level_w && n_octaves, "null pointer");
    IAMX_REQUIRE(height >= 2 && width >= 2, "bad image size");
    const Layout L = make_layout(height, width, CAP_CAND);
    *n_octaves = L.n_oct;
    IAMX_REQUIRE(kind == 0, "only the Gaussian levels (kind 0) are stored");
    IAMX_REQUIRE(octave >= 0 && octave < L.n_oct && index >= 0 && index < 6, "no such level");
    *byte_offset = L.g_off[octave][index];
    *level_h = L.h[octave];
    *level_w = L.w[octave];
    return IAMX_OK;
}
