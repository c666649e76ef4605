// Per-pair match filters on the device (gfx950) -- what scripts/lib/matcher.py does in python
// after the metric threshold, for BOTH directions of an image pair, one workgroup per pair:
//
//   matcher.py:258-269   stable sort of the survivors by metric, clip to the best 2000
//   matcher.py:271-283   < min_pairs -> []
//   matcher.py:285       cv2.xfeatures2d.matchGMS(size, size, kp1, kp2, matches,
//                        withRotation=True, withScale=False, thresholdFactor=5.0)
//                        (algorithm: scripts/lib/archive/gms_matcher.py:74-285 -- 20x20 grids,
//                        four half-cell shifted left grids, 8 rotations of the 3x3 neighbourhood,
//                        threshold factor * sqrt(mean cell population))
//   matcher.py:157-182   filter_duplicates: first come wins on "%.2f-%.2f" % kp.pt of either end
//   matcher.py:296-299   < min_pairs -> []
//   matcher.py:304-318   reverse direction only if the forward one kept >= min_pairs
//   matcher.py:187-200   filter_cross_check: p stays in fwd iff [p1, p0] is in rev
//
// Integer/index work and IEEE f64 arithmetic in the python order => results identical to the
// host implementation (imageanalysis_amd/gms.py, matcher.py), which is pinned to the
// reference's golden vectors.  All state of a pair lives in LDS (<= 64 KiB); motion statistics
// are a 4096-slot open-addressing hash table (key = left cell * 400 + right cell), built with
// LDS atomics (integer adds/max: order independent => deterministic).
#include "iamx_common.h"

namespace {

constexpr int GRID = 20, NCELL = GRID * GRID;
constexpr int CLIP = 2000;             // MYMAX of matcher.py:267
constexpr int SORT_CAP = 2048;         // elements of the in-LDS bitonic sort (>= CLIP)
constexpr int MAX_SURV = 1 << 24;      // survivors of one direction the selection accepts
constexpr int TAB = 4096;              // hash slots (<= 2000 distinct keys: load <= 0.49)
constexpr int NT = 256;

struct PostArgs {
    const int64_t *surv_off;           // [2*n_pairs + 1] first survivor of every ordered pair
    const int32_t *surv_cnt;           // [2*n_pairs]
    const int32_t *surv_q, *surv_t;    // survivor query / train rows
    const double *surv_metric;
    const int32_t *pairs;              // [2*n_pairs][2] image slots; ordered pair k = fwd of pair
                                       // k, ordered pair n_pairs + k = its reverse
    const int64_t *kp_off;             // [n_images] first keypoint of every image slot
    const float *xy;                   // [total kp][2] kp.pt
    const int32_t *key2;               // [total kp][2] round-half-even(100 * kp.pt) ("%.2f")
    int n_pairs;
    double width, height, min_pairs, thr_factor;
    int32_t *out_cnt;                  // [n_pairs] cross-checked matches (same for both directions)
    int32_t *out_pairs;                // [n_pairs][CLIP][2] forward list [query row, train row]
    int32_t *scratch;                  // [n_pairs][2][CLIP][2] work lists of both directions
    int32_t *out_stat;                 // [n_pairs][4]: fwd after GMS, fwd after de-dup, rev after
                                       // GMS, rev after de-dup (-1: stage not reached)
    int32_t *status;                   // [n_pairs] 0 ok, 1 = more than MAX_SURV survivors (host path)
};

__constant__ int ROT[8][9] = {{0, 1, 2, 3, 4, 5, 6, 7, 8}, {3, 0, 1, 6, 4, 2, 7, 8, 5},
                              {6, 3, 0, 7, 4, 1, 8, 5, 2}, {7, 6, 3, 8, 4, 0, 5, 2, 1},
                              {8, 7, 6, 5, 4, 3, 2, 1, 0}, {5, 8, 7, 2, 4, 6, 1, 0, 3},
                              {2, 5, 8, 1, 4, 7, 0, 3, 6}, {1, 2, 5, 0, 4, 8, 3, 6, 7}};

struct Lds {
    union {
        struct {                       // sort phase
            unsigned long long key[SORT_CAP];
            int idx[SORT_CAP];
            int hist[256];
        } s;
        struct {                       // GMS / de-dup / cross-check phases
            int tab_key[TAB];
            int tab_val[TAB];
            unsigned short lg[4][CLIP];
            unsigned short rg[CLIP];
            unsigned char bits[CLIP];
            int cnt[NCELL], jbest[NCELL], cell[NCELL];
        } g;
    };
    int red[NT / 64];
    int bcast[4];
};

__device__ __forceinline__ unsigned hash32(unsigned k)
{
    k ^= k >> 16; k *= 0x7feb352du; k ^= k >> 15; k *= 0x846ca68bu; k ^= k >> 16;
    return k;
}

__device__ __forceinline__ int block_sum(int v, int *red)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// neighbour k (0..8, row major dy*3+dx) of cell c on a GRID x GRID grid, -1 outside
__device__ __forceinline__ int nb9(int c, int k)
{
    const int x = c % GRID + (k % 3 - 1), y = c / GRID + (k / 3 - 1);
    return (x >= 0 && x < GRID && y >= 0 && y < GRID) ? x + y * GRID : -1;
}

// stable sort by metric + clip: list[0..m) = (q, t) of the m best survivors; returns m.
// More than SORT_CAP survivors: an 8-pass radix select on the metric bits finds the key of the
// CLIP-th smallest, an order-preserving pass collects everything below it plus the first ties
// (position order = what a stable sort would keep), and only those <= CLIP entries are sorted.
__device__ int sort_clip(Lds &L, const PostArgs &A, int op, int32_t *list)
{
    const int64_t b = A.surv_off[op];
    const int n = A.surv_cnt[op];
    const double *metric = A.surv_metric + b;
    // metric >= 0 (NaN is never kept): the f64 bit pattern orders like the value
    auto key_of = [&](int i) { return (unsigned long long)__double_as_longlong(metric[i]); };
    int m;                               // elements placed in L.s.key / L.s.idx
    if (n <= SORT_CAP) {
        m = n;
        for (int i = threadIdx.x; i < n; i += NT) { L.s.key[i] = key_of(i); L.s.idx[i] = i; }
    } else {
        unsigned long long prefix = 0;   // the bits of the CLIP-th smallest key decided so far
        int want = CLIP;                 // rank (1-based) of the wanted key among the prefix group
        for (int byte = 7; byte >= 0; --byte) {
            L.s.hist[threadIdx.x] = 0;
            __syncthreads();
            const int sh = 8 * byte;
            for (int i = threadIdx.x; i < n; i += NT) {
                const unsigned long long k = key_of(i);
                if (byte == 7 || (k >> (sh + 8)) == (prefix >> (sh + 8)))
                    atomicAdd(&L.s.hist[(int)((k >> sh) & 255)], 1);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                int acc = 0, bin = 0;
                for (; bin < 256; ++bin) {
                    if (acc + L.s.hist[bin] >= want) break;
                    acc += L.s.hist[bin];
                }
                L.bcast[0] = bin;
                L.bcast[1] = want - acc;
            }
            __syncthreads();
            prefix |= (unsigned long long)L.bcast[0] << sh;
            want = L.bcast[1];
            __syncthreads();
        }
        // prefix = key of the CLIP-th smallest; `want` ties of it are kept, in position order
        const unsigned long long T = prefix;
        int base = 0, ties = 0;
        for (int s0 = 0; s0 < n; s0 += NT) {
            const int i = s0 + threadIdx.x;
            const unsigned long long k = i < n ? key_of(i) : ~0ull;
            const bool less = i < n && k < T, tie = i < n && k == T;
            const unsigned long long bl = __ballot(less), bt = __ballot(tie);
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            __syncthreads();
            if (lane == 0) { L.red[wave] = __popcll(bl); L.bcast[wave] = __popcll(bt); }
            __syncthreads();
            int tie_before = ties, less_before = 0, tie_all = 0, less_all = 0;
            for (int w = 0; w < NT / 64; ++w) {
                if (w < wave) { tie_before += L.bcast[w]; less_before += L.red[w]; }
                tie_all += L.bcast[w]; less_all += L.red[w];
            }
            const unsigned long long below = (1ull << lane) - 1;
            const int my_tie = tie_before + __popcll(bt & below);          // ties in front of me
            const bool take = less || (tie && my_tie < want);
            // output position = kept entries in front: all `less` + the ties below the quota
            const int ties_kept_before = min(my_tie, want);
            const int pos = base + less_before + __popcll(bl & below) + (ties_kept_before - min(ties, want));
            if (take) { L.s.key[pos] = k; L.s.idx[pos] = i; }
            base += less_all + (min(ties + tie_all, want) - min(ties, want));
            ties += tie_all;
            __syncthreads();
        }
        m = base;                         // == CLIP
    }
    int cap = 1;
    while (cap < m) cap <<= 1;
    for (int i = m + threadIdx.x; i < cap; i += NT) { L.s.key[i] = ~0ull; L.s.idx[i] = 0x7FFFFFFF; }
    __syncthreads();
    for (int k = 2; k <= cap; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < cap; i += NT) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long ki = L.s.key[i], kl = L.s.key[l];
                    const int ii = L.s.idx[i], il = L.s.idx[l];
                    const bool gt = ki > kl || (ki == kl && ii > il);     // (metric, position)
                    const bool up = (i & k) == 0;
                    if (gt == up) {
                        L.s.key[i] = kl; L.s.key[l] = ki;
                        L.s.idx[i] = il; L.s.idx[l] = ii;
                    }
                }
            }
            __syncthreads();
        }
    }
    const int mm = m < CLIP ? m : CLIP;
    for (int i = threadIdx.x; i < mm; i += NT) {
        const int sidx = L.s.idx[i];
        list[2 * i] = A.surv_q[b + sidx];
        list[2 * i + 1] = A.surv_t[b + sidx];
    }
    __syncthreads();
    return mm;
}

// GMS inlier bits: after the call bit r of L.g.bits[i] is the mask of rotation r; returns the
// rotation whose mask has the most inliers (first one on ties, gms.py `c > best_n`), -1 if none
__device__ int gms(Lds &L, const PostArgs &A, const int32_t *list, int m, int img_q, int img_t)
{
    const float *xq = A.xy + 2 * A.kp_off[img_q], *xt = A.xy + 2 * A.kp_off[img_t];
    for (int i = threadIdx.x; i < m; i += NT) {
        const int q = list[2 * i], t = list[2 * i + 1];
        const double lx = (double)xq[2 * q] / A.width, ly = (double)xq[2 * q + 1] / A.height;
        const double rx = (double)xt[2 * t] / A.width, ry = (double)xt[2 * t + 1] / A.height;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const double ox = (g & 1) ? 0.5 : 0.0, oy = (g & 2) ? 0.5 : 0.0;
            const int x = (int)floor(__dadd_rn(__dmul_rn(lx, (double)GRID), ox));
            const int y = (int)floor(__dadd_rn(__dmul_rn(ly, (double)GRID), oy));
            L.g.lg[g][i] = (x >= GRID || y >= GRID) ? 0xFFFF : (unsigned short)(x + y * GRID);
        }
        const int gx = (int)floor(__dmul_rn(rx, (double)GRID)), gy = (int)floor(__dmul_rn(ry, (double)GRID));
        L.g.rg[i] = (unsigned short)(gx + gy * GRID);
        L.g.bits[i] = 0;
    }
    __syncthreads();
    for (int g = 0; g < 4; ++g) {
        for (int i = threadIdx.x; i < TAB; i += NT) { L.g.tab_key[i] = -1; L.g.tab_val[i] = 0; }
        for (int i = threadIdx.x; i < NCELL; i += NT) { L.g.cnt[i] = 0; L.g.jbest[i] = 0; }
        __syncthreads();
        // motion statistics: count of matches per (left cell, right cell)
        for (int i = threadIdx.x; i < m; i += NT) {
            const int lc = L.g.lg[g][i];
            if (lc == 0xFFFF) continue;
            const int key = lc * NCELL + L.g.rg[i];
            unsigned h = hash32((unsigned)key) & (TAB - 1);
            while (true) {
                const int prev = atomicCAS(&L.g.tab_key[h], -1, key);
                if (prev == -1 || prev == key) break;
                h = (h + 1) & (TAB - 1);
            }
            atomicAdd(&L.g.tab_val[h], 1);
            atomicAdd(&L.g.cnt[lc], 1);
        }
        __syncthreads();
        // np.argmax(stats, axis=1): largest count, smallest right cell on ties
        for (int i = threadIdx.x; i < TAB; i += NT) {
            const int key = L.g.tab_key[i];
            if (key >= 0) atomicMax(&L.g.jbest[key / NCELL], (L.g.tab_val[i] << 16) | (0xFFFF - key % NCELL));
        }
        __syncthreads();
        for (int i = threadIdx.x; i < NCELL; i += NT)
            L.g.jbest[i] = L.g.cnt[i] ? 0xFFFF - (L.g.jbest[i] & 0xFFFF) : 0;
        __syncthreads();
        for (int r = 0; r < 8; ++r) {
            for (int c = threadIdx.x; c < NCELL; c += NT) {
                int cellv = -1;
                if (L.g.cnt[c] != 0) {
                    const int jb = L.g.jbest[c];
                    int score = 0, tot = 0, npair = 0;
#pragma unroll
                    for (int k = 0; k < 9; ++k) {
                        const int ll = nb9(c, k), rr = nb9(jb, ROT[r][k]);
                        if (ll < 0 || rr < 0) continue;
                        ++npair;
                        tot += L.g.cnt[ll];
                        const int key = ll * NCELL + rr;
                        unsigned h = hash32((unsigned)key) & (TAB - 1);
                        while (true) {
                            const int kk = L.g.tab_key[h];
                            if (kk == key) { score += L.g.tab_val[h]; break; }
                            if (kk == -1) break;
                            h = (h + 1) & (TAB - 1);
                        }
                    }
                    const double thresh = A.thr_factor * sqrt((double)tot / (double)(npair > 1 ? npair : 1));
                    cellv = (double)score < thresh ? -2 : jb;
                }
                L.g.cell[c] = cellv;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < m; i += NT) {
                const int lc = L.g.lg[g][i];
                if (lc != 0xFFFF && L.g.cell[lc] == (int)L.g.rg[i]) L.g.bits[i] |= (unsigned char)(1 << r);
            }
            __syncthreads();
        }
    }
    int best_r = -1, best_n = 0;
    for (int r = 0; r < 8; ++r) {
        int c = 0;
        for (int i = threadIdx.x; i < m; i += NT) c += (L.g.bits[i] >> r) & 1;
        c = block_sum(c, L.red);
        if (c > best_n) { best_n = c; best_r = r; }
    }
    return best_r;
}

// order-preserving compaction of the entries whose `keep(i)` is set; returns the new length
template <typename F>
__device__ int compact_list(Lds &L, int32_t *list, int m, F keep)
{
    int base = 0;
    for (int s = 0; s < m; s += NT) {
        const int i = s + threadIdx.x;
        const bool k = i < m && keep(i);
        int q = 0, t = 0;
        if (k) { q = list[2 * i]; t = list[2 * i + 1]; }
        const unsigned long long bal = __ballot(k);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) L.red[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += L.red[w];
        const int total = L.red[0] + L.red[1] + L.red[2] + L.red[3];
        if (k) {
            const int pos = off + __popcll(bal & ((1ull << lane) - 1));
            list[2 * pos] = q;          // pos <= i, and every entry < s + NT was already read
            list[2 * pos + 1] = t;
        }
        base += total;
        __syncthreads();
    }
    return base;
}

// filter_duplicates: sequential first-come-wins on the "%.2f-%.2f" keys of either end point
// (a dropped pair does not reserve its keys).  A parallel pass gives every pair the hash slot of
// its query key and of its train key (equal keys <=> equal slots) and notices whether any key
// repeats at all; only then ONE thread replays the python loop -- on the slot numbers, entirely
// in LDS (two flag reads and at most two flag writes per pair; SIFT reports a pixel once per
// dominant orientation, so on real frames nearly every pair has repeats: the replay used to walk
// the lists and the key tables in global memory, ~5 us per pair of dependent loads, 10 ms per
// round of 256 pairs; profiles/r5_kernel_stats.txt against r6).
__device__ int dedupe(Lds &L, const PostArgs &A, int32_t *list, int m, int img_q, int img_t)
{
    const int32_t *kq = A.key2 + 2 * A.kp_off[img_q], *kt = A.key2 + 2 * A.kp_off[img_t];
    int *set1 = L.g.tab_key, *set2 = L.g.tab_val;          // slots hold list position + 1
    unsigned short *slot_q = L.g.lg[0], *slot_t = L.g.lg[1];    // (the GMS grids are dead by now)
    for (int i = threadIdx.x; i < TAB; i += NT) { set1[i] = 0; set2[i] = 0; }
    if (threadIdx.x == 0) L.bcast[0] = 0;
    __syncthreads();
    auto same = [](const int32_t *k, int a, int b) { return k[2 * a] == k[2 * b] && k[2 * a + 1] == k[2 * b + 1]; };
    for (int i = threadIdx.x; i < m; i += NT) {
        for (int side = 0; side < 2; ++side) {
            const int32_t *k = side ? kt : kq;
            int *set = side ? set2 : set1;
            const int row = list[2 * i + side];
            unsigned h = hash32((unsigned)k[2 * row] * 0x9E3779B1u ^ (unsigned)k[2 * row + 1]) & (TAB - 1);
            while (true) {
                const int prev = atomicCAS(&set[h], 0, i + 1);
                if (prev == 0) break;
                if (same(k, list[2 * (prev - 1) + side], row)) { L.bcast[0] = 1; break; }
                h = (h + 1) & (TAB - 1);
            }
            (side ? slot_t : slot_q)[i] = (unsigned short)h;
        }
    }
    __syncthreads();
    if (L.bcast[0] == 0) return m;
    __syncthreads();
    for (int i = threadIdx.x; i < TAB; i += NT) { set1[i] = 0; set2[i] = 0; }
    for (int i = threadIdx.x; i < m; i += NT) L.g.bits[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 0; i < m; ++i) {
            const int hq = slot_q[i], ht = slot_t[i];
            if (!(set1[hq] | set2[ht])) {
                set1[hq] = 1;
                set2[ht] = 1;
                L.g.bits[i] = 1;
            }
        }
    }
    __syncthreads();
    return compact_list(L, list, m, [&](int i) { return L.g.bits[i] != 0; });
}

// one direction up to the de-duplicated list; returns its length (0 = "[]")
__device__ int one_direction(Lds &L, const PostArgs &A, int op, int img_q, int img_t, int32_t *list,
                             int32_t *stat)
{
    int m = sort_clip(L, A, op, list);
    if ((double)m < A.min_pairs) return 0;
    const int r = gms(L, A, list, m, img_q, img_t);
    m = r < 0 ? 0 : compact_list(L, list, m, [&](int i) { return (L.g.bits[i] >> r) & 1; });
    if (threadIdx.x == 0) stat[0] = m;
    m = dedupe(L, A, list, m, img_q, img_t);
    if (threadIdx.x == 0) stat[1] = m;
    if ((double)m < A.min_pairs) return 0;
    return m;
}

__global__ __launch_bounds__(NT) void postfilter_kernel(PostArgs A)
{
    __shared__ Lds L;
    const int p = blockIdx.x, n = A.n_pairs;
    int32_t *fwd = A.scratch + (int64_t)p * 2 * CLIP * 2, *rev = fwd + CLIP * 2;
    int32_t *stat = A.out_stat + 4 * p;
    if (threadIdx.x < 4) stat[threadIdx.x] = -1;
    if (A.surv_cnt[p] > MAX_SURV || A.surv_cnt[n + p] > MAX_SURV) {
        if (threadIdx.x == 0) { A.status[p] = 1; A.out_cnt[p] = 0; }
        return;
    }
    if (threadIdx.x == 0) A.status[p] = 0;
    const int img1 = A.pairs[2 * p], img2 = A.pairs[2 * p + 1];
    const int nf = one_direction(L, A, p, img1, img2, fwd, stat);
    int kept = 0;
    if ((double)nf >= A.min_pairs) {
        const int nr = one_direction(L, A, n + p, img2, img1, rev, stat + 2);
        if (nr > 0 && nf > 0) {
            // cross check: hash set of the reverse pairs, keyed (query row of rev, train row of rev)
            int *tk = L.g.tab_key, *tv = L.g.tab_val;
            for (int i = threadIdx.x; i < TAB; i += NT) tk[i] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < nr; i += NT) {
                const int a = rev[2 * i], b = rev[2 * i + 1];
                unsigned h = hash32((unsigned)a * 0x9E3779B1u ^ (unsigned)b) & (TAB - 1);
                while (atomicCAS(&tk[h], 0, i + 1) != 0) h = (h + 1) & (TAB - 1);
            }
            __syncthreads();
            (void)tv;
            kept = compact_list(L, fwd, nf, [&](int i) {
                const int a = fwd[2 * i + 1], b = fwd[2 * i];          // [p1, p0]
                unsigned h = hash32((unsigned)a * 0x9E3779B1u ^ (unsigned)b) & (TAB - 1);
                while (tk[h]) {
                    const int j = tk[h] - 1;
                    if (rev[2 * j] == a && rev[2 * j + 1] == b) return true;
                    h = (h + 1) & (TAB - 1);
                }
                return false;
            });
        }
    }
    int32_t *out = A.out_pairs + (int64_t)p * CLIP * 2;
    for (int i = threadIdx.x; i < 2 * kept; i += NT) out[i] = fwd[i];
    if (threadIdx.x == 0) A.out_cnt[p] = kept;
}

}  // namespace

namespace {
// ---- results of a batch, packed -----------------------------------------------------------------
// The per-pair result slots are [n_pairs][clip] (8 + 8 bytes per slot) and on an all-pairs
// schedule 95-99 % of the pairs end with nothing: downloading the slots moved 130 MB per 4096
// pairs across PCIe, behind the kernels on the same stream -- as long as the sweep itself.  The
// pairs that have matches are packed back to back instead (exclusive scan of the counts by one
// workgroup, then one workgroup per pair copies its rows); `off`, `out_pairs` and `out_z` may be
// page-locked HOST memory (the device writes it directly: the size of the transfer is only known
// here), or device memory.
__global__ __launch_bounds__(1024) void pack_scan_kernel(const int32_t *__restrict__ cnt,
                                                         const int32_t *__restrict__ status, int n,
                                                         int64_t *__restrict__ off)
{
    __shared__ int64_t wsum[16];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int k = base + threadIdx.x;
        const int64_t v = (k < n && status[k] == 0) ? (int64_t)cnt[k] : 0;
        int64_t x = v;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int64_t y = __shfl_up(x, sft);
            if ((threadIdx.x & 63) >= sft) x += y;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
        __syncthreads();
        int64_t before = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wsum[w];
        if (k < n) off[k] = before + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[n] = carry;
}

__global__ __launch_bounds__(256) void pack_copy_kernel(const int32_t *__restrict__ cnt,
                                                        const int32_t *__restrict__ status,
                                                        const int32_t *__restrict__ pairs,
                                                        const double *__restrict__ z, int clip,
                                                        int64_t cap, const int64_t *__restrict__ off,
                                                        int32_t *__restrict__ out_pairs,
                                                        double *__restrict__ out_z)
{
    const int k = blockIdx.x;
    if (status[k] != 0) return;
    const int c = cnt[k];
    const int64_t o = off[k];
    if (c <= 0 || o + c > cap) return;
    const int2 *src = reinterpret_cast<const int2 *>(pairs) + (int64_t)k * clip;
    int2 *dst = reinterpret_cast<int2 *>(out_pairs) + o;
    for (int i = threadIdx.x; i < c; i += 256) dst[i] = src[i];
    if (z && out_z) {
        const double *zs = z + (int64_t)k * clip;
        for (int i = threadIdx.x; i < c; i += 256) out_z[o + i] = zs[i];
    }
}

}  // namespace

extern "C" int iamx_match_postfilter_clip(void) { return CLIP; }

extern "C" int iamx_match_pack_results(const int32_t *cnt, const int32_t *status, const int32_t *pairs,
                                       const double *z, int n_pairs, int clip, int64_t cap,
                                       int64_t *off, int32_t *out_pairs, double *out_z, void *stream)
{
    IAMX_REQUIRE(cnt && status && pairs && off && out_pairs, "null pointer");
    IAMX_REQUIRE(n_pairs >= 0 && clip > 0 && cap >= 0, "bad size");
    hipStream_t st = iamx::as_stream(stream);
    hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, status, n_pairs, off);
    if (n_pairs)
        hipLaunchKernelGGL(pack_copy_kernel, dim3(n_pairs), dim3(256), 0, st, cnt, status, pairs, z,
                           clip, cap, (const int64_t *)off, out_pairs, out_z);
    return iamx::check_launch("iamx_match_pack_results");
}

extern "C" int iamx_match_postfilter(const int64_t *surv_off, const int32_t *surv_cnt,
                                     const int32_t *surv_q, const int32_t *surv_t,
                                     const double *surv_metric, const int32_t *pairs,
                                     const int64_t *kp_off, const float *xy, const int32_t *key2,
                                     int n_pairs, double width, double height, double min_pairs,
                                     double threshold_factor, int32_t *out_cnt, int32_t *out_pairs,
                                     int32_t *scratch, int32_t *out_stat, int32_t *status,
                                     void *stream)
{
    IAMX_REQUIRE(surv_off && surv_cnt && surv_q && surv_t && surv_metric && pairs && kp_off && xy &&
                     key2 && out_cnt && out_pairs && scratch && out_stat && status,
                 "null pointer");
    IAMX_REQUIRE(n_pairs >= 0 && width > 0 && height > 0, "bad size");
    if (n_pairs == 0) return IAMX_OK;
    PostArgs A{surv_off, surv_cnt, surv_q, surv_t, surv_metric, pairs, kp_off, xy, key2, n_pairs,
               width, height, min_pairs, threshold_factor, out_cnt, out_pairs, scratch, out_stat,
               status};
    hipLaunchKernelGGL(postfilter_kernel, dim3((unsigned)n_pairs), dim3(NT), 0,
                       iamx::as_stream(stream), A);
    return iamx::check_launch("iamx_match_postfilter");
}
