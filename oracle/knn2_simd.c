/* ORACLE / TEST INFRASTRUCTURE ONLY -- a tuned CPU form of the brute-force 2-NN of
 * oracle/cpu_ref.c (knn2_rows: scripts/lib/matcher.py:203-216 restated as exact brute force,
 * squared L2 in int32, ties -> lowest train index) for bench.py's cpu_baseline leg: the scalar
 * triple loop runs at ~3.6 GMAC/s per core, which says little about what host cores can do.
 *
 * AVX-512 VNNI (vpdpbusd: 64 u8 x s8 multiply-accumulates per instruction), chosen at run time
 * (__builtin_cpu_supports); without it oracle_knn2_simd_available() is 0 and the caller keeps the
 * scalar form.  Same results as knn2_rows bit for bit (tests/test_oracle.py).
 *
 * Layout: the train image is repacked per call into tiles of 16 rows, dimension-major in groups
 * of 4 bytes -- tile[k][row][4] = (b[row][4k .. 4k+3] ^ 0x80) as int8 -- so that one vpdpbusd of
 * the query's broadcast 4 bytes against a tile register adds those 4 products into the lane of
 * each of the 16 train rows: distances come out lane per train row, no horizontal sums, and the
 * running (best, second) of a query are 16-lane vectors merged once at the end.
 *   d2(a, b) = |a|^2 + |b|^2 - 2 (a . (b - 128) + 128 sum(a))
 * Four query rows share every tile load. */
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define D 128

int oracle_knn2_simd_available(void)
{
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
           __builtin_cpu_supports("avx512vnni");
}

/* train image -> tiles; nb[row] = |b|^2 (INT32_MAX/2 for the padding rows) */
static void pack_train(const uint8_t *t, int nt, int8_t *tiles, int32_t *nb)
{
    const int n_tiles = (nt + 15) / 16;
    for (int tl = 0; tl < n_tiles; ++tl)
        for (int k = 0; k < D / 4; ++k)
            for (int r = 0; r < 16; ++r) {
                const int row = tl * 16 + r;
                int8_t *dst = tiles + (((size_t)tl * (D / 4) + k) * 16 + r) * 4;
                for (int e = 0; e < 4; ++e)
                    dst[e] = row < nt ? (int8_t)(t[(size_t)row * D + 4 * k + e] ^ 0x80) : 0;
            }
    for (int row = 0; row < n_tiles * 16; ++row) {
        int32_t s = 0;
        if (row < nt)
            for (int k = 0; k < D; ++k) s += (int32_t)t[(size_t)row * D + k] * t[(size_t)row * D + k];
        nb[row] = row < nt ? s : (INT32_MAX / 2);
    }
}

__attribute__((target("avx512f,avx512bw,avx512vnni")))
static void knn2_rows_vnni(const uint8_t *q, int q0, int q1, const int8_t *tiles, const int32_t *nb,
                           int nt, int32_t *idx, int32_t *d2)
{
    const int n_tiles = (nt + 15) / 16;
    const __m512i lane = _mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    for (int i0 = q0; i0 < q1; i0 += 4) {
        const int nqb = q1 - i0 < 4 ? q1 - i0 : 4;
        const uint8_t *a[4];
        int32_t na[4], sa[4];
        for (int u = 0; u < 4; ++u) {
            a[u] = q + (size_t)(i0 + (u < nqb ? u : 0)) * D;
            int32_t s = 0, s2 = 0;
            for (int k = 0; k < D; ++k) { s += a[u][k]; s2 += (int32_t)a[u][k] * a[u][k]; }
            sa[u] = s; na[u] = s2;
        }
        __m512i best[4], bidx[4], sec[4], sidx[4];
        for (int u = 0; u < 4; ++u) {
            best[u] = sec[u] = _mm512_set1_epi32(INT32_MAX);
            bidx[u] = sidx[u] = _mm512_set1_epi32(-1);
        }
        for (int tl = 0; tl < n_tiles; ++tl) {
            const int8_t *tp = tiles + (size_t)tl * (D / 4) * 64;
            __m512i acc0 = _mm512_setzero_si512(), acc1 = acc0, acc2 = acc0, acc3 = acc0;
            for (int k = 0; k < D / 4; ++k) {
                const __m512i b = _mm512_loadu_si512((const void *)(tp + (size_t)k * 64));
                acc0 = _mm512_dpbusd_epi32(acc0, _mm512_set1_epi32(*(const int32_t *)(a[0] + 4 * k)), b);
                acc1 = _mm512_dpbusd_epi32(acc1, _mm512_set1_epi32(*(const int32_t *)(a[1] + 4 * k)), b);
                acc2 = _mm512_dpbusd_epi32(acc2, _mm512_set1_epi32(*(const int32_t *)(a[2] + 4 * k)), b);
                acc3 = _mm512_dpbusd_epi32(acc3, _mm512_set1_epi32(*(const int32_t *)(a[3] + 4 * k)), b);
            }
            const __m512i vnb = _mm512_loadu_si512((const void *)(nb + tl * 16));
            const __m512i vidx = _mm512_add_epi32(lane, _mm512_set1_epi32(tl * 16));
            __m512i acc[4] = {acc0, acc1, acc2, acc3};
            for (int u = 0; u < 4; ++u) {
                /* d2 = na + nb - 2 (dp + 128 sa) */
                __m512i dp = _mm512_add_epi32(acc[u], _mm512_set1_epi32(128 * sa[u]));
                __m512i d = _mm512_sub_epi32(_mm512_add_epi32(vnb, _mm512_set1_epi32(na[u])),
                                             _mm512_slli_epi32(dp, 1));
                /* padding rows carry a huge nb: never selected.  Strict < keeps the lower train
                 * index of equal distances inside a lane (indices grow with the tile number) */
                const __mmask16 m1 = _mm512_cmplt_epi32_mask(d, best[u]);
                const __mmask16 m2 = _mm512_cmplt_epi32_mask(d, sec[u]);
                sec[u] = _mm512_mask_mov_epi32(sec[u], m2, d);
                sidx[u] = _mm512_mask_mov_epi32(sidx[u], m2, vidx);
                sec[u] = _mm512_mask_mov_epi32(sec[u], m1, best[u]);
                sidx[u] = _mm512_mask_mov_epi32(sidx[u], m1, bidx[u]);
                best[u] = _mm512_mask_mov_epi32(best[u], m1, d);
                bidx[u] = _mm512_mask_mov_epi32(bidx[u], m1, vidx);
            }
        }
        for (int u = 0; u < nqb; ++u) {
            int32_t cd[32], ci[32];
            _mm512_storeu_si512((void *)cd, best[u]);
            _mm512_storeu_si512((void *)(cd + 16), sec[u]);
            _mm512_storeu_si512((void *)ci, bidx[u]);
            _mm512_storeu_si512((void *)(ci + 16), sidx[u]);
            int32_t b0 = INT32_MAX, b1 = INT32_MAX, j0 = -1, j1 = -1;
            for (int e = 0; e < 32; ++e) {
                const int32_t s = cd[e], j = ci[e];
                if (j < 0) continue;
                if (s < b0 || (s == b0 && j < j0)) { b1 = b0; j1 = j0; b0 = s; j0 = j; }
                else if (s < b1 || (s == b1 && j < j1)) { b1 = s; j1 = j; }
            }
            idx[2 * (i0 + u)] = j0; idx[2 * (i0 + u) + 1] = j1;
            d2[2 * (i0 + u)] = b0;  d2[2 * (i0 + u) + 1] = b1;
        }
    }
}

/* same contract as oracle_knn2_l2_u8_batch (cpu_ref.c); returns -2 without AVX-512 VNNI */
int oracle_knn2_l2_u8_batch_simd(const uint8_t *images, int n_rows, const int32_t *pairs, int n_pairs,
                                 int32_t *idx, int32_t *d2, int nthreads)
{
    if (n_rows < 2) return -1;
    if (!oracle_knn2_simd_available()) return -2;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    /* the packed form of every image that occurs as a train image */
    int n_img = 0;
    for (int p = 0; p < n_pairs; ++p) {
        if (pairs[2 * p] + 1 > n_img) n_img = pairs[2 * p] + 1;
        if (pairs[2 * p + 1] + 1 > n_img) n_img = pairs[2 * p + 1] + 1;
    }
    const int n_tiles = (n_rows + 15) / 16;
    const size_t tile_bytes = (size_t)n_tiles * 16 * D;
    int8_t *tiles = (int8_t *)aligned_alloc(64, tile_bytes * (size_t)n_img);
    int32_t *nb = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_tiles * 16 * (size_t)n_img);
    char *used = (char *)calloc((size_t)n_img, 1);
    if (!tiles || !nb || !used) return -3;
    for (int p = 0; p < n_pairs; ++p) used[pairs[2 * p + 1]] = 1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i = 0; i < n_img; ++i)
        if (used[i])
            pack_train(images + (size_t)i * n_rows * D, n_rows, tiles + tile_bytes * (size_t)i,
                       nb + (size_t)n_tiles * 16 * (size_t)i);
    const int blocks = (n_rows + 63) / 64;
    const long items = (long)n_pairs * blocks;
#pragma omp parallel for schedule(dynamic, 1)
    for (long it = 0; it < items; ++it) {
        const int p = (int)(it / blocks), b = (int)(it % blocks);
        const int ti = pairs[2 * p + 1];
        const uint8_t *q = images + (size_t)pairs[2 * p] * n_rows * D;
        const int r0 = b * 64, r1 = r0 + 64 < n_rows ? r0 + 64 : n_rows;
        knn2_rows_vnni(q, r0, r1, tiles + tile_bytes * (size_t)ti, nb + (size_t)n_tiles * 16 * (size_t)ti,
                       n_rows, idx + (size_t)p * n_rows * 2, d2 + (size_t)p * n_rows * 2);
    }
    free(tiles); free(nb); free(used);
    return 0;
}
