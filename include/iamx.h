/* iamx.h -- C ABI of libiamx.so: the MI355X (gfx950) feature-matching and sparse
 * bundle-adjustment hot path of NorthStarUAS/ImageAnalysis.
 *
 * Conventions
 *   - every entry point returns 0 on success, a negative IAMX_E* code otherwise;
 *     iamx_last_error() gives the message of the calling thread's last failure.
 *   - every pointer marked DEV is a HIP device pointer (e.g. torch.Tensor.data_ptr());
 *     HOST pointers are ordinary host memory.  The caller allocates every buffer.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only
 *     enqueue work, the caller synchronises.  No hidden globals, no allocation.
 *   - reference citations are file:line in NorthStarUAS/ImageAnalysis.
 *
 * A reference-side binding (ctypes) is shown in INTEGRATION.md.
 */
#ifndef IAMX_H
#define IAMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IAMX_OK            0
#define IAMX_EINVAL       -1   /* bad argument (null pointer, size out of range)        */
#define IAMX_ELAUNCH      -2   /* HIP reported a launch / runtime error                 */
#define IAMX_ENODEVICE    -3   /* no gfx950 device visible                              */
#define IAMX_EUNSUPPORTED -4   /* valid input of a kind this path does not handle       */
#define IAMX_ENOMEM       -5   /* a host routine could not allocate its work arrays     */

#define IAMX_DESC_DIM      128 /* SIFT descriptor length (scripts/lib/image.py:324)      */
#define IAMX_ROW_PAD       128 /* packed images are padded to a multiple of this many rows */

int         iamx_version(void);
const char *iamx_last_error(void);
/* name of the architecture the kernels were compiled for ("gfx950") */
const char *iamx_arch(void);

/* ------------------------------------------------------------------------------------
 * Descriptor store.  cv2.SIFT hands the reference float32 rows holding integers 0..255
 * (scripts/lib/image.py:324, cache format :204-217).  The device store keeps them as
 * int8 (value-128), each image padded to IAMX_ROW_PAD rows of zeros, plus two int32 per
 * row: norm_q = sum((a-128)^2) and norm_t = norm_q + 2*sum(a-128), so that
 *     sum((a-b)^2) = norm_q[a] + norm_t[b] + 2 * sum((127-a)*(b-128))      (exact, int32)
 * and the inner sum is one i8 MFMA contraction.
 * ------------------------------------------------------------------------------------ */
/* rows of storage needed for an image with n_rows descriptors */
int64_t iamx_desc_padded_rows(int64_t n_rows);

/* src: DEV [n_rows][128] uint8 (or float32 integer-valued for _f32; values are clamped
 * to 0..255 after rounding to nearest).  dst: DEV [padded_rows][128] int8;
 * norm_q/norm_t: DEV [padded_rows] int32. */
int iamx_desc_pack_u8(const uint8_t *src, int64_t n_rows, int8_t *dst,
                      int32_t *norm_q, int32_t *norm_t, void *stream);
int iamx_desc_pack_f32(const float *src, int64_t n_rows, int8_t *dst,
                       int32_t *norm_q, int32_t *norm_t, void *stream);
/* the inverse for n_img images of a packed store: image i's rows (first stored row img_off[i]) as
 * uint8 at rows [dst_off[i], dst_off[i+1]) of dst -- the source a different layout of the same
 * images is packed from once their descriptors have left the host (scripts/lib/matcher.py:1012-1026
 * drops them from memory on a timer).  img_off DEV [n_img] int32, dst_off DEV [n_img+1] int64. */
int iamx_desc_unpack_u8(const int8_t *desc, const int32_t *img_off, const int64_t *dst_off, int n_img,
                        int max_rows_per_image, uint8_t *dst, void *stream);

/* ------------------------------------------------------------------------------------
 * K2: exact 2-nearest-neighbour search in L2 -- replaces
 *   the_matcher.knnMatch(des1, des2, k=2)           scripts/lib/matcher.py:212-214
 * (FLANN there; the exact search FLANN approximates == cv2.BFMatcher(NORM_L2)).
 * For every query row: the two train rows with the smallest squared distance, nearest
 * first, equal distances ordered by train index (lowest first).  cv2's DMatch.distance
 * is float32(sqrt(float32(d2))).
 *
 * Batched form: one launch matches many ordered (query image, train image) pairs out of
 * one packed descriptor store.
 *   desc/norm_q/norm_t  DEV  packed store (all images back to back)
 *   img_off  DEV [n_img] int32  first packed row of each image (multiple of IAMX_ROW_PAD)
 *   img_n    DEV [n_img] int32  valid rows of each image (>= 2 for a train image)
 *   pairs    DEV [n_pairs][2] int32  (query image, train image)
 *   wg_off   DEV [n_pairs+1] int32   exclusive scan of iamx_knn2_wg_per_pair(n_q)
 *   out_off  DEV [n_pairs] int64     first output row of each pair
 *   out_idx  DEV [sum n_q][2] int32  train row (0-based inside the train image)
 *   out_d2   DEV [sum n_q][2] int32  squared distances
 *   total_wg = wg_off[n_pairs] (HOST value)
 * ------------------------------------------------------------------------------------ */
int iamx_knn2_wg_per_pair(int n_query_rows);

int iamx_knn2_l2_pairs(const int8_t *desc, const int32_t *norm_q, const int32_t *norm_t,
                       const int32_t *img_off, const int32_t *img_n,
                       const int32_t *pairs, const int32_t *wg_off, const int64_t *out_off,
                       int n_pairs, int total_wg,
                       int32_t *out_idx, int32_t *out_d2, void *stream);

/* Single pair over two packed images (thin wrapper over the batched kernel).
 * q_*, t_*: DEV packed image (iamx_desc_pack_*); idx/d2: DEV [nq][2] int32. */
int iamx_knn2_l2_u8(const int8_t *q_desc, const int32_t *q_norm_q, int nq,
                    const int8_t *t_desc, const int32_t *t_norm_t, int nt,
                    int32_t *idx, int32_t *d2, void *stream);

/* ------------------------------------------------------------------------------------
 * Quality metric + threshold -- replaces the python loop scripts/lib/matcher.py:253-263:
 *     ratio = d0/d1 ; metric = d0*ratio ; keep metric < max_distance*match_ratio
 * with d0,d1 = float32 sqrt of the squared distances, the rest in float64 like python.
 *   idx,d2     DEV [n][2] int32  (output of iamx_knn2_*)
 *   seg_off    DEV [n_seg+1] int64  rows of each (ordered) pair inside idx/d2
 *   metric     DEV [n] float64     (all rows; NaN where d1 == 0 -> python raises there)
 *   keep       DEV [n] uint8
 *   seg_count  DEV [n_seg] int32   survivors per pair
 *   zero_div   DEV [1] int32       incremented for rows with d1 == 0 (matcher.py:255)
 * ------------------------------------------------------------------------------------ */
int iamx_match_metric(const int32_t *d2, const int64_t *seg_off, int n_seg, double thresh,
                      double *metric, uint8_t *keep, int32_t *seg_count, int32_t *zero_div,
                      void *stream);

/* Order-preserving compaction of the survivors of every pair.
 *   surv_off   DEV [n_seg+1] int64  exclusive scan of seg_count (caller computes it, or
 *                                   iamx_exclusive_scan_i32)
 *   surv_q/surv_t  DEV [sum seg_count] int32   query row / train row
 *   surv_metric    DEV [sum seg_count] float64 */
int iamx_match_compact(const int32_t *idx, int idx_stride, const double *metric,
                       const uint8_t *keep, const int64_t *seg_off, const int64_t *surv_off,
                       int n_seg, int32_t *surv_q, int32_t *surv_t, double *surv_metric,
                       void *stream);
/* (surv_t[k] = idx[idx_stride * row]: stride 2 reads the best index of iamx_knn2_l2_*, stride 1
 *  the tile ids of iamx_knn2v2_pairs, which iamx_knn2v2_resolve then turns into train rows) */

/* ------------------------------------------------------------------------------------
 * K2, fast form: top-2 DISTANCES without index tracking in the sweep; the train index is
 * recovered afterwards, only for the rows that survive the metric threshold.  Same survivor
 * lists as iamx_knn2_l2_pairs + iamx_match_metric + iamx_match_compact (tests pin that).
 *   exact_second = 1: out_d2 = exact (best, second) for every row (2 VALU ops / distance);
 *                     finish with iamx_match_metric/_compact + iamx_knn2v2_resolve.
 *   exact_second = 0: out_d2[.][1] is an UPPER BOUND of the second distance (smallest distance
 *                     outside the best's 16-row group; 0.75 VALU ops / distance).  Threshold
 *                     with it (iamx_match_metric/_compact keep a superset of the survivors),
 *                     then iamx_knn2v2_finish makes `second` exact for those rows (d2 updated
 *                     in place), re-applies the test, resolves the train rows and compacts
 *                     each pair's survivors in place: pair p owns
 *                     surv_*[surv_off[p] .. surv_off[p] + surv_cnt[p]).
 *
 * Train-side store ("desc2"): rows of an image stably partitioned by the parity of
 * NB = |a-128|^2 + 2*sum(a-128), each class zero-padded to 128 rows:
 *   rows_cap = iamx_desc2_rows_cap(n)   rows to reserve per image
 *   dst DEV [rows_cap][128] int8, norm2/cinit/perm DEV [rows_cap] int32 (perm = original row,
 *   -1 on padding), meta DEV [4] int32 = {n, even chunks, odd chunks, n_even}
 * Query side = the ordinary store of iamx_desc_pack_*.
 *   out_d2   DEV [sum n_q][2]  squared distances (best, second)
 *   out_tile DEV [sum n_q]     32-row tile of the packed train image holding the best row
 * iamx_knn2v2_resolve: surv_t holds tile ids on entry and train rows (original numbering,
 * lowest row among equal distances) on exit; n_unresolved counts internal inconsistencies (0).
 * ------------------------------------------------------------------------------------ */
int64_t iamx_desc2_rows_cap(int64_t n_rows);
/* scratch: DEV [3 * n_rows] int32 (3 * total_rows for the batch form).  Batch form: images
 * back to back in `src`, src_off DEV [n_img+1] int64 first source row, dst_off DEV [n_img]
 * int32 first packed row (multiples of 128, >= rows_cap apart), meta DEV [n_img][4]. */
int iamx_desc2_pack_u8(const uint8_t *src, int64_t n_rows, int8_t *dst, int32_t *norm2,
                       int32_t *cinit, int32_t *perm, int32_t *meta, int32_t *scratch,
                       void *stream);
int iamx_desc2_pack_f32(const float *src, int64_t n_rows, int8_t *dst, int32_t *norm2,
                        int32_t *cinit, int32_t *perm, int32_t *meta, int32_t *scratch,
                        void *stream);
int iamx_desc2_pack_batch_u8(const uint8_t *src, const int64_t *src_off, const int32_t *dst_off,
                             int n_img, int64_t total_rows, int max_rows_per_image, int8_t *dst,
                             int32_t *norm2, int32_t *cinit, int32_t *perm, int32_t *meta,
                             int32_t *scratch, void *stream);
int iamx_knn2v2_pairs(const int8_t *desc_q, const int32_t *norm_q, const int32_t *qimg_off,
                      const int32_t *qimg_n, const int8_t *desc_t, const int32_t *cinit,
                      const int32_t *timg_off, const int32_t *tmeta, const int32_t *pairs,
                      const int32_t *wg_off, const int64_t *out_off, int n_pairs, int total_wg,
                      int rows_per_wg /* 256, 512 or (exact_second = 0 only) 1024:
                                         wg_off = scan of ceil(n_q / rows_per_wg) */,
                      int exact_second, int32_t *out_d2, int32_t *out_tile, void *stream);
int iamx_knn2v2_resolve(const int8_t *desc_q, const int32_t *norm_q, const int32_t *qimg_off,
                        const int8_t *desc_t, const int32_t *norm2_t, const int32_t *perm,
                        const int32_t *timg_off, const int32_t *pairs, const int64_t *out_off,
                        const int32_t *d2, const int64_t *surv_off, const int32_t *surv_q,
                        int32_t *surv_t, int n_pairs, int32_t *n_unresolved, void *stream);
/* thresh: the same value as given to iamx_match_metric; zero_div is incremented for every row
 * whose exact second distance is 0 (scripts/lib/matcher.py:255 divides by it). */
int iamx_knn2v2_finish(const int8_t *desc_q, const int32_t *norm_q, const int32_t *qimg_off,
                       const int8_t *desc_t, const int32_t *norm2_t, const int32_t *perm,
                       const int32_t *timg_off, const int32_t *pairs, const int64_t *out_off,
                       int32_t *d2, double thresh, const int64_t *surv_off, int32_t *surv_q,
                       int32_t *surv_t, double *surv_metric, int32_t *surv_cnt, int n_pairs,
                       int32_t *zero_div, int32_t *n_unresolved, void *stream);

/* ------------------------------------------------------------------------------------
 * K2, symmetric form (shipped for batches that hold both directions of every image pair, which
 * is what bidirectional_pair_matches asks for, scripts/lib/matcher.py:304-347): ONE MFMA sweep
 * of the distance matrix of an unordered pair serves both directions.  The sweep only produces,
 * for every row of both images, a lower bound of the best and an upper bound of the second
 * squared distance; the reference's test d0*(d0/d1) < thresh (:253-263) is monotone in both,
 * so iamx_knn2sym_candidates keeps a superset of its survivors and iamx_knn2sym_exact
 * recomputes exactly those rows against the whole train image (lowest train row on ties, like
 * cv2.BFMatcher) and re-applies the test.  What leaves the path -- surv_q / surv_t /
 * surv_metric per ordered pair in ascending query order, d2 of the survivors, zero_div -- is
 * identical to iamx_knn2_l2_pairs + iamx_match_metric + iamx_match_compact.
 *
 * Store ("desc3"): the rows of an image sorted by sn2 = |a-128|^2 (stable), zero-padded to
 * rows_cap = iamx_desc3_rows_cap(n) (a multiple of 128) rows:
 *   dst DEV [rows_cap][128] int8; sn2 / sct / sperm / sinv DEV [rows_cap] int32 with
 *   sct = (sn2 + 2*sum(a-128)) >> 1 (2^30-ish on padding), sperm = original row of a sorted
 *   row (-1 on padding), sinv = sorted position of an original row.
 *   scratch DEV [3 * n_rows] int32 (3 * total_rows for the batch form, laid out like
 *   iamx_desc2_pack_batch_u8).
 * Sweep: upairs DEV [n_u][2] = (B image, A image): B rows stay in registers (1024 / 512 / 256
 * per workgroup for form 2 / 1 / 0 = iamx_knn2sym_rows_per_wg(form)), A rows stream through LDS.
 *   wg_off   DEV [n_u+1] int32  scan of ceil(n_B / rows_per_wg);  total_wg = wg_off[n_u]
 *   col_off  DEV [n_u] int64    first entry of pair u in col  (scan of rows_cap(B))
 *   rowp_off DEV [n_u] int64    first entry of pair u in rowp (scan of workgroups x rows_cap(A))
 *   col  DEV [..][2] int32, rowp DEV [..][2] int32 (8-byte aligned): the bounds, internal format
 * Candidates: pairs DEV [n_pairs][2] ordered (query image, train image); osrc DEV [n_pairs][2] =
 *   (index u of its unordered pair, role: 0 if the query image is B, 1 if it is A);
 *   out_off DEV [n_pairs+1] int64 rows of each ordered pair; keep DEV [max(rows, 8)] uint8 scratch
 *   (4-byte aligned; used as a bit map, one bit per query row of rows_total, cleared by the call);
 *   cand_cnt DEV [n_pairs] (written); cand_q DEV [rows]: the candidate query rows (original
 *   numbering, ascending) of pair p at out_off[p] .. + cand_cnt[p] -- a pair's list lives in
 *   its own slice of the row range, no scan over the pairs;
 *   d2 DEV [rows][2] int32 (the array iamx_knn2sym_exact fills): entry [r][1] of a candidate row
 *   receives an upper bound of its exact second squared distance, which the exact stage prunes its
 *   scan with before it replaces the pair of values.
 *   task_total DEV [2] (must be 0 on entry; iamx_knn2sym_exact leaves it 0), tasks DEV
 *   [2 n_pairs + rows/32 + 2][2]: the tasks (ordered pair, block) of the exact stage -- wave
 *   tasks (pairs with <= 64 candidates) from entry 0, workgroup tasks (256 candidates) from
 *   entry n_pairs.
 * Exact: desc / norm_q / norm_t / img_off = the ORIGINAL-order store of iamx_desc_pack_*.  A
 *   wave task: <= 64 candidates x the whole train image on the MFMA, train tiles straight from
 *   L2; a workgroup task: 4 x 64 candidates, every train tile staged once in LDS for the four
 *   waves (the form real frames need: a third to two thirds of a pair's rows are candidates
 *   there, against 0.1 % on uncorrelated synthetic descriptors).  Writes d2 DEV [rows][2]
 *   for the candidate rows, then compacts every pair's list in place to its survivors: pair p
 *   owns cand_q / cand_t / cand_metric [out_off[p] .. + surv_cnt[p]); zero_div is incremented
 *   for every row whose exact second distance is 0 (matcher.py:255 divides by it).
 * ------------------------------------------------------------------------------------ */
int64_t iamx_desc3_rows_cap(int64_t n_rows);
int iamx_knn2sym_rows_per_wg(int form);
/* the sweep kernel a form launches, spelled the way rocprofv3 prints it ("knn2sym_kernel<4, 8, ...>"):
 * profile summaries are matched against the binary that is loaded (bench.py refuses stale ones) */
const char *iamx_knn2sym_kernel_id(int form);
int iamx_desc3_pack_u8(const uint8_t *src, int64_t n_rows, int8_t *dst, int32_t *sn2,
                       int32_t *sct, int32_t *sperm, int32_t *sinv, int32_t *scratch,
                       void *stream);
int iamx_desc3_pack_f32(const float *src, int64_t n_rows, int8_t *dst, int32_t *sn2,
                        int32_t *sct, int32_t *sperm, int32_t *sinv, int32_t *scratch,
                        void *stream);
int iamx_desc3_pack_batch_u8(const uint8_t *src, const int64_t *src_off, const int32_t *dst_off,
                             int n_img, int64_t total_rows, int max_rows_per_image, int8_t *dst,
                             int32_t *sn2, int32_t *sct, int32_t *sperm, int32_t *sinv,
                             int32_t *scratch, void *stream);
/* Narrow exact stage (round 6).  The sweep knows WHERE a candidate's two smallest distances can
 * be: a query that was a B row has eight group minima over the streamed image (group = (32-row
 * tile & 3, lane half): colmask DEV [col rows] uint8 receives the groups whose minimum is <= v2),
 * a query that was an A row one (L, U1, U2) per row block of the register-resident image (block w
 * can hold a row at or below the upper bound of the second distance only if L_w <= U2).  With
 * nar != NULL (DEV, iamx_knn2sym_narrow_bytes(rows_total, n_pairs) bytes, rows_total = rows of
 * all ordered pairs = out_off[n_pairs]; its first 256 bytes must be 0 on the first call, the exact
 * stage leaves them 0) iamx_knn2sym_candidates turns the candidates of every ordered pair with more
 * than 64 of them (train image >= 1024 rows, <= 64 row blocks) into items (candidate, class) bucketed
 * by class, and iamx_knn2sym_exact scans ONE class per task -- 1/8 of the train image (column
 * direction) or one row block (row direction) instead of all of it -- and merges a candidate's
 * items.  Results are those of the full scan: every row that can be the best, the second or tied
 * with either lies in a class of the candidate's mask.  Pairs whose items do not fit the buffer take
 * the full scan.  The same nar / colmask / form must be given to both calls; nar == NULL (or the
 * environment switch IAMX_EXACT_NARROW=0, read by both) = full scan for every pair. */
int64_t iamx_knn2sym_narrow_bytes(int64_t rows_total, int n_pairs);
int iamx_knn2sym_sweep(const int8_t *sdesc, const int32_t *sn2, const int32_t *sct,
                       const int32_t *img_off, const int32_t *img_n, const int32_t *upairs,
                       const int32_t *wg_off, const int64_t *col_off, const int64_t *rowp_off,
                       int n_u, int total_wg, int form, int32_t *col, int32_t *rowp,
                       uint8_t *colmask /* may be NULL */, void *stream);
int iamx_knn2sym_candidates(const int32_t *sn2, const int32_t *sperm, const int32_t *img_off,
                            const int32_t *img_n, const int32_t *pairs, const int32_t *osrc,
                            const int32_t *wg_off, const int64_t *col_off, const int64_t *rowp_off,
                            const int64_t *out_off, const int32_t *col, const int32_t *rowp,
                            int n_pairs, double thresh, uint8_t *keep, int32_t *cand_cnt,
                            int32_t *cand_q, int32_t *task_total, int32_t *tasks, int32_t *d2,
                            const uint8_t *colmask, void *nar /* may be NULL */, int64_t rows_total,
                            int max_query_rows /* rows of the largest query image */, int form,
                            void *stream);
/* key_t DEV [total_rows] int32 scratch beside norm_t (total_rows = rows of the whole original-
 * order store): rewritten by every call with the per-row constant of the packed (distance, row)
 * key.  task_total DEV [2], tasks DEV [2 n_pairs + rows / 32 + 2][2] (iamx_knn2sym_candidates
 * fills them: a pair with <= 64 candidates is one WAVE task, entries from 0; a pair with more
 * gets WORKGROUP tasks of 256 candidates, entries from n_pairs -- four waves share every train
 * tile through LDS; both counters are reset by this call).  sdesc / sn2 / sct / sperm / img_off3 =
 * the SORTED store of the sweep, osrc as for iamx_knn2sym_candidates: read by the narrow stage only
 * (may be NULL with nar == NULL). */
int iamx_knn2sym_exact(const int8_t *desc, const int32_t *norm_q, const int32_t *norm_t,
                       int32_t *key_t, int64_t total_rows,
                       const int32_t *img_off, const int32_t *img_n, const int32_t *pairs,
                       const int64_t *out_off, const int32_t *cand_cnt, int32_t *task_total,
                       const int32_t *tasks, int32_t *cand_q, int n_pairs, double thresh,
                       int32_t *d2, int32_t *cand_t, double *cand_metric, uint8_t *cand_keep,
                       int32_t *surv_cnt, int32_t *zero_div, const int8_t *sdesc,
                       const int32_t *sn2, const int32_t *sct, const int32_t *sperm,
                       const int32_t *img_off3, const int32_t *osrc, void *nar /* may be NULL */,
                       int64_t rows_total, int form, void *stream);

/* ------------------------------------------------------------------------------------
 * Per-pair match filters on the device, both directions of n_pairs image pairs, one workgroup
 * per pair -- replaces the python between the metric threshold and find_matches' bookkeeping:
 *   scripts/lib/matcher.py:258-269 (stable sort by metric, clip 2000), :271-283 (< min_pairs),
 *   :285 cv2.xfeatures2d.matchGMS(withRotation=True, withScale=False, thresholdFactor),
 *   :157-182 filter_duplicates, :296-299, :304-318 (reverse only if forward kept >= min_pairs),
 *   :187-200 filter_cross_check.
 * Ordered pair k (k < n_pairs) is the forward direction of pair k, ordered pair n_pairs + k its
 * reverse: surv_off DEV [2 n_pairs + 1], surv_cnt DEV [2 n_pairs], pairs DEV [2 n_pairs][2]
 * (image slots) and surv_q/_t/_metric are the outputs of the matching stage.
 *   kp_off DEV [n_images] int64   first keypoint of an image slot in xy / key2
 *   xy     DEV [total kp][2] f32  kp.pt (full-res pixels, inside [0,width) x [0,height))
 *   key2   DEV [total kp][2] i32  round-half-even(100 * kp.pt): the "%.2f" keys of :166-167
 *   out_cnt DEV [n_pairs], out_pairs DEV [n_pairs][clip][2]: the cross-checked forward list
 *   [query row, train row] (the reverse list is its mirror); scratch DEV [n_pairs][2][clip][2];
 *   out_stat DEV [n_pairs][4]: forward after GMS / after de-dup, reverse after GMS / after
 *   de-dup (-1 = stage not reached); status DEV [n_pairs]: 1 = a direction has more than 2^24
 *   survivors and must take the host path.  clip = iamx_match_postfilter_clip() = 2000.
 * ------------------------------------------------------------------------------------ */
int iamx_match_postfilter_clip(void);
/* The per-pair results of a batch packed back to back (what find_matches downloads: on an
 * all-pairs schedule almost every pair ends with nothing and the [n_pairs][clip] slots are 130 MB
 * per 4096 pairs).  cnt / status / pairs as written by iamx_match_postfilter, z DEV
 * [n_pairs][clip] of iamx_triangulate_pairs or NULL.  off [n_pairs + 1] int64: exclusive scan of
 * the counts of the pairs with status 0 (off[n_pairs] = total); out_pairs [cap][2], out_z [cap]
 * (NULL without z): pair k's rows at [off[k], off[k] + cnt[k]) when off[k] + cnt[k] <= cap.
 * off / out_pairs / out_z may be page-locked HOST memory (written by the device directly). */
int iamx_match_pack_results(const int32_t *cnt, const int32_t *status, const int32_t *pairs,
                            const double *z, int n_pairs, int clip, int64_t cap, int64_t *off,
                            int32_t *out_pairs, double *out_z, void *stream);
int iamx_match_postfilter(const int64_t *surv_off, const int32_t *surv_cnt, const int32_t *surv_q,
                          const int32_t *surv_t, const double *surv_metric, const int32_t *pairs,
                          const int64_t *kp_off, const float *xy, const int32_t *key2, int n_pairs,
                          double width, double height, double min_pairs, double threshold_factor,
                          int32_t *out_cnt, int32_t *out_pairs, int32_t *scratch,
                          int32_t *out_stat, int32_t *status, void *stream);

/* iamx_thp_pays -- HOST.  1 when MADV_HUGEPAGE speeds up the first touch of fresh anonymous memory
 * on this host right now, 0 when it slows it down (a fragmented host makes every such fault wait
 * for the kernel's compaction); probed once per process with 2 x 32 MiB, IAMX_THP=0 / 1 overrides.
 * The work arrays of iamx_link_matches and the package's result arrays (matchpairs.empty_huge,
 * what find_matches' rounds land in: scripts/lib/matcher.py:918-1031) go by it. */
int iamx_thp_pays(void);

/* ------------------------------------------------------------------------------------
 * match consolidation (host) and initial triangulation (device).
 *
 * iamx_link_matches -- scripts/lib/match_cleanup.py:246-301 link_matches(): HOST arrays in and
 *   out (no device involved).  Chain i of the input = points [ptr[i], ptr[i+1]) of (img[],
 *   kp[]); the linked chains come back in the reference's order (the caller sorts by length),
 *   out arrays sized like the input.  Returns the number of chains or a negative error code.
 * iamx_triangulate_ground -- match_cleanup.py:320-347 triangulate_smart(): per feature the mean
 *   of the ground-plane intersections of its observation rays.
 *   M DEV [n_images][9] = (body2ned . cam2body) . inv(K) row major, ned DEV [n_images][3],
 *   base_elev DEV [n_images], obs_img DEV [n_obs] int32, obs_uv DEV [n_obs][2] f64,
 *   feat_ptr DEV [n_feat+1] int64, out_ned DEV [n_feat][3], n_sky DEV [1] (+= rays that point
 *   above the horizon; they add nothing to the sum but count in the divisor, like the reference)
 * ------------------------------------------------------------------------------------ */
int64_t iamx_link_matches(const int32_t *img, const int32_t *kp, const int64_t *ptr,
                          int64_t n_matches, int32_t *out_img, int32_t *out_kp, int64_t *out_ptr,
                          int32_t *n_passes);
/* iamx_link_pair_blocks -- the same for the pair matches of make_match_structure()
 * (match_cleanup.py:190-215) handed over as they lie in the images' match lists: blocks[b] = HOST
 * int32 [counts[b]][2] (keypoint of image ij[2b], keypoint of image ij[2b+1]), pairs in the
 * reference's order; out arrays hold 2 * sum(counts) points / sum(counts) + 1 offsets. */
int64_t iamx_link_pair_blocks(const int32_t *const *blocks, const int64_t *counts, const int32_t *ij,
                              int64_t n_blocks, int32_t *out_img, int32_t *out_kp, int64_t *out_ptr,
                              int32_t *n_passes);
/* iamx_chain_members_uv -- HOST.  The "replace keypoint indices with uv coordinates" step of
 * link_matches (scripts/lib/match_cleanup.py:277-287) for all chain members at once: uv[k] =
 * (double) kp.pt of keypoint f_kp[k] of image f_img[k], from the images' own position arrays
 * (xy[i] float32 [n_kp[i]][2]).  Negative indices count from the end like a python list's; an index
 * outside the image's list -> IAMX_EINVAL, *bad_member = the first such member. */
int iamx_chain_members_uv(const int32_t *f_img, const int32_t *f_kp, int64_t n_members,
                          const float *const *xy, const int64_t *n_kp, int32_t n_images, double *uv,
                          int64_t *bad_member, int threads);
/* iamx_kp_key2 -- HOST.  key2[k] = round-half-even(100 * xy[k]) in exact integer arithmetic: the
 * "%.2f-%.2f" % kp.pt keys of scripts/lib/matcher.py:166-167 and match_cleanup.py:36-38 as integer
 * pairs (xy float32 [n][2] in [0, 16384), key2 int32 [n][2]).
 * iamx_kp_dup_remap -- HOST.  merge_duplicates' per-image table (match_cleanup.py:19-60) for
 * n_images images at once: remap[k] (index inside the image) = the first USED keypoint of the
 * image with keypoint k's key, k itself when k is unused; identity[i] = 1 when image i has no two
 * used keypoints on one pixel.  Flat arrays, image i at [kp_base[i], kp_base[i+1]). */
int iamx_kp_key2(const float *xy, int64_t n, int32_t *key2);
int iamx_kp_dup_remap(const float *xy, const uint8_t *used, const int64_t *kp_base, int32_t n_images,
                      int32_t *remap, uint8_t *identity, int threads);
/* iamx_match_lists_scan -- HOST.  One pass over the images' match lists for the per-list loops in
 * front of link_matches (scripts/lib/match_cleanup.py:19-188, lib/project.py:331-350
 * compute_kp_usage): list b = int32 [cnt[b]][2] (keypoint of image ia[b], keypoint of image ib[b]),
 * keypoints of image i at [kp_base[i], kp_base[i+1]) of the flat arrays `used` (uint8) / `remap`.
 * mode & 1: mark both keypoints used; & 2: replace both by remap[...] in place (merge_duplicates);
 * & 4: dup_pairs[b] = rows equal to an earlier row, dup_first[b] = rows whose first keypoint
 * occurred before (check_for_pair_dups / check_for_1vn_dups).  A keypoint index outside its image
 * is IAMX_EINVAL (the reference raises IndexError there). */
int iamx_match_lists_scan(int32_t *const *lists, const int64_t *cnt, const int32_t *ia,
                          const int32_t *ib, int64_t n_lists, const int64_t *kp_base, int32_t n_images,
                          uint8_t *used, const int32_t *remap, int mode, int32_t *dup_pairs,
                          int32_t *dup_first, int threads);
/* HOST helpers of the same stage.  iamx_chains_longest_first: the chains of iamx_link_matches in
 * the order match_cleanup.py:291-292 leaves them (stable sort by length, longest first), members
 * copied on `threads` threads; out arrays sized like the input, out_ptr [n_chains + 1].
 * iamx_first_occurrence: first[k] = smallest j with key[j] == key[k] (merge_duplicates,
 * match_cleanup.py:19-104: keypoints of an image on the same pixel collapse onto the first). */
int iamx_chains_longest_first(const int32_t *img, const int32_t *kp, const int64_t *ptr,
                              int64_t n_chains, int32_t *out_img, int32_t *out_kp, int64_t *out_ptr,
                              int threads);
int iamx_first_occurrence(const int64_t *key, int64_t n, int64_t *first);
/* iamx_ledger_index -- HOST arrays.  matcher.find_matches (scripts/lib/matcher.py:978-979) gives
 * BOTH images of every processed pair a match_list entry, in processing order; for the pairs
 * without matches -- 95-99 % of an all-pairs schedule -- this package keeps index arrays (qi, qj,
 * seq ascending) and this routine turns them into the per-image lists the .match writer walks:
 * image k's partners and their seq at [bounds[k], bounds[k+1]) of other / seq_out (2 m entries),
 * bounds [n_images + 1].  One counting sort. */
int iamx_ledger_index(const int64_t *qi, const int64_t *qj, const int64_t *seq, int64_t m,
                      int64_t n_images, int64_t *other, int64_t *seq_out, int64_t *bounds);
/* iamx_pairs_fwd_rev, iamx_segment_mean_std -- HOST arrays, `threads` host threads.  The bulk
 * work of one round of find_matches between the device's packed result buffer and the match
 * lists of scripts/lib/matcher.py:978-979: fwd = the n (query, train) int32 rows copied out of the
 * page-locked buffer, rev = the same rows with the columns swapped (the list of the reversed
 * direction); and, per pair, the mean and standard deviation of its triangulated heights
 * (lib/smart.py:117-130 estimate_surface_elevation: np.mean / np.std of the pair's values) over
 * the segments z[starts[s] .. + counts[s]) of the round's packed heights, summed in numpy's
 * order. */
int iamx_pairs_fwd_rev(const int32_t *src, int64_t n, int32_t *fwd, int32_t *rev, int threads);
/* reads one byte of every 4 KiB page of a HOST buffer on `threads` threads (first touch of the
 * page-locked landing buffers of find_matches' rounds, off the critical path) */
int iamx_touch_pages(const void *p, int64_t bytes, int threads);
int iamx_segment_mean_std(const double *z, const int64_t *starts, const int64_t *counts,
                          int64_t n_seg, int64_t n_z, double *mean, double *std, int threads);

/* iamx_hbm_copy16 -- measurement yardstick of the HBM-bound kernels (bench.py, tools/): a plain
 * grid-stride device kernel that writes n16 16-byte words, each the sum of reads_per_write (1..4)
 * source words (src holds reads_per_write x n16 words: 1 = a copy, 3 = the read : write mix of the
 * BA residual kernel), 16 bytes per lane and step, `workgroups` x 256 threads.  No counterpart in
 * the reference. */
int iamx_hbm_copy16(const void *src, void *dst, int64_t n16, int reads_per_write, int workgroups,
                    void *stream);

/* iamx_yaw_feedback_* -- the yaw-error FEEDBACK of the reference's pair loop as a prefix
 * computation over the schedule (HOST code).  scripts/lib/matcher.py:987-993 sets both images'
 * aircraft yaw-error estimate after every pair (lib/smart.py:251-283 update_yaw_error_estimate:
 * 0 without matches / without a similarity fit, else the weighted average over the image's
 * yaw_pairs entries -- values rounded through "%.1f", weights truncated like getInt, children in
 * sorted-name order, entries closer than 0.5 m or more than 30 degrees off skipped), and
 * lib/image.py:434-457 turns it into the camera pose the image's NEXT pair triangulates with.
 * _new: name_rank [n_images] rank of every image's name in sorted order; _seed: entries an image's
 * yaw_pairs node holds before the call; _feed: one round in schedule order (see the definition);
 * _state: every image's current estimate and whether a pair of the call has touched it. */
void *iamx_yaw_feedback_new(int n_images, const int32_t *name_rank);
void iamx_yaw_feedback_free(void *h);
int iamx_yaw_feedback_seed(void *h, int image, int n, const int32_t *partner_image, const double *err,
                           const double *weight, const double *dist);
int iamx_yaw_feedback_feed(void *h, int64_t n, const int32_t *pi, const int32_t *pj,
                           const uint8_t *quiet, int64_t n_hits, const int64_t *hit_rows,
                           const double *yv_f, const double *yv_r, const uint8_t *ok, double *e1,
                           double *e2, uint8_t *fresh1, uint8_t *fresh2);
int iamx_yaw_feedback_state(void *h, double *value, uint8_t *touched);

/* iamx_group_level -- one group level of scripts/lib/groups.py:59-118 compute() (HOST arrays):
 * seed chain + sweeps until nothing can be added.  level [n_matches] in/out (-1 = unused),
 * placed_images [n_images] 0/1 from earlier levels, placed_matches [n_images] out.  Returns the
 * seed chain index, -1 if there is none, or a negative error code < -1. */
int64_t iamx_group_level(const int32_t *img, const int64_t *ptr, int64_t n_matches, int n_images,
                         int32_t *level, const uint8_t *placed_images, int group_level,
                         int use_single_pairs, int max_wanted, int min_connections,
                         int32_t *placed_matches);
int iamx_triangulate_ground(const double *M, const double *ned, const double *base_elev,
                            int n_images, const int32_t *obs_img, const double *obs_uv,
                            const int64_t *feat_ptr, int64_t n_feat, double *out_ned,
                            int32_t *n_sky, void *stream);

/* iamx_triangulate_pairs -- scripts/lib/smart.py:26-63 triangulate_features(): two-view linear
 * (DLT) triangulation of the matches of every pair of a batch, what cv2.triangulatePoints
 * computes (null direction of the 4x4 system, f64), w-normalised; only the NED "down" component
 * is returned because estimate_surface_elevation() (:117-130) needs nothing else.
 *   pair_img DEV [n_pairs][2] image slots, PROJ DEV [n_images][12] = [R | t] row major,
 *   IK DEV [9] inverse camera matrix, kp_off / xy as for iamx_match_postfilter,
 *   m_cnt DEV [n_pairs], m_pairs DEV [n_pairs][clip][2] (query row, train row),
 *   out_z DEV [n_pairs][clip] */
int iamx_triangulate_pairs(const int32_t *pair_img, const double *PROJ, const double *IK,
                           const int64_t *kp_off, const float *xy, const int32_t *m_cnt,
                           const int32_t *m_pairs, int n_pairs, int clip, double *out_z,
                           void *stream);

/* iamx_triangulate_pairs_xyz -- the same triangulation with all three w-normalised NED components,
 * what `points /= points[3]` leaves in rows 0..2 of triangulate_features()'s 4xN return value
 * (scripts/lib/smart.py:61-63); out_xyz DEV [n_pairs][clip][3] */
int iamx_triangulate_pairs_xyz(const int32_t *pair_img, const double *PROJ, const double *IK,
                               const int64_t *kp_off, const float *xy, const int32_t *m_cnt,
                               const int32_t *m_pairs, int n_pairs, int clip, double *out_xyz,
                               void *stream);

/* iamx_triangulate_packed -- the triangulation of iamx_triangulate_pairs over PACKED match lists
 * with one pair of projection matrices PER PAIR: the surface stage of find_matches.  The reference
 * rewrites both images' camera poses after every pair (scripts/lib/matcher.py:990-993 ->
 * lib/image.py:434-457 set_aircraft_yaw_error_estimate) and the next pair triangulates with them
 * (lib/smart.py:26-63 via lib/image.py:542-553 get_proj); the host replays that chain in schedule
 * order and passes every pair the two matrices the reference would have used.
 *   pair_img  DEV [n_pairs][2] image slots (kp_off / xy as above)
 *   pair_proj DEV [n_pairs][2][12] float64 row-major [R | t] of image 1 / image 2 of the pair
 *   m_off     DEV [n_pairs + 1] int64 first match of every pair; m_pairs DEV [total][2]
 *   out_z     DEV [total] NED "down" of every match (w-normalised) */
int iamx_triangulate_packed(const int32_t *pair_img, const double *pair_proj, const double *IK,
                            const int64_t *kp_off, const float *xy, const int64_t *m_off,
                            const int32_t *m_pairs, int n_pairs, int64_t total, double *out_z,
                            void *stream);

/* iamx_similarity_pairs -- scripts/lib/smart.py:66-89 find_affine(): the 2x3 similarity
 * (rotation, uniform scale, translation) between the matched keypoints of every pair of a batch,
 * for estimate_yaw_error() (:138-192).  The reference asks cv2.estimateAffinePartial2D (RANSAC);
 * this is a deterministic robust fit instead: least squares on all matches, then nine re-fits on
 * the matches within 200, 50, 10, 3, 3, ... px of the current model.
 *   pair_img / kp_off / xy / m_cnt / m_pairs / clip as for iamx_triangulate_pairs
 *   out_aff DEV [n_pairs][2][6] float64 row major: [p][0] maps image b's pixels onto image a's
 *   (find_affine(a, b)), [p][1] the other way; out_ok DEV [n_pairs][2]: 1 = a fit exists
 *   (>= 2 matches that do not coincide) */
int iamx_similarity_pairs(const int32_t *pair_img, const int64_t *kp_off, const float *xy,
                          const int32_t *m_cnt, const int32_t *m_pairs, int n_pairs, int clip,
                          double *out_aff, int32_t *out_ok, void *stream);

/* out[i] = sum_{j<i} in[j], out[n] = total; in DEV [n] int32, out DEV [n+1] int64 */
int iamx_exclusive_scan_i32(const int32_t *in, int64_t n, int64_t *out, void *stream);

/* ------------------------------------------------------------------------------------
 * K3: bundle-adjustment reprojection residual -- replaces Optimizer.fun
 *   scripts/lib/optimizer.py:174-229 (per-camera cv2.projectPoints + concatenate).
 *   cams    DEV [n_cams][7] float64  ned(3), quat w,x,y,z (unnormalised; :326-328)
 *   pts     DEV [n_pts][3]  float64
 *   cam_idx / pt_idx  DEV [n_obs] int32   camera-major observation list (:397-404)
 *   uv      DEV [n_obs][2] float64   observed (distorted, full-res px; :383)
 *   calib   DEV [9] float64  fx, fy, cu, cv, k1, k2, p1, p2, k3   (lib/camera.py:94)
 *   r       DEV [2*n_obs] float64   (du0, dv0, du1, dv1, ...) = observed - projected
 * ------------------------------------------------------------------------------------ */
int iamx_ba_residual(const double *cams, int n_cams, const double *pts, int n_pts,
                     const int32_t *cam_idx, const int32_t *pt_idx, const double *uv,
                     int64_t n_obs, const double *calib, double *r, void *stream);

/* Same residual, faster form: each workgroup (512 observations, two per thread) builds the
 * rotation/position blocks of the cameras it touches in LDS first, then does 9 FMAs, one
 * reciprocal and the distortion polynomial per observation.  Fastest with camera-major
 * cam_idx (any order is legal).  uv and r must be 32-byte aligned, cam_idx and pt_idx 8-byte
 * aligned; cam_scratch is unused (may be NULL).  Results agree with iamx_ba_residual to
 * rounding. */
int iamx_ba_residual_prepared(const double *cams, int n_cams, const double *pts, int n_pts,
                              const int32_t *cam_idx, const int32_t *pt_idx, const double *uv,
                              int64_t n_obs, const double *calib, double *cam_scratch, double *r,
                              void *stream);

/* Residual + analytic Jacobian blocks (the reference has only finite differences:
 * scripts/lib/optimizer.py:142-169,491-501).  d r / d params, row-major per observation:
 *   Jc  DEV [n_obs][2][7]   wrt that observation's camera (ned, quat)
 *   Jp  DEV [n_obs][2][3]   wrt its 3-D point
 *   Jk  DEV [n_obs][2][8] or NULL   wrt (f, cu, cv, k1, k2, p1, p2, k3) with fx=fy=f
 *                                   ('global' calibration, optimizer.py:181-189)
 *   r may be NULL. */
int iamx_ba_residual_jac(const double *cams, int n_cams, const double *pts, int n_pts,
                         const int32_t *cam_idx, const int32_t *pt_idx, const double *uv,
                         int64_t n_obs, const double *calib, double *r,
                         double *Jc, double *Jp, double *Jk, void *stream);

/* ------------------------------------------------------------------------------------
 * Image preparation -- replaces the cv2 calls of Image.load_rgb(equalize=True) and the resize
 * in detect_features (scripts/lib/image.py:105-112,313): BGR->HSV, CLAHE(clip_limit, 8x8) on
 * V, HSV->BGR, cv2.resize(fx=fy=scale, INTER_LINEAR).
 *   bgr  DEV [height][width][3] uint8;  out  DEV [out_h][out_w][3] uint8 with
 *   (out_h, out_w) = iamx_image_resized_dims = round(size*scale);  equalize = 0 skips CLAHE.
 * ------------------------------------------------------------------------------------ */
int64_t iamx_image_prep_workspace_bytes(int height, int width);
int iamx_image_resized_dims(int height, int width, double scale, int *out_h, int *out_w);
int iamx_image_equalize_resize(const uint8_t *bgr, int height, int width, int equalize,
                               float clip_limit, double scale, void *workspace,
                               int64_t workspace_bytes, uint8_t *out, void *stream);

/* ------------------------------------------------------------------------------------
 * K1: SIFT detect + describe -- replaces cv2.SIFT_create().detectAndCompute(scaled, None)
 *   scripts/lib/image.py:235-237,324 (OpenCV defaults: 3 layers/octave, sigma 1.6, image
 *   doubled, contrastThreshold 0.04, edgeThreshold 10, no feature cap).
 *   image      DEV [height][width][channels] uint8, channels = 3 (BGR) or 1 (gray)
 *   workspace  DEV, iamx_sift_workspace_bytes(height, width) bytes (pyramids + candidates)
 *   kp         DEV [cap][8] float32: x, y (input-image px), size, angle (deg), response,
 *              packed octave (int32 bit pattern, cv2.KeyPoint.octave), 2 internal words
 *   desc       DEV [cap][128] uint8 (what cv2 returns as float32 0..255)
 *   n_out      DEV [1] int32: keypoints found (may exceed cap; only cap are stored)
 * Keypoints are appended in no particular order and still hold duplicates; iamx_sift_sort is the
 * rest of detectAndCompute: KeyPointsFilter::removeDuplicatedSorted (duplicates dropped, OpenCV's
 * output order).  Behind the float32 pyramid the arithmetic follows OpenCV's sift.simd.hpp scalar
 * code in float32 (adjustLocalExtrema with Matx33f::solve, calcOrientationHist,
 * calcSIFTDescriptor, hal::fastAtan2); oracle/sift_oracle.py states the three departures.
 * ------------------------------------------------------------------------------------ */
int64_t iamx_sift_workspace_bytes(int height, int width);
int iamx_sift_detect(const uint8_t *image, int height, int width, int channels,
                     float contrast_threshold, float edge_threshold, float sigma,
                     void *workspace, int64_t workspace_bytes, float *kp, uint8_t *desc, int cap,
                     int32_t *n_out, void *stream);

/* Where a pyramid level lives inside the workspace after iamx_sift_detect (tests / diagnosis):
 * kind 0 = Gaussian level index 0..5 of `octave` (0 = the doubled image); float32
 * [level_h][level_w] at workspace + byte_offset.  Other kinds fail: the DoG levels are not
 * stored (level i + 1 minus level i is taken where the scan and the sub-pixel fit need it). */
int iamx_sift_pyramid_level(int height, int width, int octave, int kind, int index,
                            int64_t *byte_offset, int *level_h, int *level_w, int *n_octaves);

/* KeyPointsFilter::removeDuplicatedSorted of detectAndCompute (OpenCV features2d/keypoint.cpp;
 * cv2.SIFT.detectAndCompute at scripts/lib/image.py:324 returns its result) on the lists
 * iamx_sift_detect appended: sort by KeyPoint12_LessThan -- x, y ascending, size descending, angle
 * ascending, response descending, octave descending -- and drop every keypoint equal to its
 * predecessor in (x, y, size, angle).  order 1 = that order (what cv2 returns); order 0 = the
 * pyramid-local (octave, layer, y, x, angle) order with the same duplicates dropped.
 * n_out DEV [1] as written by iamx_sift_detect; width, height: the detect image's size in pixels;
 * out_kp DEV [cap][8], out_desc DEV [cap][128] (rows [0, n_sorted)); n_sorted DEV [1] int32 =
 * rows kept; workspace DEV iamx_sift_sort_workspace_bytes(cap) bytes. */
int64_t iamx_sift_sort_workspace_bytes(int cap);
int iamx_sift_sort(const float *kp, const uint8_t *desc, const int32_t *n_out, int cap, int order,
                   int width, int height, void *workspace, int64_t workspace_bytes, float *out_kp,
                   uint8_t *out_desc, int32_t *n_sorted, void *stream);

/* ------------------------------------------------------------------------------------
 * Image ingest: split JPEG decoder (csrc/jpeg.hip) in place of cv2.imread(file, ANYCOLOR |
 * ANYDEPTH | IGNORE_ORIENTATION) of scripts/lib/image.py:99-104.  The host does the Huffman
 * decode (serial by nature), the device dequantisation + the libjpeg "islow" integer IDCT +
 * fancy chroma upsampling + YCbCr -> BGR: integer arithmetic throughout, pixels bit-identical to
 * libjpeg-turbo's (what OpenCV and Pillow decode with).  8-bit baseline / extended-sequential
 * Huffman files with one interleaved scan, grey or YCbCr 4:4:4 / 4:2:2 / 4:2:0; anything else
 * returns IAMX_EUNSUPPORTED (the caller decodes such a file the host way).
 *   iamx_jpeg_info: HOST.  info [16] int32: width, height, components, max h / v sampling,
 *     blocks_w / blocks_h of components 0..2 (8x8 blocks, padded to whole MCUs), [11] total
 *     blocks, [12] restart interval.
 *   iamx_jpeg_decode_coefficients: HOST, no device involved, thread safe.  coef HOST
 *     [coef_blocks >= info[11]][64] int16: the quantised coefficients of every block in natural
 *     (row major) order, components back to back, blocks of a component in raster order;
 *     quant HOST [3][64] uint16: the components' quantisation tables in natural order.
 *   iamx_jpeg_reconstruct: coef / quant DEV (as written above), info HOST, workspace DEV
 *     iamx_jpeg_workspace_bytes(info) bytes (the component planes), bgr DEV [height][width][3].
 * ------------------------------------------------------------------------------------ */
int iamx_jpeg_info(const uint8_t *data, int64_t len, int32_t *info);
int iamx_jpeg_decode_coefficients(const uint8_t *data, int64_t len, int16_t *coef,
                                  int64_t coef_blocks, uint16_t *quant);
int64_t iamx_jpeg_workspace_bytes(const int32_t *info);
int iamx_jpeg_reconstruct(const int16_t *coef, const uint16_t *quant, const int32_t *info,
                          void *workspace, int64_t workspace_bytes, uint8_t *bgr, void *stream);

/* The reference's descriptor cache file (gzip of np.save(float32 [N, 128]),
 * scripts/lib/image.py:205-217) written straight from the detector's uint8 descriptors: HOST,
 * no device involved, thread safe.  out receives ONE gzip member whose payload is `header` (the
 * .npy header) followed by the little-endian float32 values of `values`; any gzip reader gets the
 * bytes np.save would have written (a dynamic-Huffman DEFLATE block of literals built from the
 * histogram of the 256 possible values: ~25 ms per 50 k-keypoint frame instead of ~0.5 s of
 * zlib).  Returns the member's size in bytes (out_cap >= iamx_gzip_f32_from_u8_bound(...)) or a
 * negative error code. */
int64_t iamx_gzip_f32_from_u8_bound(int64_t n_header, int64_t n_values);
int64_t iamx_gzip_f32_from_u8(const uint8_t *header, int64_t n_header, const uint8_t *values,
                              int64_t n_values, uint8_t *out, int64_t out_cap);

/* The other cache-file byte work of Image.save_features / save_descriptors / detect_features
 * (scripts/lib/image.py:187-217, 324-346), HOST only, one call each so that no interpreter lock
 * is held or handed around while the bytes are produced (threads are created inside):
 *   iamx_gzip_members   gzip.open(..., compresslevel=level).write(...) as a multi-member gzip
 *                       stream (members of <= member_bytes of input, compressed in parallel with
 *                       zlib at `level`, `strategy` 0 default .. 4 fixed as in zlib.h); bufs / lens: the buffers that follow each other (an .npy header and
 *                       the array's own memory, or one pickle).  Returns bytes written
 *                       (out_cap >= iamx_gzip_members_bound(total, members)) or a negative code.
 *   iamx_u8_to_f32      des_list = float32 of the detector's uint8 descriptors.
 *   iamx_feat_records   the .feat pickle's records "( G x G y TUPLE2 G size G angle G response
 *                       J octave J class_id t" (58 bytes per keypoint) from the columns. */
int64_t iamx_gzip_members_bound(int64_t total_bytes, int64_t n_members);
int64_t iamx_gzip_members(const uint8_t *const *bufs, const int64_t *lens, int n_bufs,
                          int64_t member_bytes, int level, int strategy, int threads, uint8_t *out,
                          int64_t out_cap);
/* iamx_gzip_records -- HOST.  iamx_gzip_members for streams of fixed-width records (the .feat
 * pickle of scripts/lib/image.py:192-203: 58 bytes per keypoint): a DEFLATE encoder that only
 * looks for matches at distance record_bytes (one compare pass, dynamic Huffman code, no hash
 * chains) -- valid gzip members, the same payload for every reader, smaller than zlib's level 4
 * output on such streams and several times faster.  out_cap >= iamx_gzip_records_bound(total
 * input bytes, number of members). */
int64_t iamx_gzip_records_bound(int64_t total, int64_t n_members);
int64_t iamx_gzip_records(const uint8_t *const *bufs, const int64_t *lens, int n_bufs,
                          int64_t member_bytes, int record_bytes, int threads, uint8_t *out,
                          int64_t out_cap);
int iamx_u8_to_f32(const uint8_t *src, float *dst, int64_t n, int threads);
/* the other way for a whole survey: the float32 des_list of n images (scripts/lib/image.py:324,
 * integer valued) -> uint8 back to back in dst (HOST), threads of its own; srcs HOST [n] pointers,
 * counts HOST [n] elements per image.  Feeds ONE upload + the batched pack kernels. */
int iamx_f32_to_u8_many(const float *const *srcs, const int64_t *counts, int n, uint8_t *dst,
                        int threads);
/* n host byte blocks (counts[i] bytes at srcs[i]) copied back to back into dst by `threads` threads:
 * the uint8 descriptor arrays of a group of images into one page-locked staging buffer. */
int iamx_u8_gather_many(const uint8_t *const *srcs, const int64_t *counts, int n, uint8_t *dst,
                        int threads);
int iamx_feat_records(const float *x, const float *y, const float *size, const float *angle,
                      const float *response, const int32_t *octave, const int32_t *class_id,
                      int64_t n, uint8_t *out);
/* The pair lists of the .match pickles (Image.save_matches, scripts/lib/image.py:222-233): list k =
 * rows [off[k], off[k+1]) of pairs [.][2] as the protocol-2 stream "] ( { ] ( int int e }* e" (no
 * memo entries), an empty list as "]"; out_off [n_lists + 1] receives where each list's bytes start
 * (out_cap >= sum of 3 + 13 * rows).  HOST only.  Returns bytes written or a negative code. */
int64_t iamx_pickle_pair_lists(const int32_t *pairs, const int64_t *off, int64_t n_lists, uint8_t *out,
                               int64_t out_cap, int64_t *out_off);

/* ------------------------------------------------------------------------------------
 * K4: linear algebra on the device-resident block Jacobian (what SciPy's TRF/LSMR does on
 * the sparse matrix the reference gives it: scripts/lib/optimizer.py:491-501,
 * scipy/optimize/_lsq/trf.py:205-400).  n = 7*n_cams + 3*n_pts (+8 with Jk), m = 2*n_obs.
 *   iamx_ba_jv   y[m] = J x
 *   iamx_ba_jtv  out[n] = J^T u            (square = 0)
 *                out[n] = column sums of J.^2 (square != 0; u unused) -- x_scale='jac'
 *     cam_ptr DEV [n_cams+1] int32: observations of camera c are [cam_ptr[c], cam_ptr[c+1])
 *             (the camera-major order of optimizer.py:397-404)
 *     pt_ptr  DEV [n_pts+1], pt_obs DEV [n_obs] int32: observation ids grouped by point
 *     scratch DEV [2048] float64 (only read/written when Jk != NULL)
 *   No atomics: results are bitwise reproducible.
 * ------------------------------------------------------------------------------------ */
int iamx_ba_jv(const double *Jc, const double *Jp, const double *Jk, const int32_t *cam_idx,
               const int32_t *pt_idx, int64_t n_obs, int n_cams, int n_pts, const double *x,
               double *y, void *stream);
int iamx_ba_jtv(const double *Jc, const double *Jp, const double *Jk, const int32_t *cam_ptr,
                const int32_t *pt_ptr, const int32_t *pt_obs, int64_t n_obs, int n_cams,
                int n_pts, const double *u, int square, double *out, double *scratch,
                void *stream);

/* Host-free, matrix-free LSMR (scipy/sparse/linalg/_isolve/lsmr.py) on
 * A = [J diag(d); diag(dreg)], b = [r; 0], no calibration columns.  J is the analytic Jacobian
 * of iamx_ba_residual_jac at (cams, pts, calib); the kernels re-derive an observation's 2x10
 * block from its camera and point instead of reading a stored copy.
 * iamx_ba_lsmr_prepare: once per solve, the tables
 *   ctab DEV [n_cams][32] (rotation, position, quaternion, 1/|q|^2, d of the 7 columns),
 *   ptab DEV [n_pts][6]  (X, d of the 3 columns).
 * iamx_ba_lsmr_iterate: enqueues n_iter (even) iterations, three launches each, no host
 *   synchronisation.  `state` (DEV, iamx_ba_lsmr_state_size() doubles) holds the double-
 *   buffered scalar recurrences, the tolerances and the latched results (layout and
 *   initialisation: imageanalysis_amd/ba_solver.py); once istop is latched the remaining
 *   iterations are no-ops.  u1 [2 n_obs], u2/vt/h/hbar/x [n] DEV work vectors; partials DEV
 *   [iamx_ba_lsmr_partials_size]; xr DEV [4] (sums that cross kernels), tbuf DEV [n] (raw J^T ut1).
 *   pt_idx [n_obs] camera-major; cam_ptr / pt_ptr / pt_obs as for iamx_ba_jtv; slot_cp DEV
 *   [n_obs][2] int32 = (camera, point) of the observation in point-sorted slot e = pt_obs[e].
 *   eprod DEV [n_obs][3] float64 work buffer: the point part of J^T ut' per observation, written
 *   by the forward kernel (which has the observation's geometry and the new ut' in registers) and
 *   summed per point by the adjoint kernel.
 *   Deterministic (fixed reduction trees, no atomics). */
int iamx_ba_lsmr_state_size(void);
int64_t iamx_ba_lsmr_partials_size(int n_cams, int n_pts);
int iamx_ba_lsmr_prepare(const double *cams, const double *pts, const double *d, int n_cams,
                         int n_pts, double *ctab, double *ptab, void *stream);
int iamx_ba_lsmr_iterate(const double *ctab, const double *ptab, const double *calib,
                         const int32_t *pt_idx, const int32_t *cam_ptr, const int32_t *pt_ptr,
                         const int32_t *pt_obs, const int32_t *slot_cp, int64_t n_obs, int n_cams,
                         int n_pts, const double *dreg, double *u1, double *u2, double *vt,
                         double *h, double *hbar, double *x, double *state, double *partials,
                         double *xr, double *tbuf, double *eprod, int n_iter, void *stream);

/* Multi-rank form of the same iteration: observations AND the point part of every n-vector are
 * sharded by point (this rank owns the points [pt_lo, pt_hi) of the internal order and every
 * observation of them); only the camera part (7 n_cams entries) is replicated.  One call per
 * phase; the caller all-reduces (sum, RCCL, same stream) xr[0..2) after phase 0 and
 * tbuf[0 .. 7 n_cams] (7 n_cams + 1 doubles: 157 KB at BASELINE configs[3]) after phase 1.
 *   phase 0: ut', raw camera part of J^T ut1' -> tbuf; xr[0] / xr[1] = this rank's parts of
 *            |ut'|^2 / |x|^2, xr[2] / xr[3] = their replicated camera parts
 *   phase 1: stopping tests of the previous iteration, beta', point part of vt' (rank local),
 *            tbuf[7 n_cams] = its sum of squares
 *   phase 2: camera part of vt' from the reduced tbuf, alpha', plane rotations, h / hbar / x
 * parity = iteration & 1.  xr DEV [4], tbuf DEV [n].  The point entries of x that belong to
 * other ranks are never touched (all-reduce x[7 n_cams ..) once per solve when x started 0).
 * In iamx_ba_lsmr_iterate (single rank) xr is DEV [4] as well. */
int iamx_ba_lsmr_phase(const double *ctab, const double *ptab, const double *calib,
                       const int32_t *pt_idx, const int32_t *cam_ptr, const int32_t *pt_ptr,
                       const int32_t *pt_obs, const int32_t *slot_cp, int64_t n_obs, int n_cams,
                       int n_pts, int pt_lo, int pt_hi, const double *dreg, double *u1, double *u2,
                       double *vt, double *h, double *hbar, double *x, double *state,
                       double *partials, double *xr, double *tbuf, double *eprod, int phase,
                       int parity, void *stream);

/* ------------------------------------------------------------------------------------
 * Normal-equation blocks and the Schur-complement Gauss-Newton step (csrc/ba_schur.hip).
 *
 * iamx_ba_accumulate (the J^T J / J^T r accumulators of BASELINE.json's north star):
 *   one pass over the stored block Jacobian of iamx_ba_residual_jac emits
 *     U  DEV [n_cams][7][7]  = sum over the camera's observations of Jc^T Jc
 *     V  DEV [n_pts][3][3]   = sum over the point's observations of Jp^T Jp
 *     gc DEV [n_cams][7]     = sum Jc^T r,     gp DEV [n_pts][3] = sum Jp^T r
 *   i.e. the block diagonal of J^T J and the gradient J^T r that scipy's TRF derives from the
 *   sparse J the reference gives it (scripts/lib/optimizer.py:142-169, 491-501;
 *   scipy/optimize/_lsq/trf.py:241-247 g = compute_grad(J, f), compute_jac_scale).  diag(U),
 *   diag(V) are the column sums of J.^2 of x_scale='jac'.  The off-diagonal blocks W = Jc^T Jp
 *   (7x3 per observation) are never stored: the solve below applies them from Jc / Jp.
 *   cam_ptr / pt_ptr / pt_obs as for iamx_ba_jtv.  Observations sharded by point over several
 *   ranks: U and gc are partial sums -- all-reduce them (n_cams x 56 doubles, once per outer
 *   iteration); V and gp are complete on the rank that owns the point.  No atomics.
 *
 * The trust-region subproblem  min || [J diag(d); diag(dreg)] p - [r; 0] ||  of trf.py:303-314
 * (which SciPy hands to LSMR: iamx_ba_lsmr_* keeps that form) through its normal equations:
 * points eliminated exactly, the reduced camera system S p_c = b solved by conjugate gradients
 * preconditioned with the inverses of S's 7x7 diagonal blocks, then p_p back-substituted.
 *   iamx_ba_schur_prepare: Y DEV [n_pts][6] = (D_p V D_p + Dreg_p^2)^-1 (upper triangle),
 *     yg DEV [n_pts][3] = Y D_p gp, zp DEV [n_pts][3] work, sraw DEV [n_cams][35] = the packed
 *     upper triangle of this rank's part of S_cc before regularisation (28) + its part of b (7)
 *     -- all-reduce sraw over ranks.  d, dreg DEV [n] (n = 7 n_cams + 3 n_pts).
 *   iamx_ba_schur_factor: one workgroup: M_c = (S_cc + Dreg_c^2)^-1 -> minv DEV [n_cams][28]
 *     (Cholesky; a block that is not positive definite falls back to its diagonal), x = 0,
 *     r = b, z = M r, p = z, y = d_c .* p, state (DEV, iamx_ba_schur_state_size() doubles = two
 *     buffers of 16; buffer 0 is the start:
 *     [0] r.z, [1] initial r.z, [2] iterations done, [3] stop: 0 running / 1 sqrt(r.z / r0.z0) <=
 *     eta / 2 max_iter reached / 3 breakdown / 4 the decrease of the quadratic model
 *     Q(x) = x.Sx/2 - b.x has levelled off: i (Q_{i-1} - Q_i) <= qtol |Q_i|, [4] eta,
 *     [5] max_iter, [6] alpha, [7] beta, [8] p.Sp, [9] -Q, [10] qtol)
 *   iamx_ba_schur_iterate: n_iter iterations, five launches each, no host synchronisation;
 *     a latched stop turns everything enqueued behind it into no-ops.  The state block holds
 *     two buffers: iteration i (first_iter = the number of iterations enqueued since
 *     iamx_ba_schur_factor) reads buffer i & 1 and writes the other; after k iterations the
 *     current scalars are in buffer k & 1.  phase = -1: whole iterations (one rank); several
 *     ranks: phase 0 = the three passes of q = S y over this rank's observations -> qraw DEV
 *     [n_cams][7], all-reduce qraw, then phase 1 = the update (scalars replicated on every
 *     rank).  t DEV [2 n_obs], part DEV [2 n_cams] (partials of the two inner products),
 *     x r z p y DEV [7 n_cams].
 *   iamx_ba_schur_finish: step DEV [n] = (x, Y D_p (gp - W^T x)) in the scaled variables of the
 *     subproblem; the point entries outside [pt_lo, pt_hi) are written as 0 (several ranks:
 *     all-reduce the point part once).
 * ------------------------------------------------------------------------------------ */
int iamx_ba_accumulate(const double *Jc, const double *Jp, const double *r, const int32_t *cam_ptr,
                       const int32_t *pt_ptr, const int32_t *pt_obs, int64_t n_obs, int n_cams,
                       int n_pts, double *U, double *V, double *gc, double *gp, void *stream);
/* out DEV [7 n_cams + 3 n_pts] = (diag U, diag V): the column sums of J.^2 */
int iamx_ba_block_diag(const double *U, const double *V, int n_cams, int n_pts, double *out,
                       void *stream);
int iamx_ba_schur_state_size(void);
/* optimize_calib='global' (scripts/lib/optimizer.py:142-169,181-189: 8 calibration columns that
 * every observation shares; `process.py --cam-calibration`, scripts/process.py:384): pass Jk DEV
 * [n_obs][2][8] (iamx_ba_residual_jac) to all four calls for the BORDERED form -- the calibration
 * block joins the camera side of the reduced system as one more block of 8 parameters, with the
 * inverse of its own normal-equation block (D_k sum Jk^T Jk D_k + Dreg_k^2) as preconditioner.
 * Then n = 7 n_cams + 3 n_pts + 8 (calibration entries of d / dreg / step behind the points),
 * x r z p y qraw DEV [7 n_cams + 8], minv DEV [28 n_cams + 36], part DEV [2 (n_cams + 1)], sraw DEV
 * [35 n_cams + 44] (all-reduce all of it), ckpart DEV [n_cams][44] scratch; several ranks
 * all-reduce qraw [7 n_cams + 8] per iteration.  Jk = NULL: cameras and points only, sizes as
 * documented above, ckpart unused. */
int iamx_ba_schur_prepare(const double *Jc, const double *Jp, const double *Jk, const double *r,
                          const int32_t *cam_ptr, const int32_t *pt_idx, int64_t n_obs, int n_cams,
                          int n_pts, const double *V, const double *gp, const double *d,
                          const double *dreg, double *Y, double *yg, double *zp, double *sraw,
                          double *ckpart, void *stream);
int iamx_ba_schur_factor(const double *sraw, const double *d, const double *dreg, int n_cams,
                         int n_pts, int with_calib, double eta, double qtol, int max_iter,
                         double *minv, double *x, double *r, double *z, double *p, double *y,
                         double *state, void *stream);
int iamx_ba_schur_iterate(const double *Jc, const double *Jp, const double *Jk,
                          const int32_t *cam_idx, const int32_t *pt_idx, const int32_t *cam_ptr,
                          const int32_t *pt_ptr, const int32_t *pt_obs, int64_t n_obs, int n_cams,
                          int n_pts, const double *d, const double *dreg, const double *Y,
                          const double *minv, double *t, double *zp, double *qraw, double *part,
                          double *ckpart, double *x, double *r, double *z, double *p, double *y,
                          double *state, int first_iter, int n_iter, int phase, void *stream);
int iamx_ba_schur_finish(const double *Jc, const double *Jp, const double *Jk,
                         const int32_t *cam_idx, const int32_t *pt_ptr, const int32_t *pt_obs,
                         int64_t n_obs, int n_cams, int n_pts, int pt_lo, int pt_hi, const double *d,
                         const double *Y, const double *yg, const double *x, double *y, double *t,
                         double *step, void *stream);

/* ------------------------------------------------------------------------------------
 * Collectives of the hot path for callers that are not python: RCCL over
 * xGMI, bound at run time (a process that already holds an RCCL -- torch bundles one -- re-uses
 * it).  The python layer does the same two exchanges through torch.distributed.
 *   iamx_comm_unique_id: HOST id128 [128 bytes], created by one rank and handed to the others by
 *     whatever channel the caller has (file, MPI, torch store)
 *   iamx_comm_init: one communicator per process / GPU (the current HIP device); *comm out
 *   iamx_comm_allgather: rank r contributes bytes_per_rank bytes at `send`; recv DEV
 *     [n_ranks * bytes_per_rank]; in place when send == recv + rank * bytes_per_rank -- the packed
 *     descriptor store of the images a rank detected (iamx_desc_pack_* / iamx_desc3_pack_*) in
 *     front of the pair-sharded matching: one large transfer per buffer, never per image
 *   iamx_comm_allreduce_f64: in-place sum -- xr[0..2) and tbuf[0 .. 7 n_cams] between the phases
 *     of iamx_ba_lsmr_phase, the gradient / column norms once per outer iteration
 * All enqueue on `stream`; errors come back as IAMX_ELAUNCH with RCCL's message.
 * ------------------------------------------------------------------------------------ */
int iamx_comm_unique_id(void *id128);
int iamx_comm_init(int n_ranks, int rank, const void *id128, void **comm);
int iamx_comm_destroy(void *comm);
int iamx_comm_allgather(void *comm, const void *send, void *recv, int64_t bytes_per_rank,
                        void *stream);
int iamx_comm_allreduce_f64(void *comm, double *buf, int64_t n, void *stream);

/* float64 vector kernels used by the device LSMR (scipy/sparse/linalg/_isolve/lsmr.py):
 *   axpby: y = a*x + b*y (b == 0 ignores y's old content)
 *   mul2:  out = a.*b (+ c.*d when c != NULL)
 *   dot:   out[0] = sum x.*y, fixed reduction tree (scratch: DEV [256] float64)
 *   lsmr_update: hbar = h + c_hbar*hbar; x += c_x*hbar; h = v + c_h*h */
int iamx_vec_axpby(int64_t n, double a, const double *x, double b, double *y, void *stream);
int iamx_vec_mul2(int64_t n, const double *a, const double *b, const double *c, const double *d,
                  double *out, void *stream);
int iamx_vec_dot(int64_t n, const double *x, const double *y, double *out, double *scratch,
                 void *stream);
int iamx_vec_lsmr_update(int64_t n, double *h, double *hbar, double *x, const double *v,
                         double c_hbar, double c_x, double c_h, void *stream);

/* n-vector kernels of the trust-region-reflective outer loop (csrc/trf_vec.hip), float64, all
 * pointers DEV unless noted.  They restate the O(n) helpers of scipy.optimize.least_squares
 * (method='trf'), which the reference calls at scripts/lib/optimizer.py:352-399:
 *   lincomb: out = a*x + b*y + c*z (y, z may be NULL);  mul: out = s*x.*y (y NULL: s*x)
 *   gather:  out[i] = x[idx[i]] (idx DEV int64 [n]; out must not alias x) -- the private point
 *            order of the device problem <-> the reference's parameter order
 *   sqrt_shift: out = sqrt(x + shift)
 *   dots: out[i] = sum a[i].*b[i] (.*w[i] when w and w[i] are not NULL), 1 <= k <= 8; a, b, w are
 *         HOST arrays of k device pointers; scratch: DEV [iamx_vec_scratch_doubles()] float64
 *   absmax_prod: out[0] = max |x.*y| (y NULL: max |x|), NaN if any product is
 *   trf_cl_scaling:  _lsq/common.py CL_scaling_vector -> v, dv
 *   trf_scale:       trf.py trf_bounds: v[dv != 0] *= scale_inv; d = sqrt(v)/scale_inv;
 *                    diag_h = g.*dv./scale_inv; g_h = d.*g   (v_out may be NULL)
 *   trf_jac_scale:   common.py compute_jac_scale from the column sums of J.^2
 *   trf_step_to_bound: common.py step_size_to_bound -> out[0] = min step
 *   trf_reflect:     its `hits` (equal(steps, min_step) * sign(s)) and r_h = p_h with the hit
 *                    components negated (either output may be NULL)
 *   trf_count_outside: out[0] = number of components of x (+ p) outside [lb, ub] (in_bounds)
 *   trf_strictly_feasible: common.py make_strictly_feasible(x (+ step), lb, ub, rstep=0)
 *   trf_active:      common.py find_active_constraints(x, lb, ub, rtol > 0) as -1 / 0 / +1 */
int iamx_vec_lincomb(int64_t n, double a, const double *x, double b, const double *y, double c,
                     const double *z, double *out, void *stream);
int iamx_vec_mul(int64_t n, double s, const double *x, const double *y, double *out, void *stream);
int iamx_vec_gather(int64_t n, const double *x, const int64_t *idx, double *out, void *stream);
int iamx_vec_sqrt_shift(int64_t n, const double *x, double shift, double *out, void *stream);
int iamx_vec_scratch_doubles(void);
int iamx_vec_dots(int64_t n, int k, const double *const *a, const double *const *b,
                  const double *const *w, double *out, double *scratch, void *stream);
int iamx_vec_absmax_prod(int64_t n, const double *x, const double *y, double *out, double *scratch,
                         void *stream);
int iamx_trf_cl_scaling(int64_t n, const double *x, const double *g, const double *lb,
                        const double *ub, double *v, double *dv, void *stream);
int iamx_trf_scale(int64_t n, const double *v, const double *dv, const double *g,
                   const double *scale_inv, double *v_out, double *d, double *diag_h, double *g_h,
                   void *stream);
int iamx_trf_jac_scale(int64_t n, const double *colsq, double *scale_inv, int first, void *stream);
int iamx_trf_step_to_bound(int64_t n, const double *x, const double *s, const double *lb,
                           const double *ub, double *out, double *scratch, void *stream);
int iamx_trf_reflect(int64_t n, const double *x, const double *s, const double *lb, const double *ub,
                     double min_step, const double *p_h, double *r_h, double *hits, void *stream);
int iamx_trf_count_outside(int64_t n, const double *x, const double *p, const double *lb,
                           const double *ub, double *out, double *scratch, void *stream);
int iamx_trf_strictly_feasible(int64_t n, const double *x, const double *step, const double *lb,
                               const double *ub, double *out, void *stream);
int iamx_trf_active(int64_t n, const double *x, const double *lb, const double *ub, double rtol,
                    double *active, void *stream);
/* make_strictly_feasible(x, lb, ub, rstep) for rstep > 0: the start point of trf_bounds
 * (scipy/optimize/_lsq/trf.py:214 through common.py:440) */
int iamx_trf_feasible_start(int64_t n, const double *x, const double *lb, const double *ub, double rstep,
                            double *out, void *stream);
/* out = x * scale_inv / sqrt(v') with v' = v * scale_inv where dv != 0: the vector whose norm is the
 * first trust-region radius (trf.py:243-246) */
int iamx_trf_scaled_start(int64_t n, const double *x, const double *scale_inv, const double *v,
                          const double *dv, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IAMX_H */
