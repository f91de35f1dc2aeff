/* meld_hip.h -- C-ABI of libmeld_hip.so: the MI355X (gfx950) hot path of MELD.fit_transform.
 *
 * The reference (KrishnaswamyLab/MELD v1.0.2) is pure Python and has no FFI; its hot path is the
 * arithmetic that graphtools and pygsp perform under
 *     meld/meld.py:273   self.fit(X)            -> graphtools.Graph(...)   (kNN + alpha-decay kernel)
 *     meld/meld.py:235   filter.filter(...)     -> meld/filter.py:39,56,59 (lmax + Chebyshev filter)
 * Each entry point below names the reference interface (file:line, or the [UPSTREAM] library
 * routine called from that line) whose work it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream;
 *   - no hidden allocation, no global state besides the last-error string (thread local);
 *   - return value: 0 = ok, <0 = error (MELD_ERR_*), message via meld_last_error();
 *   - N, nnz counts are int64_t; column indices int32 (N < 2^31); values are IEEE fp64 wherever
 *     the reference computes in fp64.  fp32 appears only in the candidate search, whose result
 *     is re-evaluated exactly in fp64 before it is used.
 */
#ifndef MELD_HIP_H
#define MELD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MELD_OK 0
#define MELD_ERR_INVALID (-1)     /* bad argument */
#define MELD_ERR_UNSUPPORTED (-2) /* size outside what the kernels are instantiated for */
#define MELD_ERR_HIP (-3)         /* a HIP runtime call failed */

typedef void* meld_stream_t;

/* ---- library info ------------------------------------------------------------------------- */
int meld_abi_version(void);
const char* meld_last_error(void);
/* hipGetDeviceCount-style probe: number of visible devices, or <0 */
int meld_device_count(void);

/* ---- kNN candidate search (replaces [UPSTREAM graphtools kNNGraph.build_kernel_to_data ->
 *      sklearn NearestNeighbors.kneighbors], reached from meld/meld.py:273) ------------------ */

/* Geometry of the search kernel, so that the caller can size its buffers. */
int meld_knn_padded_dim(int d);     /* KP: augmented (d+2) dimension rounded up to an instantiated size; <0 if d unsupported */
int meld_knn_tile_refs(void);       /* TS: reference points per LDS tile (reference array is padded to a multiple) */
int meld_knn_block_queries(void);   /* BQ: queries per workgroup (query arrays are padded to a multiple) */
int meld_knn_row_capacity(int ksel);/* CAP: row stride of the candidate buffers for a given ksel; <0 if unsupported */
double meld_knn_error_coef(int d);  /* fp32 FMA-chain bound KP * 2^-21 of meld_knn_topk */

/* column sums of X[N,d] (fp64) -> sums[d] (zeroed by the call).  Used for centring. */
int meld_col_sums_f64(const double* X, int64_t N, int d, double* sums, meld_stream_t stream);
/* column sums, minima and maxima of X[N,d] in ONE pass (fixed-order reduction: reproducible bits): the centring mean, the
 * front end's NaN / infinity check (graphtools rejects such input before building, reached from meld/meld.py:273) and the
 * extremes meld_knn16_prepare_scaled derives the operand scale from.  temp: meld_col_stats_temp_bytes(d) bytes. */
size_t meld_col_stats_temp_bytes(int d);
int meld_col_stats_f64(const double* X, int64_t N, int d, double* sums, double* mins, double* maxs, void* temp, size_t temp_bytes,
                       meld_stream_t stream);

/* Build the fp32 operands of the distance GEMM from fp64 data (centred by `mean[d]`):
 *   refs:    tile-major augmented reference form  Rt[n_tiles][KP/2][TS][2]  of  [-2x, 1, |x|^2, 0..]
 *            rows >= N are filled so that their distance is huge; norm2[N] = |x|^2 (fp32),
 *            norm2_max[1] = max_i norm2[i] (atomic max; must be zeroed by the caller);
 *   queries: row-major augmented query form  Q[q_pad][KP]  of  [x, |x|^2, 1, 0..]  for rows
 *            q_begin .. q_begin+q_count (rows beyond q_count replicate the last valid row). */
int meld_knn_prepare_refs(const double* X, int64_t N, int d, const double* mean, int KP, float* Rt,
                          float* norm2, float* norm2_max, meld_stream_t stream);
int meld_knn_prepare_queries(const double* X, int64_t N, int d, const double* mean, int KP,
                             int64_t q_begin, int64_t q_count, float* Q, meld_stream_t stream);

/* Brute-force search on the matrix cores: for each of the q_count queries in Q, the ksel
 * references with the smallest fp32 squared distance (self included).  Output rows have stride
 * CAP = meld_knn_row_capacity(ksel); the first min(cand_cnt, ksel) entries of a row are valid and
 * sorted by (d2, idx) ascending.  Buffers must hold q_pad = roundup(q_count, BQ) rows. */
int meld_knn_topk(const float* Q, const float* Rt, int64_t n_ref, int KP, int64_t q_count, int ksel,
                  int32_t* cand_idx, float* cand_d2, int32_t* cand_cnt, meld_stream_t stream);

/* ---- second-generation search: split-fp16 operands on v_mfma_f32_32x32x16_f16 (knn16.hip).
 * Same role and output contract as meld_knn_topk (ksel smallest approximate squared distances,
 * rows of stride CAP sorted by (d2, idx), d2 in input units); ~4x less matrix-pipe time.
 *   meld_knn16_prepare: centre by mean[d], scale into [-1,1], split hi/lo (d2 = |q|^2 + |r|^2 - 2 q.r:
 *       |r|^2 - 2 q.r runs on the matrix cores -- |r|^2 as three exact fp16 pieces in K slots d .. d+2
 *       against 1.0 on the query side -- and |q|^2 is folded into the selection threshold):
 *       Rt16  : ceil(N / TS) * meld_knn16_tile_bytes(d)   (per tile: [kb][half][plane][ref][8 x fp16]
 *               of [-2 x, |x|^2 pieces])
 *       Q16   : roundup(q_count, BQ) * meld_knn16_query_bytes(d);  Qn : roundup(q_count, BQ) fp32 norms
 *       norm2[N], norm2_max[1] (input units), scale_info[4] floats (s, 1/s^2, absmax, pad)
 *   meld_knn16_error_coef: E / max|x~|^2 to pass to meld_knn_refine for this search. */
int meld_knn16_kblocks(int d);           /* KB = ceil((d+3)/16); <0 if d unsupported (d <= 141) */
int meld_knn16_tile_refs(void);          /* TS */
/* SPLIT operand layout (wherever d > 13 and d + 6 K slots fit the K blocks of d + 3): K block 0 holds the first
 * meld_knn16_split_dims(d) = 13 coordinates and the pieces of |r_A|^2 (less a margin that the second set of pieces,
 * behind the other coordinates, gives back), so that the list-driven first pass (meld_knn16_topk_listed) can test a block of 32
 * references on its accumulators behind K block 0 -- a distance over some of the coordinates never exceeds the distance -- and
 * skip the other K blocks where no partial value is within reach of its row.  Results are the same in any orthonormal frame;
 * the test prunes when the leading coordinates carry the distances (the Python host rotates the cells to principal
 * coordinates for the search; PCA-reduced input, the reference's default, already is such a frame).  0: plain layout.
 * The layout is a function of d alone (no process-wide switch). */
int meld_knn16_split_dims(int d);
/* The cells' principal frame for that search (frame.hip; no reference counterpart -- graphtools searches the data as given):
 *   meld_cov_sample_f64:  cov[d*d] (row-major, upper triangle; zeroed by the caller) += the scatter matrix about mean[d] of
 *                         the rows 0, stride, 2 stride, ... of X[N][d]
 *   meld_rotate_rows_f64: out[N][d] = (X - mean) A^T, At[d][meld_frame_max_dims()] = the new axes as rows, zero-padded; out != X
 * d <= meld_frame_max_dims().  The eigenvectors are the caller's business (a d x d problem: host LAPACK). */
int meld_frame_max_dims(void);
int meld_cov_sample_f64(const double* X, int64_t N, int d, const double* mean, int64_t stride, double* cov, meld_stream_t stream);
int meld_rotate_rows_f64(const double* X, int64_t N, int d, const double* mean, const double* At, double* out, meld_stream_t stream);
int meld_knn16_block_queries(void);      /* BQ */
int meld_knn16_row_capacity(int ksel);   /* CAP */
double meld_knn16_error_coef(int nprod, int d);       /* worst case: E <= coef * max|x~|^2 (depends on the dimension:
                                                        the accumulators sum nprod * d + 3 terms) */
double meld_knn16_error_coef_const(int nprod, int d); /* per-row form: E_i = c_const max|x~|^2 + c_lin |x~_i| max|x~| */
double meld_knn16_error_coef_lin(int nprod);
size_t meld_knn16_tile_bytes(int d);     /* bytes of one reference tile (KB * 4096) */
size_t meld_knn16_query_bytes(int d);    /* bytes of one query row of Q16 */
int meld_knn16_prepare(const double* X, int64_t N, int d, const double* mean, int64_t q_begin,
                       int64_t q_count, void* Rt16, void* Q16, float* Qn, float* norm2, float* norm2_max,
                       float* scale_info, meld_stream_t stream);
/* meld_knn16_prepare with the operand scale (max |x - mean| over all entries) taken from the columns' extremes
 * (meld_col_stats_f64) instead of a pass of its own over X; bit-identical operands. */
int meld_knn16_prepare_scaled(const double* X, int64_t N, int d, const double* mean, const double* col_min, const double* col_max,
                              int64_t q_begin, int64_t q_count, void* Rt16, void* Q16, float* Qn, float* norm2, float* norm2_max,
                              float* scale_info, meld_stream_t stream);
/* Optional exact pruning.  meld_knn16_bounds fills lb2[n_query_waves][n_tiles] (fp16, rounded towards
 * zero) with a lower bound (scaled space) on the squared distance between any query of a wave of the
 * search kernel (64 consecutive cells = one reference tile) and any reference of a tile (TS consecutive
 * cells): (min over the wave's cells of the distance to the tile centroid - tile radius)^2, the minimum
 * taken point by point by a cells x centroids distance GEMM on the MFMA path (Rt16 = the reference
 * operand of meld_knn16_prepare).  In meld_knn16_topk a wave sits out every tile whose bound exceeds all
 * of its thresholds plus the search-error allowance, and a tile no wave of the workgroup needs is not
 * even loaded -- neither can change the result.  It pays when consecutive cells are spatially close
 * (meld_assign_nearest ordering).  lb2 = NULL disables pruning.  q_begin (a multiple of TS) = global
 * index of query 0 (the scan of every workgroup starts at its own position among the references and
 * wraps around).
 * thr_seed (optional, q_count floats, scaled units) = the thr_init the search of the same queries will be
 * started with (meld_knn16_seed_thresholds*), nprod = that search's products: thresholds only fall, so a tile
 * that no query of the wave can reach from its OWN start threshold (|p - c_t| - rho_t > sqrt(seed_p + E) for
 * all 64 cells p) is marked +inf in the table.  The table is then valid only for a search started from
 * exactly these thresholds (or lower ones).
 * No reference counterpart: graphtools delegates the search to sklearn's trees. */
size_t meld_knn16_bounds_bytes(int64_t n_ref, int64_t q_count);
size_t meld_knn16_bounds_temp_bytes(int64_t n_ref, int d, int64_t q_count);
int meld_knn16_bounds(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                      const float* norm2_max, const void* Rt16, int64_t q_begin, int64_t q_count,
                      const float* thr_seed, const float* q_norm2 /* optional: Qn of meld_knn16_prepare, per-row allowance */,
                      int nprod, void* temp, void* lb2, meld_stream_t stream);
/* The same in two calls, for a row-sharded build: every rank computes the spheres of a range of tiles into its slots of a zeroed
 * temp (three arrays, meld_knn16_sphere_layout: [rows][row_bytes] centres, [rows] fp32 |c|^2, [rows] fp32 radii), the ranks
 * all-gather them, and the table is built from the complete arrays. */
int meld_knn16_tile_spheres(const double* X, int64_t N, int d, const double* mean, const float* scale_info, void* temp,
                            int64_t tile_begin, int64_t tile_count, meld_stream_t stream);
int meld_knn16_sphere_layout(int64_t n_ref, int d, int64_t* rows, int64_t* row_bytes);
int meld_knn16_bounds_from_spheres(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                      const float* norm2_max, const void* Rt16, int64_t q_begin, int64_t q_count,
                      const float* thr_seed, const float* q_norm2 /* optional: Qn of meld_knn16_prepare, per-row allowance */,
                      int nprod, void* temp, void* lb2, meld_stream_t stream);
/* nprod selects the precision of the products: 3 = hi.hi + hi.lo + lo.hi (error bound
 * 2^-14 max|x~|^2), 1 = hi.hi only (a third of the MFMAs, bound 2^-9 |x~_q| max|x~|; rows the looser
 * bound cannot certify are searched again / go through meld_knn_radius_exact, so results are
 * identical).  The norm pieces live in the hi plane and are exact in either mode. */
/* meld_knn16_prepare for a search between two point sets (the blocks between two samples of graphtools' MNN kernel,
 * reached through reference meld/meld.py:117-118 with sample_idx=, test/test_meld.py:34): X holds n_total rows, the
 * references are rows [0, n_refs), the queries a range of the rest.  Scaling over all rows; norm2 / norm2_max cover the
 * references only (the caller merges the queries' norms from Qn). */
int meld_knn16_prepare_cross(const double* X, int64_t n_refs, int64_t n_total, int d, const double* mean, int64_t q_begin,
                             int64_t q_count, void* Rt16, void* Q16, float* Qn, float* norm2, float* norm2_max,
                             float* scale_info, meld_stream_t stream);
/* query operands for a list of local rows (rows[i] + q_begin = global index): the re-search of the
 * rows a reduced-precision first pass could not certify */
int meld_knn16_prepare_rows(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                            int64_t q_begin, const int32_t* rows, int64_t n_rows, void* Q16, float* Qn,
                            meld_stream_t stream);
/* Start thresholds (scaled units, thr[roundup(n_rows, BQ)]) for the re-search of those rows through
 * meld_knn16_topk(thr_init = thr): a first pass that found ksel references with approximate d2 <= tau bounds the re-search's
 * ksel-th distance by tau + E1 + E3 (both passes' error allowances); rows with a shorter first list start at +inf.
 * cand_cnt / cand_d2 (stride cap): the first pass's lists; err_coef / err_lin: its coefficients, err3: meld_knn16_error_coef(3, d). */
int meld_knn16_research_thresholds(const int32_t* rows, int64_t n_rows, int64_t q_begin, const int32_t* cand_cnt,
                                   const float* cand_d2, int cap, int ksel, const float* norm2, const float* norm2_max,
                                   double err_coef, double err_lin, double err3, const float* scale_info, float* thr,
                                   meld_stream_t stream);
/* thr_init (optional, roundup(q_count, BQ) floats in the SCALED units of the search, i.e. input
 * d2 * scale_info[0]^2): per query, a bound below which all wanted neighbours are known to lie; the
 * selection thresholds start there instead of at +inf (the re-search passes the first pass's
 * ksel-th distance plus both error allowances).  NULL = +inf.
 * n_slices > 1 (small query sets): the references are cut into n_slices ranges, each scanned by its
 * own workgroups into its own candidate rows (buffers of n_slices * roundup(q_count, BQ) rows);
 * meld_knn16_merge_slices then writes the ksel smallest of the union to the final rows. */
/* Start values for the thresholds of meld_knn16_topk's first pass (thr_init, scaled units, roundup(q_count, BQ)
 * floats): the (knn+1)-th smallest distance of a query within its own block of BQ cells bounds its bandwidth from
 * above, so nothing beyond radius_factor^2 times that (plus the error allowance) can be wanted.  Spares the scan the
 * wholesale appends of its first tiles.  q_begin must be a multiple of BQ.  No reference counterpart. */
int meld_knn16_seed_thresholds(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                               const float* norm2_max, int64_t q_begin, int64_t q_count, int knn, double radius_factor,
                               int nprod, float* thr_init, meld_stream_t stream);
/* The same start values from the fp16 operands of meld_knn16_prepare, computed on the matrix pipe over the query
 * block's own tiles and side_tiles on either side in index order (side_tiles <= 0: n_ref / 12500, between 8 and 64 --
 * the cost grows with N, what tighter seeds save in the search and through meld_knn16_bounds' per-query test with N^2). */
int meld_knn16_seed_thresholds_mfma(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info,
                                    const float* norm2_max, int64_t n_ref, int d, int64_t q_begin, int64_t q_count,
                                    int knn, double radius_factor, int nprod, int side_tiles, float* thr_init,
                                    meld_stream_t stream);
int meld_knn16_topk(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info,
                    int64_t n_ref, int d, int64_t q_count, int ksel, int nprod, int n_slices, const void* lb2,
                    const float* norm2_max, int64_t q_begin, const float* thr_init, int knn, double radius_factor,
                    int32_t* cand_idx, float* cand_d2, int32_t* cand_cnt, float* cand_thr,
                    uint64_t* tiles_done /* += (wave, tile) pairs actually computed, or NULL */,
                    const int32_t* block_order /* workgroup b searches query block block_order[b]; NULL = b */,
                    meld_stream_t stream);
/* Dispatch order of the pruned search.  Workgroups start in index order as slots free up, and with pruning the tiles a
 * query block has to stage differ by an order of magnitude; meld_knn16_block_work writes, per query block, the number of
 * tiles its four waves cannot rule out at their start thresholds (lb2 from meld_knn16_bounds, thr_seed = the thr_init of
 * the search, NULL = +inf).  Passing the blocks by decreasing work as block_order starts the longest first
 * (1M cells: the last slot finishes 4 % after the mean instead of 12 %).  The result does not depend on the order. */
int meld_knn16_block_work(const void* lb2, const float* thr_seed, int64_t n_ref, int d, int64_t q_count, int nprod,
                          const float* norm2_max, const float* scale_info, int32_t* work, meld_stream_t stream);
/* Step lists of the first pass (replaces the per-step use of the pruning table inside the search; graphtools
 * build_kernel_to_data's kNN query reached from /root/reference/meld/meld.py:273 has no counterpart -- this is how the
 * brute-force scan is cut down).  With start thresholds from meld_knn16_seed_thresholds* the tiles a query block can rule out
 * are known before the search (what the falling thresholds add is 1.5 % of the blocks at 1M cells), so they are written
 * down once: list[b * list_stride + i] = tile | (bit w set: wave w of block b cannot rule the tile out) << 24 for the
 * i-th step of block b in scan order, cnt[b] = its steps (>= 1), which is also the block's work for block_order.
 * list_stride >= ceil(n_ref / 64); q_begin as for meld_knn16_topk (it fixes where a block's scan starts).
 * meld_knn16_topk_listed is meld_knn16_topk(nprod = 1) walking those lists: same candidate rows, counts,
 * thresholds (the lists are a superset of the tiles the table-driven kernel visits; the extra ones hold no candidate). */
int meld_knn16_step_lists(const void* lb2, const float* thr_seed, int64_t n_ref, int d, int64_t q_count, int nprod,
                          const float* norm2_max, const float* scale_info, int64_t q_begin, uint32_t* list,
                          int64_t list_stride, int32_t* cnt, meld_stream_t stream);
/* The same lists straight from the cells when the queries are all the cells (one GPU): tile spheres, bounds, symmetrisation and list
 * building in one call, with the bounds kept as two bits per (wave, tile) instead of the fp16 table (lb2 is never written).
 * temp: meld_knn16_bounds_temp_bytes(n_ref, d, n_ref) bytes; scratch: meld_knn16_list_scratch_bytes(n_ref) bytes;
 * thr_seed / q_norm2: the start thresholds and |q|^2 of the search (both required). */
size_t meld_knn16_list_scratch_bytes(int64_t n_ref);
int meld_knn16_step_lists_direct(const double* X, int64_t N, int d, const double* mean, const float* scale_info, const float* norm2_max,
                                 const void* Rt16, const float* thr_seed, const float* q_norm2, int nprod, void* temp, void* scratch,
                                 uint32_t* list, int64_t list_stride, int32_t* cnt, meld_stream_t stream);
/* ... with the bounds taken from the first K block of the operands alone (lead_only != 0; SPLIT layout, cells in a frame whose
 * leading coordinates carry the distances -- see meld_knn16_split_dims): lower bounds all the same, a quarter of the tile stream. */
int meld_knn16_step_lists_direct_lead(const double* X, int64_t N, int d, const double* mean, const float* scale_info, const float* norm2_max,
                                 const void* Rt16, const float* thr_seed, const float* q_norm2, int nprod, void* temp, void* scratch,
                                 uint32_t* list, int64_t list_stride, int32_t* cnt, int lead_only, meld_stream_t stream);
int meld_knn16_topk_listed(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info, int64_t n_ref, int d,
                           int64_t q_count, int ksel, const uint32_t* step_list, const int32_t* step_cnt, int64_t list_stride,
                           const float* norm2_max, int64_t q_begin, const float* thr_init, int knn, double radius_factor,
                           int32_t* cand_idx, float* cand_d2, int32_t* cand_cnt, float* cand_thr, uint64_t* tiles_done,
                           const int32_t* block_order, int n_slices /* as meld_knn16_topk: slice y walks the entries y, y + S, ... of a
                           block's list into its own candidate rows; merge with meld_knn16_merge_slices */, meld_stream_t stream);
/* ... with the partial test of the SPLIT layout: partial_test != 0 lets the pass drop a block of 32 references behind its first K
 * block when no partial value is within reach of its row (same rows, counts and thresholds; see meld_knn16_split_dims).
 * tiles_done (optional) holds TWO counters here: [0] (wave, tile) pairs computed, [1] blocks of 32 references that went on. */
int meld_knn16_topk_listed_partial(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info, int64_t n_ref,
                                   int d, int64_t q_count, int ksel, const uint32_t* step_list, const int32_t* step_cnt,
                                   int64_t list_stride, const float* norm2_max, int64_t q_begin, const float* thr_init, int knn,
                                   double radius_factor, int32_t* cand_idx, float* cand_d2, int32_t* cand_cnt, float* cand_thr,
                                   uint64_t* tiles_done, const int32_t* block_order, int n_slices, int partial_test,
                                   meld_stream_t stream);
/* The partial test as a pass of its own over the step lists (round 6; the default route in the principal frame): every listed
 * (wave, tile) pair is tested on K block 0 alone against the row's START threshold (thr_init, scaled units: the seeds the lists
 * were built for) and loses its bit when no partial distance is within reach; step_list is rewritten in place, emptied entries
 * dropped, cnt_out[block] = the new length (may alias step_cnt), tested (optional) += pairs tested.  meld_knn16_topk_listed over
 * the thinned lists returns the rows the search over the full ones would (no reference counterpart: graphtools searches a tree). */
int meld_knn16_partial_filter(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info, const float* norm2_max, int d,
                              int64_t q_count, const float* thr_init, uint32_t* step_list, const int32_t* step_cnt, int64_t list_stride,
                              int32_t* cnt_out, uint64_t* tested, const int32_t* block_order /* optional: workgroup g takes block
                              block_order[g] */, meld_stream_t stream);
/* Radius cut (cand_thr != NULL; knn and radius_factor = (-ln thresh)^(1/decay) of the kernel that will be
 * built from the lists): once a row holds knn + 1 entries, its bandwidth^2 is at most A + E (A = its
 * (knn+1)-th smallest approximate d2, E = the row's search-error allowance), so nothing with approximate d2
 * above R = radius_factor^2 (A + E) + E can lie inside the kernel radius: the row's threshold drops to R, the
 * list stays short (far fewer appends and compactions, tighter pruning) and cand_thr[q] (input units,
 * roundup(q_count, BQ) floats) publishes the final threshold -- the list holds every reference below it --
 * for meld_knn_refine's completeness test.  The graph is the same with or without the cut. */
/* workgroups of meld_knn16_topk resident on the device at once (occupancy x CUs); the host hands a
 * nearly empty last wave of workgroups to a sliced launch instead of letting it run alone */
int meld_knn16_resident_blocks(int d, int nprod);
int meld_knn16_max_slices(int ksel); /* largest n_slices meld_knn16_merge_slices accepts */
int meld_knn16_merge_slices(const int32_t* s_idx, const float* s_d2, const int32_t* s_cnt, int64_t q_count,
                            int ksel, int n_slices, int32_t* out_idx, float* out_d2, int32_t* out_cnt,
                            meld_stream_t stream);

/* ---- exact re-evaluation + alpha-decay kernel (replaces [UPSTREAM graphtools
 *      kNNGraph.build_kernel_to_data, "affinities" block]) ---------------------------------- */

/* For each query row: exact fp64 distances to its candidates, bandwidth bw = (knn+1)-th smallest
 * (self counted), kernel value v = exp(-(d/bw)^decay), kept iff v >= thresh.  A row whose candidate
 * list cannot be proven to contain every reference with v >= thresh ("completeness")
 * is appended to flag_rows (its keep_cnt is set to 0) and must go through meld_knn_radius_exact.
 *   cand_val[q_count][ksel] : kernel value or 0
 *   keep_cnt[q_count]       : number of kept OFF-DIAGONAL entries (0 for flagged rows)
 *   err_coef, err_coef_lin  : search-error allowance of the kernel that produced the candidates,
 *                             E_i = err_coef * norm2_max + err_coef_lin * sqrt(norm2[i] * norm2_max)
 *                             (meld_knn_error_coef(d) / 0 for meld_knn_topk; meld_knn16_error_coef_const /
 *                             _lin for meld_knn16_topk, or the worst case meld_knn16_error_coef with norm2 = NULL)
 *   n_flag[1]               : atomic counter, zeroed by the caller
 *   rows (optional)         : second-stage form -- candidate row q belongs to local row rows[q]:
 *                             bw / cand_val / keep_cnt are written at that row, the candidate indices
 *                             are copied to cand_idx_out (row stride out_cap) and flag_rows receives
 *                             the local row.  NULL = row q is local row q. */
int meld_knn_refine(const double* X, int64_t N, int d, int64_t q_begin, int64_t q_count,
                    const int32_t* cand_idx, const float* cand_d2, const int32_t* cand_cnt,
                    const float* cand_thr /* [q_count] thresholds published by meld_knn16_topk's radius cut, or NULL */,
                    int ksel, int cap /* row stride of the candidate buffers */, int knn, double decay, double thresh,
                    const float* norm2_max, double err_coef,
                    const float* norm2 /* [N] per-row |x~|^2 or NULL */, double err_coef_lin, double* bw,
                    double* cand_val, int32_t* keep_cnt, int32_t* flag_rows, int32_t* n_flag,
                    const int32_t* rows, int out_cap, int32_t* cand_idx_out,
                    double bw_scale /* graphtools' bandwidth_scale: the kernel uses max(bw * bw_scale, eps); bw[] records the unscaled value */,
                    const double* bw_fixed /* [N] graphtools' bandwidth= (a given bandwidth per cell, knn then only sizes the search), or NULL */,
                    int max_rank /* graphtools' knn_max + 1 (self counted): a row keeps its max_rank nearest cells at most; 0 = no limit */,
                    meld_stream_t stream);

/* Exact fp64 radius search for the flagged rows (the analogue of graphtools' re-search /
 * radius_neighbors fallback), rows x reference chunks over the whole device.  mode 0: fb_cnt[f] = number
 * of off-diagonal references with v >= thresh, and err_flag[0] |= 1 if the row's bandwidth cannot be
 * confirmed (fb_cursor[n_flag] is scratch for that check and is left zeroed);
 * mode 1: write (column, value) at fb_off[f] + running cursor (fb_cursor[f], zero on entry). */
int meld_knn_radius_exact(const double* X, int64_t N, int d, int64_t q_begin, const int32_t* flag_rows,
                          int32_t n_flag, const double* bw, int knn, double decay, double thresh,
                          int mode, int32_t* fb_cnt, const int64_t* fb_off, int32_t* fb_cursor,
                          int32_t* fb_col, double* fb_val, int32_t* err_flag,
                          double bw_scale /* as in meld_knn_refine: bw[] is unscaled; with a fixed bandwidth pass knn = INT32_MAX (nothing to verify) */,
                          meld_stream_t stream);
/* out[i][c] = |X[rows[i]] - X[cand[i][c]]| (rows, cand: global row numbers; cand [n][kk]) in the summation order of meld_knn_refine
 * and meld_knn_radius_exact: a bandwidth ranked from these is one the sweep confirms (it counts the references strictly closer
 * in this arithmetic; a library norm differs by a few ulps at d ~ 50).  The order (csrc/refine.hip): even d <= 256 -- coordinate pair
 * kk in slot kk & 3, an even and an odd FMA chain per slot, d2 = (u0 + u1) + (u2 + u3); any other d -- one even and one odd chain.
 * X must be 16-byte aligned when d is even (rows are read as pairs of doubles). */
int meld_knn_pair_distances(const double* X, int d, const int64_t* rows, const int64_t* cand, int64_t n, int kk, double* out,
                            meld_stream_t stream);

/* ---- symmetrise / anisotropy / Laplacian pieces (replaces [UPSTREAM graphtools
 *      BaseGraph.symmetrize_kernel, apply_anisotropy, PyGSPGraph._build_weight_from_kernel;
 *      pygsp Graph.compute_laplacian]) ------------------------------------------------------- */

/* exclusive prefix sum int32 -> int64 (out has n+1 entries; out[n] = total) */
size_t meld_scan_temp_bytes(int64_t n);
int meld_exclusive_scan_i32_i64(const int32_t* in, int64_t* out, int64_t n, void* temp, size_t temp_bytes,
                                meld_stream_t stream);

/* Emit the directed off-diagonal kernel entries as COO, twice: (i,j,v/2) into slot e and the
 * transposed (j,i,v/2) into slot M + e, keys = (row << 32) | col.  keep_off = exclusive scan of
 * keep_cnt; flagged rows take their entries from the fallback arrays at fb_base + fb_off[f]. */
int meld_coo_emit(int64_t q_begin, int64_t q_count, const int32_t* cand_idx, const double* cand_val,
                  const int32_t* cand_cnt, int ksel, int cap, const int64_t* keep_off, const int32_t* flag_rows,
                  int32_t n_flag, const int64_t* fb_off, const int32_t* fb_col, const double* fb_val,
                  int64_t fb_base, int64_t M, uint64_t* keys, double* vals, meld_stream_t stream);

/* radix sort of (u64 key, f64 value) pairs on bits [0, end_bit) */
size_t meld_sort_temp_bytes(int64_t n);
int meld_sort_pairs_u64_f64(const uint64_t* keys_in, uint64_t* keys_out, const double* vals_in,
                            double* vals_out, int64_t n, int end_bit, void* temp, size_t temp_bytes,
                            meld_stream_t stream);

/* Sum runs of equal keys (K + K^T): unique keys/values and their count. */
size_t meld_merge_temp_bytes(int64_t n);
int meld_coo_merge(const uint64_t* keys_sorted, const double* vals_sorted, int64_t n, uint64_t* ukeys,
                   double* uvals, int64_t* n_unique, void* temp, size_t temp_bytes, meld_stream_t stream);

/* keys of rows [row_begin, row_begin + n_rows) -> CSR: rowptr[n_rows+1] (int64), col[nnz] int32 */
int meld_csr_from_keys(const uint64_t* ukeys, int64_t nnz, int64_t row_begin, int64_t n_rows,
                       int64_t* rowptr, int32_t* col, meld_stream_t stream);

/* The same assembly without a global sort (the default; replaces graphtools' (K + K^T)/2 on scipy.sparse, SURVEY.md
 * section 8a A4, like the calls above).  meld_coo_scatter_rows sends every entry to the bucket of its row -- slots
 * [r * B, (r + 1) * B) of tcol / tval, B = meld_csr_bucket_slots(), n_rows * B entries each; cursor[n_rows] (zeroed
 * inside) ends as the number of entries sent to each row, whether they fitted or not.  meld_csr_rows_sort_merge sorts
 * every bucket by column inside one wave and sums pairs of equal columns in place (ucnt[r] = distinct columns of row r;
 * flags[0] bit 0: a row with more than B entries, bit 1: a column that occurs more than twice -- in either case the
 * caller must use the sort-based calls above, whose summation order is defined), and after an exclusive scan of ucnt
 * (rowptr) meld_csr_compact_rows copies the merged buckets to their final place.  Entries whose row lies outside
 * [row_begin, row_begin + n_rows) are ignored. */
/* Row-sharded build: the transposed entries this rank owes to the other ranks (owner of key k = (k >> 32) / rows_per_rank,
 * clipped to world - 1; entries of self_rank are left out: the caller hands its whole transposed array to
 * meld_coo_scatter_rows, which ignores foreign rows) laid out for ONE equal-split all-to-all with no host round trip:
 * send[world][2][cap] int64 -- [o][0][slot] keys, [o][1][slot] the bits of the values; unused slots hold the key ~0, a row
 * outside every slice.  counts[world] (int32) ends as the number of entries owed to each rank, fitted or not: a count above
 * cap means the caller must fall back to a variable-length exchange. */
int meld_coo_partition_remote(const uint64_t* keys, const double* vals, int64_t n, int64_t rows_per_rank, int world,
                              int self_rank, int64_t cap, int32_t* counts, int64_t* send, meld_stream_t stream);
/* Single-GPU shortcut for meld_coo_emit + meld_coo_scatter_rows (all N rows local, q_begin = 0): the kept candidates of
 * meld_knn_refine go straight into the row buckets -- a row's own entries into its first slots without an atomic, the
 * transposed copies behind one -- instead of through 2 M (key, value) pairs.  cursor[N] holds on entry the number of own
 * entries of every row (keep_cnt; for the rows of the exact sweep its count) and ends as meld_coo_scatter_rows leaves it;
 * ksel <= meld_csr_bucket_slots(). */
int meld_coo_emit_scatter(int64_t q_count, const int32_t* cand_idx, const double* cand_val, int ksel, int cap,
                          const int32_t* keep_cnt, const int32_t* flag_rows, int32_t n_flag, const int64_t* fb_off,
                          const int32_t* fb_col, const double* fb_val, int64_t fb_total, int32_t* cursor, int32_t* tcol,
                          double* tval, meld_stream_t stream);
int meld_csr_bucket_slots(void);
int meld_coo_scatter_rows(const uint64_t* keys, const double* vals, int64_t n, int64_t row_begin, int64_t n_rows,
                          int32_t* cursor, int32_t* tcol, double* tval, meld_stream_t stream);
int meld_csr_rows_sort_merge(const int32_t* cursor, int64_t n_rows, int32_t* tcol, double* tval, int32_t* ucnt,
                             int32_t* flags,
                             int symm /* how K and K^T combine [UPSTREAM graphtools kernel_symm]: 0 "+" (K + K^T) / 2, 1 "*" K o K^T, 2 "mnn" */,
                             double theta /* symm 2: theta min(K, K^T) + (1 - theta) max(K, K^T) */, meld_stream_t stream);
int meld_csr_compact_rows(const int64_t* rowptr, int64_t n_rows, const int32_t* tcol, const double* tval,
                          int32_t* col, double* val, meld_stream_t stream);
/* The same, and sums[r] = diag + the row's sum on the way out: bit for bit what meld_csr_row_sums returns for the compacted rows. */
int meld_csr_compact_rows_sums(const int64_t* rowptr, int64_t n_rows, const int32_t* tcol, const double* tval, int32_t* col,
                               double* val, double diag, double* sums, meld_stream_t stream);

/* ksum[i] = diag + sum_j val[i,j]   (row sums of the symmetrised kernel; diag = K_ii = 1) */
int meld_csr_row_sums(const int64_t* rowptr, const double* val, int64_t n_rows, double diag,
                      double* out, meld_stream_t stream);
/* W_ij = K_ij / (ksum_i * ksum_j)^anisotropy, in place; ksum_all is indexed by global column,
 * ksum_row_offset = global index of local row 0. */
int meld_csr_anisotropy(const int64_t* rowptr, const int32_t* col, double* val, int64_t n_rows,
                        const double* ksum_all, int64_t ksum_row_offset, double anisotropy,
                        meld_stream_t stream);
/* The same, and the degrees dw[r] = sum_j W_rj of the result in the same pass: equal, bit for bit, to meld_csr_row_sums(diag = 0)
 * called afterwards (same lanes, same order). */
int meld_csr_anisotropy_degrees(const int64_t* rowptr, const int32_t* col, double* val, int64_t n_rows,
                                const double* ksum_all, int64_t ksum_row_offset, double anisotropy, double* dw,
                                meld_stream_t stream);

/* ---- Laplacian operator: lmax and the Chebyshev recurrence (replaces [UPSTREAM pygsp
 *      Graph.estimate_lmax] at meld/filter.py:39 and [UPSTREAM pygsp
 *      filters.approximations.cheby_op] at meld/filter.py:59) ------------------------------- */

/* One step of the three-term recurrence on local rows [0, n_rows) of L = diag(dw) - W:
 *     y      = alpha * (dw .* x_loc - W x_full) + beta * x_loc + gamma * z
 *     r     += coef * y                      (if r != NULL)
 *     dots   = per-slot partial sums of [ <y, x_loc>, <y, y> ]   (if dots != NULL; p == 1 only;
 *              dots has 2 * meld_spmm_dot_slots() entries: slot-major per quantity; the caller
 *              sums the slots.  Used by the Lanczos lmax estimate.)
 * x_full is the full-length gathered vector ([n_cols, p] row-major), x_loc = x_full + x_row_offset*p,
 * z, y, r are local ([n_rows, p]); y may alias z.  z may be NULL when gamma == 0.
 * nnz_hint (total nonzeros of the local rows, or 0) only sizes the LDS staging area.
 *   T1 = (L s - a2 s)/a1          : alpha = 1/a1, beta = -a2/a1, gamma = 0
 *   Tk = (2/a1)(L - a2) T - Told  : alpha = 2/a1, beta = -2 a2/a1, gamma = -1  */
int meld_spmm_dot_slots(void);
int meld_cheby_step(const int64_t* rowptr, const int32_t* col, const double* val, const double* dw,
                    int64_t n_rows, int64_t nnz_hint, int p, const double* x_full, int64_t x_row_offset,
                    const double* z, double* y, double* r, double alpha, double beta, double gamma,
                    double coef, double* dots, meld_stream_t stream);

/* One step of the same recurrence for a WIDE signal (the probe block of the filter-bank VertexFrequencyCluster, stands in for the
 * dense window products of /root/reference/meld/cluster.py:98-156,179-194): 1 <= p <= 64 columns, row-major [rows, p], lanes =
 * columns -- the matrix is streamed once for all columns instead of once per column pair.  No accumulator, no dot products:
 *   y = alpha (dw .* x - W x) + beta x + gamma z     (y and z may alias) */
int meld_cheby_step_wide(const int64_t* rowptr, const int32_t* col, const double* val, const double* dw, int64_t n_rows, int p,
                         const double* x_full, int64_t x_row_offset, const double* z, double* y, double alpha, double beta,
                         double gamma, meld_stream_t stream);

/* r = a * x  (n doubles) -- initialises r = c0/2 * T0 */
/* Device-resident Lanczos iterations [it_begin, it_begin + n_iter) of L = diag(dw) - W (single GPU:
 * all n_rows rows local), for the lmax estimate ([UPSTREAM pygsp Graph.estimate_lmax], reference
 * meld/filter.py:39).  v0/v1/v2 [n_rows]: the three rotating vectors -- before iteration 0, v1 holds the
 * (un-normalised) start vector and v0 zeros; state[8]: state[0] = state[3] = 1 / |start|, the rest zero; scratch:
 * 3 * meld_spmm_dot_slots() doubles, zero before iteration 0.  alphas[it] / betas[it] receive the
 * tridiagonal entries; nothing is synchronised -- read them back when a convergence check is due. */
int meld_lanczos_steps(const int64_t* rowptr, const int32_t* col, const double* val, const double* dw,
                       int64_t n_rows, int64_t nnz_hint, double* v0, double* v1, double* v2, double* state,
                       double* alphas, double* betas, int it_begin, int n_iter, double* scratch,
                       meld_stream_t stream);
/* The same iteration as four stream-ordered phases, for the row-sharded driver (it all-reduces dots after
 * the SpMV and nrm2 after the axpy, and all-gathers the new vector): x_full is the gathered iterate,
 * x_row_offset the first local row in it; z_local / y_local / x_local are local rows; state / dots
 * (2 * slots) / nrm2 (slots) as in meld_lanczos_steps. */
int meld_lanczos_spmv(const int64_t* rowptr, const int32_t* col, const double* val, const double* dw, int64_t n_rows,
                      int64_t nnz_hint, const double* x_full, int64_t x_row_offset, const double* z_local,
                      double* y_local, const double* state, double* dots, meld_stream_t stream);
int meld_lanczos_alpha(double* state, const double* dots, double* nrm2, double* alphas, int it, meld_stream_t stream);
int meld_lanczos_axpy(const double* x_local, double* y_local, int64_t n_rows, const double* state, double* nrm2,
                      meld_stream_t stream);
int meld_lanczos_beta(double* state, const double* nrm2, double* dots, double* betas, int it, meld_stream_t stream);
/* One-reduction form of the sharded iteration (one all-reduce + one all-gather per iteration instead of two + one):
 * the iterate stays un-normalised, u_{k+1} = w_k, so meld_lanczos_spmv (with state[3] = 1, state[4] = 0) computes
 * z = L u_k without a scalar and adds the partial sums of <z, u_k> to acc[0 .. slots); meld_lanczos_axpy3 of the previous
 * iteration has added |u_k|^2 to acc[2 slots .. 3 slots).  The caller all-reduces acc (3 * slots doubles, ONCE), then
 *   meld_lanczos_fold:  alpha_k = <z, u_k> / |u_k|^2 -> alphas[it]; n_k = |u_k| -> betas[it - 1] (it > 0); coefficients
 *                       of the update into state[5..7], n_k into state[2]; acc zeroed;
 *   meld_lanczos_axpy3: y <- y / n_k - (alpha_k / n_k) u - (n_k / n_{k-1}) u_prev  (= u_{k+1}, local rows, in place of z)
 *                       and acc[2 slots ..) += |u_{k+1}|^2 (pass nrm2 = acc + 2 * slots).
 * Before iteration 0: acc[2 slots ..) holds the partial sums of |u_0|^2, the rest of acc and state[2] are zero.
 * beta_{k} (= n_{k+1}) is written one iteration late. */
int meld_lanczos_fold(double* state, double* acc, double* alphas, double* betas, int it, meld_stream_t stream);
int meld_lanczos_axpy3(double* y_local, const double* u_local, const double* u_prev_local, int64_t n_rows,
                       const double* state, double* nrm2, meld_stream_t stream);
int meld_scale_f64(const double* x, double a, double* r, int64_t n, meld_stream_t stream);
/* y = a * x + b * y  (n doubles) -- Lanczos vector update.  If nrm2 != NULL it receives
 * meld_spmm_dot_slots() partial sums of <y, y> (zeroed by the call; the caller adds them up). */
int meld_axpby_f64(double a, const double* x, double b, double* y, int64_t n, double* nrm2,
                   meld_stream_t stream);

/* ---- panel-tiled, symmetry-folded layout of W for the recurrence (csrc/spmm_tiled.hip) ----------
 * Same operator and same call sites as meld_cheby_step / meld_lanczos_* ([UPSTREAM pygsp cheby_op /
 * estimate_lmax] at reference meld/filter.py:59 / :39), on a copy of W laid out so that the iterate is
 * staged in LDS instead of being gathered per nonzero: rows in nb nnz-balanced blocks (one per CU); the
 * nonzeros whose column lies inside the block's own row range are stored once per symmetric pair (IN
 * part: 12 instead of 24 bytes per pair); the distinct columns outside it are listed, cut into tiles of
 * tile_cols columns and staged through a ring of LDS buffers (OUT part); every consumer wave streams its
 * own contiguous slice of the block in whole chunks of 64 entries.  Built once per graph.  The fold needs
 * a bitwise symmetric W: the builder verifies it (status 5) and can be told not to fold.  Results are
 * reproducible to rounding, not bit for bit (several waves add into one LDS accumulator).
 * All arrays are device memory owned by the caller:
 *   blk_row   [nb + 1]  int32   first row of every block
 *   blk_ntile [nb]      int32   tiles of the block (-1: the block could not be laid out, see status)
 *   blk_ndist [nb]      int32   distinct OUT columns of the block
 *   seg       [meld_pt_seg_len(nb)] int32  per (block, wave) chunk schedule and segment offsets
 *   list_cols [nnz]     int32   block b's sorted distinct OUT columns start at rowptr[blk_row[b]]
 *   pval      [meld_pt_stream_len(nnz, nb)] fp64  values in stream order (+ optional pval32, fp32, same length)
 *   pidx      [meld_pt_stream_len(nnz, nb)] uint32  LDS offsets of the entry's column slot and row   */
typedef struct meld_pt_layout {
  const int32_t* blk_row;
  const int32_t* blk_ntile;
  const int32_t* blk_ndist;
  const int32_t* seg;
  const int32_t* list_cols;
  const double* pval;
  const uint32_t* pidx;
  int32_t nb;
  const float* pval32; /* optional: pval rounded to fp32, streamed by the Lanczos SpMV of the lmax estimate
                          (meld_pt_lanczos_*) instead of pval; NULL = not kept */
  int64_t stream_len;  /* entries of pval / pidx / pval32 (= meld_pt_stream_len(nnz, nb)) */
  const uint16_t* cdesc; /* [meld_pt_desc_len(nb)] chunk descriptors of every consumer wave's stream */
} meld_pt_layout_t;
int meld_pt_geometry(int* consumer_waves, int* rows_max, int* tile_cols, int* tiles_max);
int meld_pt_num_blocks(int64_t n_rows); /* nb the builder wants for n_rows local rows */
int64_t meld_pt_seg_len(int nb);
int64_t meld_pt_stream_len(int64_t nnz, int nb);
int64_t meld_pt_desc_len(int nb);
/* timing-only ablations of the step kernel (tools/spmm_compare.py); results are wrong while mask != 0 */
int meld_pt_debug_ablate(int mask);
/* development: per-wave wall-clock stamps of the following step launches into buf[nb][16][8] (NULL: off) */
int meld_pt_debug_stamps(unsigned long long* buf);
/* Build the layout of the local rows [0, n_rows) of a CSR matrix with n_cols columns (the arrays of
 * `layout` are written); local row r is column col_base + r of the matrix (0 on one GPU, the shard's first
 * row on a row shard); symmetric != 0 folds the in-block pairs.  status[0] (device) receives 0, or the
 * reason the layout cannot be used:
 * 1 = (no longer raised: a block's column panels are handled in groups), 2 = too many distinct columns in a block,
 * 3 = n_cols beyond the builder's index range, 4 = a segment / a wave's pairs / its padding beyond the
 * builder's ranges, 5 = W is not bitwise symmetric inside a block (build again with symmetric = 0),
 * 6 = a diagonal entry -- the caller then stays on meld_cheby_step. */
int meld_pt_build(const int64_t* rowptr, const int32_t* col, const double* val, int64_t n_rows, int64_t n_cols,
                  int64_t col_base, int symmetric, const meld_pt_layout_t* layout,
                  uint32_t* codes /* scratch, nnz entries */, int32_t* status, meld_stream_t stream);
/* meld_cheby_step on the layout (p = 1, 2 or any p as passes of 2 + 1 columns; dots as there). */
int meld_pt_cheby_step(const meld_pt_layout_t* layout, const int64_t* rowptr, const double* dw, int64_t n_rows, int p,
                       const double* x_full, int64_t x_row_offset, const double* z, double* y, double* r,
                       double alpha, double beta, double gamma, double coef, double* dots, meld_stream_t stream);
/* Steps 2 .. n_coef - 1 of the Chebyshev recurrence in one call (single GPU, no collective between the steps):
 *   T_k = alpha2 L T_{k-1} + beta2 T_{k-1} - T_{k-2},   r += coeffs[k] T_k      [UPSTREAM pygsp cheby_op, /root/reference/meld/filter.py:59]
 * t_prev2 / t_prev1: [n_rows, p] buffers holding T_0 / T_1 on entry, used as ping-pong buffers; r: holds c_0/2 T_0 + c_1 T_1 on
 * entry, the filtered signal on return; coeffs: n_coef doubles on the HOST.  The accumulator is read and written every other step
 * only (a step adds c_k T_k + c_{k-1} T_{k-1} at once), which removes 16 of the 80 vector bytes per row and step at p = 2.
 * *last (optional) = 1 if t_prev1 holds the last T, 0 if t_prev2 does. */
int meld_pt_cheby_run(const meld_pt_layout_t* layout, const int64_t* rowptr, const double* dw, int64_t n_rows, int p,
                      double* t_prev2, double* t_prev1, double* r, const double* coeffs, int n_coef, double alpha2, double beta2,
                      int* last, meld_stream_t stream);
/* meld_lanczos_steps / meld_lanczos_spmv on the layout.  Same contracts, except that meld_pt_lanczos_steps needs
 * scratch = 8 * meld_spmm_dot_slots() doubles (it keeps its partial sums and scalars in parity buffers there: two launches per
 * iteration, the SpMV derives its own scalars; only state[0] = 1 / |start| is read, before iteration 0) and that betas[it] of
 * the LAST iteration of a call is written by a closing one-wave launch.  stop (optional, device memory): a launch of the call that
 * finds *stop != 0 when it starts does nothing -- the caller checks convergence on the host while the NEXT batch already runs
 * and voids what is left of it once it has its answer (the vectors and the entries of that batch are then undefined). */
int meld_pt_lanczos_steps(const meld_pt_layout_t* layout, const int64_t* rowptr, const double* dw, int64_t n_rows,
                          double* v0, double* v1, double* v2, double* state, double* alphas, double* betas,
                          int it_begin, int n_iter, double* scratch, const int32_t* stop, meld_stream_t stream);
int meld_pt_lanczos_spmv(const meld_pt_layout_t* layout, const int64_t* rowptr, const double* dw, int64_t n_rows,
                         const double* x_full, int64_t x_row_offset, const double* z_local, double* y_local,
                         const double* state, double* dots, meld_stream_t stream);

/* ---- row-sharded recurrences with the host out of the loop (csrc/sharded.hip) ----------
 * One process per GPU, cells row-sharded: rank g owns rows [g * rows_pad, (g + 1) * rows_pad) of every full-length vector.
 * RCCL is reached through dlopen("librccl.so.1") (no link-time dependency: the library loads without it and
 * meld_rccl_available() says 0); the communicator is the library's own, built from a unique id the ranks share.
 *   meld_rccl_unique_id: 128 bytes on the HOST (rank 0 calls it, the bytes travel by any means, e.g. torch.distributed);
 *   meld_rccl_comm_create: collective over the `world` ranks, the calling thread's current device = the rank's GPU;
 *   meld_rccl_all_gather / _all_reduce_sum_f64: the two collectives of the path, on `stream`. */
int meld_rccl_available(void);
int meld_rccl_unique_id(void* id_host);
int meld_rccl_comm_create(const void* id_host, int world, int rank, void** comm);
int meld_rccl_comm_destroy(void* comm);
int meld_rccl_all_gather(void* comm, const void* send, void* recv, size_t bytes_per_rank, meld_stream_t stream);
int meld_rccl_all_reduce_sum_f64(void* comm, double* buf, size_t count, meld_stream_t stream);
/* Steps 2 .. n_coef - 1 of the Chebyshev recurrence on a row shard in ONE call (meld_pt_cheby_run's contract with a row
 * offset and an all-gather behind every step): t_a / t_b are the FULL-length iterates [world * rows_pad, p] holding the
 * gathered T_0 / T_1, r [rows_pad, p] the local rows of the result.  layout: the shard's panel-tiled layout, or NULL for
 * the CSR-stream kernel (col / val read only then).  coeffs on the HOST.  Replaces one kernel call + one
 * torch.distributed all-gather per step issued from Python. */
int meld_cheby_run_sharded(void* comm, const meld_pt_layout_t* layout, const int64_t* rowptr, const int32_t* col, const double* val,
                           const double* dw, int64_t n_rows, int64_t nnz, int64_t rows_pad, int64_t row_begin, int p, double* t_a,
                           double* t_b, double* r, const double* coeffs, int n_coef, double alpha2, double beta2, int* last,
                           meld_stream_t stream);
/* Iterations [it_begin, it_begin + n_iter) of the one-reduction Lanczos recurrence (meld_lanczos_spmv with state[3] = 1,
 * state[4] = 0; ONE all-reduce of acc [3 * meld_spmm_dot_slots()]; meld_lanczos_fold; meld_lanczos_axpy3; all-gather of
 * the new vector) on a row shard in one call.  v0 / v1 / v2: FULL-length rotating vectors [world * rows_pad]. */
int meld_lanczos_steps_sharded(void* comm, const meld_pt_layout_t* layout, const int64_t* rowptr, const int32_t* col,
                               const double* val, const double* dw, int64_t n_rows, int64_t nnz, int64_t rows_pad, int64_t row_begin,
                               double* v0, double* v1, double* v2, double* state, double* acc, double* alphas, double* betas,
                               int it_begin, int n_iter, meld_stream_t stream);

/* ---- KMeans step of VertexFrequencyCluster.predict (reference meld/cluster.py:315-345 -> [UPSTREAM
 *      sklearn.cluster.KMeans], Lloyd iteration; csrc/kmeans.hip) ---------------------------------------
 * One pass over X[n, d] (fp64, d <= 32): labels[i] = nearest of the k <= 64 centroids (ties: lowest index),
 * and per workgroup b < n_blocks the partial sums part_sum[b][k][d], counts part_cnt[b][k] (as fp64) and
 * inertia part_inertia[b]; the caller reduces the partials over b (fixed order) and divides. */
int meld_kmeans_max_blocks(void);
int meld_kmeans_assign(const double* X, int64_t n, int d, const double* centroids, int k, int32_t* labels,
                       double* part_sum, double* part_cnt, double* part_inertia, int n_blocks, meld_stream_t stream);

/* ---- cache-locality ordering helper (no reference counterpart; csrc/reorder.hip) ---------- */
/* out[i] = index (within its group) of the centroid nearest to X[i]; cents holds n_per_group
 * centroids per group, group[i] selects the group of point i (NULL = one shared set).  order
 * (optional, grouped form only) = the points sorted by group: the traversal order that keeps a
 * wave inside one group. */
int meld_assign_nearest(const double* X, int64_t N, int d, const double* cents, int n_per_group,
                        const int32_t* group, const int64_t* order, int32_t* out, meld_stream_t stream);
/* Greedy nearest-neighbour chain over the m (<= 64) rows of every group P[g] ([n_groups][m][d] fp64),
 * starting at the row with the smallest first coordinate: rank[g][i] = position of row i along the chain
 * (orders the centroids of the locality permutation so that consecutive groups are close in space). */
int meld_chain_order(const double* P, int64_t n_groups, int m, int d, int32_t* rank, meld_stream_t stream);
/* out[i][:] = X[perm[i]][:] (rows of d doubles; out must not alias X): the cells brought into the device order. */
int meld_gather_rows_f64(const double* X, const int64_t* perm, int64_t N, int d, double* out, meld_stream_t stream);
/* Glue of the ordering levels, one launch each (meld_amd/reorder.py): stable argsort of 32-bit keys below 2^end_bit
 * (order[i] = index of the i-th smallest key; keys_sorted alongside); starts[g] = first position of key g among the sorted
 * keys, g = 0 .. n_groups; cents[(g f + c) d ..] = the cell at fraction (c + 1/2) / f of group g's sorted members;
 * key[i] <- key[i] f + rank[key[i] f + child[i]] (position of the cell's child along its group's chain). */
size_t meld_argsort_u32_temp_bytes(int64_t n);
int meld_argsort_u32(const uint32_t* keys, int64_t n, int end_bit, int64_t* order, uint32_t* keys_sorted, void* temp,
                     size_t temp_bytes, meld_stream_t stream);
int meld_order_starts(const uint32_t* keys_sorted, int64_t n, int n_groups, int64_t* starts, meld_stream_t stream);
int meld_order_pick_centroids(const double* X, int64_t N, int d, const int64_t* order, const int64_t* starts, int n_groups, int f,
                              double* cents, meld_stream_t stream);
int meld_order_update_keys(uint32_t* key, const int32_t* child, const int32_t* rank, int64_t n, int f, meld_stream_t stream);


/* ---- sample labels -> codes, counts and the indicator signal (replaces MELD._create_sample_indicators,
 *      meld/meld.py:143-191: np.unique + LabelBinarizer, and the column normalisation of meld/meld.py:229-232;
 *      csrc/labels.hip) ------------------------------------------------------------------------------------
 * Labels arrive as fixed-width words (numpy 'U' / 'S' / 64-bit integer arrays viewed as int32 [n_rows, n_words],
 * n_words <= meld_factorize_max_words()).  Two labels get the same group iff all their words are equal (a dictionary
 * of whole labels in LDS: no hashing).  meld_factorize_labels leaves in head (int64 [2 + 2 G], G =
 * meld_factorize_max_groups()): [0] status (1: more than G distinct labels, nothing else valid), [1] number of groups,
 * [2 .. 2 + G) the first row of every group, [2 + G ..) the group sizes; the per-row group numbers stay in temp.
 * meld_factorize_codes then writes codes[i] = rank[group of row i] (rank: int32 on the device, the position of every
 * group among the SORTED labels as np.unique orders them -- only the host can compare label strings; NULL: the group
 * numbers themselves). */
int meld_factorize_max_groups(void);
int meld_factorize_max_words(void);
size_t meld_factorize_temp_bytes(int64_t n_rows);
int meld_factorize_labels(const int32_t* words, int64_t n_rows, int n_words, void* temp, size_t temp_bytes, int64_t* head,
                          meld_stream_t stream);
int meld_factorize_codes(const void* temp, int64_t n_rows, const int32_t* rank, int64_t* codes, meld_stream_t stream);
/* out[n_pad, p] (fp64): row i = scale[c] at column c = codes[perm[i]], 0 elsewhere (scale NULL: 1; perm NULL: i);
 * rows n_rows .. n_pad are zero.  The filter's input signal in the device's row order, one pass. */
int meld_indicator_signal(const int64_t* codes, const double* scale, const int64_t* perm, int64_t n_rows, int64_t n_pad, int p,
                          double* out, meld_stream_t stream);
/* out[perm[i]][:] = in[i][:] for i < n_rows (rows of p doubles): the densities back in the caller's cell order
 * (meld/meld.py:246-250 wraps them with the labels' index). */
int meld_scatter_rows_f64(const double* in, const int64_t* perm, int64_t n_rows, int p, double* out, meld_stream_t stream);

/* ---- next#1: normalize_densities (meld/utils.py:35-47) ------------------------------------ */
/* out[i,:] = in[i,:] / sum_j |in[i,j]|  (rows of zeros are copied unchanged, as sklearn does) */
int meld_normalize_rows_l1(const double* in, double* out, int64_t n_rows, int p, meld_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MELD_HIP_H */
