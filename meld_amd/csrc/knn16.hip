// knn16.hip -- candidate search, second generation: split-fp16 distance GEMM on
// v_mfma_f32_32x32x16_f16.
//
// Same contract as knn.hip (ksel smallest approximate squared distances per query, every
// candidate re-evaluated in fp64 afterwards); what changes is how the N x N distance matrix is
// produced.  gfx950 has no fast fp32 matrix path (f32 MFMA = 157 TF, 1/16 of the f16 rate) and
// no xf32, so each fp32 operand v is split into two fp16 numbers  v = hi + lo  (|v - hi - lo| <=
// 2^-22 |v| after scaling the data into [-1, 1]) and the product is evaluated as
//     a.b  ~=  a_hi.b_hi + a_hi.b_lo + a_lo.b_hi          (the dropped a_lo.b_lo is O(2^-22))
// i.e. 3 f16 MFMAs per 16-deep K block, accumulated in fp32: 12 MFMAs x 32 cycles per 32x32
// block of distances at d = 50 (K = 64) against 26 x 64 cycles on the f32 MFMA -- 4.3x less
// matrix-pipe time for an error of a few 1e-6 |x|^2, which refine.hip's completeness test
// budgets for (err_coef below).
//
// Decomposition: workgroup = 4 waves = 256 queries (each wave: 2 groups of 32 queries held as
// B fragments, 64 VGPRs); the waves share the reference tile stream (64 refs; the hi-only first pass
// keeps just the hi planes, 8 KiB at d = 50), copied global -> LDS by LDS DMA into two alternating
// buffers.  One A fragment (ds_read_b128) feeds two MFMA chains (the two query groups), so the matrix
// pipe never waits on a dependent accumulator.  Three workgroups are resident per CU (165 VGPRs,
// 27 KiB LDS): a selection slow path in one workgroup stalls its four waves at the tile barrier, and
// the other workgroups keep the matrix pipe fed (measured: 8-wave workgroups, one per CU, left the
// pipe 65 % idle).
//
// Exact tile pruning (meld_knn16_bounds): every wave owns a row of lower bounds (its 64 queries against
// every reference tile); a wave sits out the tiles whose bound exceeds all of its thresholds, and a tile no
// wave of the workgroup needs is not staged at all.  With the cells in locality order 65 % of the
// 64 x 64 distance blocks are never computed; the candidate lists are bit-identical.
//
// Selection: as in knn.hip (threshold per query, append to the row buffer, compact when full),
// but compaction finds the new threshold by a 32-step radix select on the ordered float bits and
// squeezes the survivors with ballots (~5x cheaper than ranking); rows are ranked once, at the end.
// Radius cut: once a row holds knn + 1 entries its threshold drops to the kernel radius that entry
// implies (knn16_squeeze_row), and the final threshold is published for refine's completeness test.
#include "common.hpp"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>

namespace meld {

constexpr int K16_TS = 64;         // references per LDS tile
#ifndef K16_WG_WAVES
#define K16_WG_WAVES 4
#endif
constexpr int K16_BQ = 64 * K16_WG_WAVES;        // queries per workgroup
constexpr int K16_THREADS = 64 * K16_WG_WAVES;   // waves of 64 queries each; 2-3 workgroups resident per CU
constexpr int K16_NWAVE = K16_THREADS / 64;
constexpr int K16_SLACK = 128;     // CAP = ksel + slack
#ifndef K16_A_AHEAD
#define K16_A_AHEAD 4  // A fragments of a pipeline segment requested ahead of its first MFMA (hi-only search; measured 2 -> 4: -2 %)
#endif
constexpr int K16_CAPMAX = 256;
// Centroids per workgroup of the pruning-table kernel (64 per wave).  Every workgroup streams all cells of its query
// slice past its centroids, so the tile traffic is N x d x 2 B x (tiles / centroids per workgroup): 7.9 GB at 1M cells
// with 256 centroids (the kernel was bound by it, not by its MFMAs or the per-distance VALU work), 2 GB with 1024.
constexpr int K16_BOUNDS_THREADS = 1024;  // (512 from 5 K blocks on: the centroid fragments of a wave no longer fit 128 registers)
constexpr int K16_SLOTS = K16_CAPMAX / 64;  // row entries per lane in the compaction routines
constexpr int K16_DMAX = 16 * 9 - 3;  // largest d (KB = 9)

// K-slot layout of the operands.  PLAIN (dA = 0): coordinates 0 .. d-1, then the three fp16 pieces of |r|^2 (1.0 on the query
// side).  SPLIT (dA = 13, whenever d > 13 and d + 6 fits the K blocks d + 3 needs): K block 0 = coordinates 0 .. 12 and three
// pieces of N_A = |r_A|^2 - m_r; from slot 16 on the other coordinates and three pieces of N_B = |r_B|^2 + m_r.  Every kernel
// that sums all K blocks sees |r|^2 - 2 q.r as before; the list-driven first pass (EE) looks at its accumulators after K block
// 0 -- a distance over a subset of the coordinates never exceeds the distance -- and drops the other K blocks of a block of
// references in which no partial value is below its row's threshold (+ |q_B|^2: the most the other blocks can take away, m_r
// covering the fp16 rounding of r_B: see prepare16_kernel).  It pays when the leading coordinates carry the distances (principal
// coordinates: the host rotates the cells for the search, meld_amd/graph.py; any orthonormal frame is valid).
#ifndef K16_EE_W4
#define K16_EE_W4 1
#endif
#ifndef K16_EE_PAIRS
#define K16_EE_PAIRS 1
#endif
constexpr int K16_SPLIT_DA = 13;
__host__ __device__ inline int k16_split_dims_of(int d, int KB, int enabled) {
  return (enabled && d > K16_SPLIT_DA && d + 6 <= 16 * KB) ? K16_SPLIT_DA : 0;
}
// content of physical K slot c: *coord >= 0 a coordinate; *piece 0..2 the pieces of N_A, 3..5 of N_B (PLAIN: of |r|^2); both -1: zero
__host__ __device__ inline void k16_slot(int c, int d, int dA, int* coord, int* piece) {
  *coord = -1;
  *piece = -1;
  if (dA == 0) {
    if (c < d) *coord = c;
    else if (c < d + 3) *piece = 3 + (c - d);
    return;
  }
  const int nb = d - dA;  // coordinates behind K block 0
  if (c < dA) *coord = c;
  else if (c < dA + 3) *piece = c - dA;
  else if (c < 16) return;
  else if (c < 16 + nb) *coord = dA + (c - 16);
  else if (c < 16 + nb + 3) *piece = 3 + (c - 16 - nb);
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// minimum of the 16 accumulator values of a lane: plain fminf so that the compiler forms
// v_min3_f32, schedules them between MFMAs and inserts the MFMA -> VALU wait states itself
// (an inline-asm v_min3 is invisible to the hazard recognizer: stale accumulators were read).
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float min16(const f32x16& v) {
  const float a = min3f(v[0], v[1], v[2]);
  const float b = min3f(v[3], v[4], v[5]);
  const float c = min3f(v[6], v[7], v[8]);
  const float d = min3f(v[9], v[10], v[11]);
  const float e = min3f(v[12], v[13], v[14]);
  return __builtin_fminf(min3f(a, b, v[15]), min3f(c, d, e));
}

__device__ __forceinline__ unsigned ordered_bits(float f) {
  const unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float ld_sc1_f(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_sc1_i(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// End of a cold region (row compaction) inside the scan loop: s_waitcnt vmcnt(0) issued through the BUILTIN,
// which the compiler's wait-count model sees (an inline-asm s_waitcnt it does not).  Without it the
// model carries "some load of the region may still be pending" over the loop back edge and guards the first
// ds_read of EVERY iteration with s_waitcnt vmcnt(0) -- i.e. the wave waits for the tile loads it has just
// issued instead of overlapping them with the iteration (measured: 134 ms with luck in the register
// allocation, 151 ms without).  Encoding (gfx9): vmcnt = 0, expcnt = 7, lgkmcnt = 15.
#define K16_COLD_REGION_END() __builtin_amdgcn_s_waitcnt(0x0F70)

// A candidate row is two half-rows: slots [0, half) are filled by the lanes that hold the query with
// h = 0, slots [half, 2 half) by the lanes with h = 1, each lane appending with a private register
// counter (no atomics, no returning LDS operation on the append path).  n0 / n1 = their lengths.
// Slot p of the row -> is it filled, and its position q in the packed sequence of n0 + n1 entries.
__device__ __forceinline__ bool knn16_row_slot(int p, int n0, int n1, int half, int* q) {
  if (p < half) {
    *q = p;
    return p < n0;
  }
  *q = n0 + (p - half);
  return (p - half) < n1;
}
// position pos of a packed sequence that is split as (m0 | rest) over the two half-rows
__device__ __forceinline__ int knn16_split_pos(int pos, int m0, int half) { return pos < m0 ? pos : half + (pos - m0); }

// Keep (unsorted) the entries of a candidate row whose d2 is <= the ksel-th smallest d2, spread evenly
// over the two half-rows.  Returns the new lengths through *n0_out / *n1_out and the threshold as
// return value.  Wave-uniform arguments.
__device__ __forceinline__ float from_ordered_bits(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Radius cut (knn1 > 0): with A = the knn1-th smallest d2 of the row (knn1 = knn + 1, self included) and E
// the search-error allowance of the row, the exact bandwidth^2 is <= A + E, so every reference inside the
// kernel radius has approximate d2 <= R = rf2 (A + E) + E: entries above R are dropped and *r_out = R
// (+inf when the row has fewer than knn1 entries) lets the caller lower the row's threshold to it.
// The rows hold RAW accumulator values during the scan, v = |r|^2 - 2 q.r = d2 - |q|^2 (nq_row = |q|^2 of the row is
// added when the row is ranked for output): the order within a row is the same, the append path needs no |q|^2, and
// the thresholds in registers are raw too.  Return value and *r_out are raw.
__device__ float knn16_squeeze_row(int n0, int n1, int half, int ksel, int knn1, float rf2, float err, float nq_row,
                                   float* __restrict__ d2row, int* __restrict__ idxrow, int lane, int* n0_out,
                                   int* n1_out, float* r_out) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's append stores have reached L2
  float d[K16_SLOTS];
  int ix[K16_SLOTS];
  unsigned key[K16_SLOTS];
  bool valid[K16_SLOTS];
#pragma unroll
  for (int e = 0; e < K16_SLOTS; ++e) {
    const int p = lane + 64 * e;
    int q;
    valid[e] = p < 2 * half && knn16_row_slot(p, n0, n1, half, &q);
    if (valid[e]) {
      d[e] = ld_sc1_f(d2row + p);
      ix[e] = ld_sc1_i(idxrow + p);
      key[e] = ordered_bits(d[e]);
    } else {
      d[e] = INFINITY;
      ix[e] = 0x7fffffff;
      key[e] = 0xffffffffu;
    }
  }
  // radix select: T = ksel-th smallest key  (largest T with count(key < T) < ksel)
  unsigned T = 0;
#pragma unroll 1
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned trial = T | (1u << bit);
    int c = 0;
#pragma unroll
    for (int e = 0; e < K16_SLOTS; ++e) c += __popcll(__ballot(key[e] < trial));
    if (c < ksel) T = trial;
  }
  float R = INFINITY;
  if (knn1 > 0 && n0 + n1 >= knn1) {
    unsigned A = 0;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned trial = A | (1u << bit);
      int c = 0;
#pragma unroll
      for (int e = 0; e < K16_SLOTS; ++e) c += __popcll(__ballot(key[e] < trial));
      if (c < knn1) A = trial;
    }
    R = (rf2 * ((from_ordered_bits(A) + nq_row) + err) + err) * 1.00001f + 1e-30f - nq_row;
  }
  *r_out = R;
  // survivors: key <= min(T, R) (ties at T all stay; if that leaves no room the caller re-ranks);
  // return value: the d2 whose key is T (the ksel-th smallest), +inf if the row is shorter than ksel
  const unsigned Tc = min(T, ordered_bits(R));
  int total = 0;
#pragma unroll
  for (int e = 0; e < K16_SLOTS; ++e) total += __popcll(__ballot(valid[e] && key[e] <= Tc));
  const int m0 = (total + 1) >> 1;
  int base = 0;
  float thr = INFINITY;
#pragma unroll
  for (int e = 0; e < K16_SLOTS; ++e) {
    const bool keep = valid[e] && key[e] <= Tc;
    const unsigned long long b = __ballot(keep);
    if (keep) {
      const int dst = knn16_split_pos(base + __popcll(b & ((1ull << lane) - 1ull)), m0, half);
      d2row[dst] = d[e];
      idxrow[dst] = ix[e];
    }
    base += __popcll(b);
    const unsigned long long bt = __ballot(valid[e] && key[e] == T);
    if (bt) thr = __shfl(d[e], __ffsll((long long)bt) - 1, 64);
  }
  *n0_out = m0;
  *n1_out = total - m0;
  return thr;
}

// Ordering of a row: rank by (d2, idx), keep the ksel smallest sorted, scale by out_scale.
// split_m0 < 0: final form, entry of rank r at slot r.  split_m0 >= 0 (mid-scan, after pathological
// ties): rank r goes to the half-row position knn16_split_pos(r, split_m0, half).
// Values are written as (v + add) * out_scale: add = |q|^2 of the row in the final form (the rows hold raw values,
// see knn16_squeeze_row), 0 mid-scan.
__device__ void knn16_rank_row(int n0, int n1, int half, int ksel, float out_scale, float add, int split_m0,
                               float* __restrict__ d2row, int* __restrict__ idxrow, float* sd, int* si, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int n = n0 + n1;
  float d[K16_SLOTS];
  int ix[K16_SLOTS];
  bool valid[K16_SLOTS];
#pragma unroll
  for (int e = 0; e < K16_SLOTS; ++e) {
    const int p = lane + 64 * e;
    int q = 0;
    valid[e] = p < 2 * half && knn16_row_slot(p, n0, n1, half, &q);
    if (valid[e]) {
      d[e] = ld_sc1_f(d2row + p);
      ix[e] = ld_sc1_i(idxrow + p);
      sd[q] = d[e];
      si[q] = ix[e];
    } else {
      d[e] = INFINITY;
      ix[e] = 0x7fffffff;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int rk[K16_SLOTS];
#pragma unroll
  for (int q = 0; q < K16_SLOTS; ++q) rk[q] = 0;
  for (int e = 0; e < n; ++e) {
    const float de = sd[e];
    const int ie = si[e];
#pragma unroll
    for (int q = 0; q < K16_SLOTS; ++q) rk[q] += (de < d[q] || (de == d[q] && ie < ix[q])) ? 1 : 0;
  }
#pragma unroll
  for (int q = 0; q < K16_SLOTS; ++q) {
    if (valid[q] && rk[q] < ksel) {
      const int dst = split_m0 < 0 ? rk[q] : knn16_split_pos(rk[q], split_m0, half);
      d2row[dst] = (d[q] + add) * out_scale;
      idxrow[dst] = ix[q];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// Scan order of a query block: step s -> tile.  Own tiles first (t0 .. t0 + 3), then outwards on both sides of the own
// position (two_sided): in the locality order both the next and the previous leaves are spatial neighbours, so the
// thresholds tighten sooner than on a forward-only walk.  Offsets -B .. F-1 with F + B = n_scan: every tile once.
// (Shared by the search kernel and by meld_knn16_step_lists, which must agree on it.)
__device__ __forceinline__ int k16_scan_tile(int s, int t0, int n_scan, int two_sided) {
  int off = s;
  if (two_sided && s >= K16_BQ / K16_TS) {
    const int j = s - K16_BQ / K16_TS;
    off = (j & 1) ? -1 - (j >> 1) : K16_BQ / K16_TS + (j >> 1);
  }
  const int t = t0 + off;
  return t >= n_scan ? t - n_scan : (t < 0 ? t + n_scan : t);
}

// Kernel arguments of knn16_topk_kernel, one struct passed by value: its layout IS the kernel-argument
// segment, which lets cold code re-read a field where it is needed instead of holding it in an SGPR for
// the whole scan.
struct K16Args {
  const _Float16* Q16;
  const float* Qn;
  const _Float16* Rt16;
  const float* scale_info;
  int n_ref, n_tiles, ksel, cap;
  const __half* lb2;
  const float* norm2_max;
  float err_coef;
  int tile_origin, batch_every, batch_slack, two_sided;
  unsigned long long* stats;
  const float* thr_init;
  int knn1;
  float rf2, err_c, err_l;
  int* cand_idx;
  float* cand_d2;
  int* cand_cnt;
  float* cand_thr;
  unsigned long long* tiles_done;
  const int* block_order;
  // LIST kernels: the steps of every query block, precomputed (meld_knn16_step_lists): entry = tile | live waves << 24
  const unsigned* step_list;
  const int* step_cnt;
  long long list_stride;
  int ee_hi;  // EE kernels (SPLIT layout): physical K slots [16, ee_hi) hold the coordinates behind K block 0
  int planes_used;  // 64-vector planes (8 K slots each) of a reference tile that hold anything: the hi-only pass copies no others
  int count_go;     // EE kernels: tiles_done has a second slot for the blocks that went on (meld_knn16_topk_listed_partial only)
};
#define K16_COLD(FIELD) \
  (((const volatile K16Args __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr())->FIELD)

// Waves per SIMD the kernel is compiled for (min = max).  Pinning it (a) keeps the MFMA results in VGPRs
// -- with a 512-register budget hipcc selects the AGPR form and every vote pays 32 v_accvgpr_read -- and
// (b) lets the scheduler keep the interleaved MFMA / vote order instead of trading it for occupancy it
// cannot reach.  The first pass at d <= 61 fits three waves (168 registers): measured 135 ms vs 151 ms
// with two at 1M cells -- the waves mostly wait (barrier per tile, selection slow path), so occupancy pays.
__host__ __device__ constexpr int k16_waves(int KB, int ABL, int NPROD) {
  // (the full split at KB = 6 would fit two by registers, but its three 24 KB tile buffers leave room for one workgroup per CU)
  return (NPROD == 1 && KB <= 4 && ABL != 6) ? 3 : ((KB <= 5 || (NPROD == 1 && KB <= 8)) ? 2 : 1);
}

template <int KB, int ABL, int NPROD, bool LIST = false, bool EE = false>  // 16 KB >= d + 3; ABL: 0 = product, 2 = product + selection counters, 1 / 3 = profiling ablations (no selection / MFMAs
                                      // only), 6 = product at two waves per SIMD where three are the default;
                                      // NPROD: split products (1 = hi.hi only, 3 = hi.hi + hi.lo + lo.hi)
                                      // LIST: the steps of a query block (tile + the waves that need it) come from a precomputed list
                                      // (meld_knn16_step_lists) instead of the pruning table: no per-step masks, ballots or window logic
                                      // EE (with LIST, NPROD = 1, KB >= 2, operands in the SPLIT layout): a block of 32 references is tested
                                      // on its accumulators behind K block 0 and dropped when no partial value is within reach of its row
__global__ __launch_bounds__(K16_THREADS)
__attribute__((amdgpu_waves_per_eu(k16_waves(KB, ABL, NPROD) + ((EE && K16_EE_PAIRS && K16_EE_W4 && KB <= 4) ? 1 : 0),
                                   k16_waves(KB, ABL, NPROD) + ((EE && K16_EE_PAIRS && K16_EE_W4 && KB <= 4) ? 1 : 0)))) void knn16_topk_kernel(const K16Args a) {
  // Arguments the scan loop needs stay in SGPRs; the cold ones (K16_COLD: compaction parameters, the outputs
  // of the epilogue, profiling) are re-read from the kernel-argument segment where they are used -- kept live
  // across the loop they pushed the kernel past its 102 SGPRs, the spills went to VGPR lanes and on to
  // scratch, and the MFMA / vote interleaving fell apart (measured: 74 -> 81 ms).
  const _Float16* __restrict__ Q16 = a.Q16;
  const float* __restrict__ Qn = a.Qn;
  const _Float16* __restrict__ Rt16 = a.Rt16;
  const int n_tiles = a.n_tiles, ksel = a.ksel, cap = a.cap;
  const __half* __restrict__ lb2 = LIST ? nullptr : a.lb2;
  const int batch_every = a.batch_every, two_sided = a.two_sided;
  int* __restrict__ cand_idx = a.cand_idx;
  float* __restrict__ cand_d2 = a.cand_d2;
  // reference tile = KB K-blocks [kb][k-half][plane][ref][8 halves]; K slots d .. d+2 of the hi plane
  // hold |r|^2 as three fp16 pieces (against 1.0 on the query side), so the MFMAs deliver
  // |r|^2 - 2 q.r directly and no norm is read in the loop
  constexpr int TILE_H = KB * 2 * 2 * K16_TS * 8;  // halves per reference tile
  constexpr int TILE_V4 = TILE_H / 8;              // 16-byte vectors per tile = KB * 256
  constexpr int LDS_TILE_H = (NPROD == 3) ? TILE_H : TILE_H / 2;  // (the hi-only search keeps just the hi planes in LDS)
  // A ring of three tile buffers: the tile of the step after next is in flight across the tile barrier.  A tile
  // requested at the top of an iteration and needed at its end has one iteration (~1.6 us at 1M cells) to arrive
  // from the Infinity Cache; requested one iteration earlier it has two, and the wave only waits for the OLDER of
  // its two requests (counted s_waitcnt vmcnt(n) + a raw s_barrier: __syncthreads() would drain the counter).
  // The copies (global -> LDS DMA) are issued from inline asm: the compiler's wait-count pass knows that the
  // builtin form writes LDS and guards every ds_read that may alias a pending copy with s_waitcnt vmcnt(0) -- with
  // a ring addressed by a rotating offset that is every A-fragment read.  (The earlier build kept two separate
  // arrays and the loop unrolled by two so that the pass could prove them disjoint; three arrays unrolled by three
  // spill and no longer fit the instruction cache.)  The waits are explicit instead: K16_STAGED / _BUT_LAST.
  // PAIR (EE kernels, -DK16_EE_PAIRS): two tiles per step and barrier, two buffers of two tiles -- the steps of the partial-test
  // pass are short, and a barrier + the control around it per 64 references is a third of what a wave spends
  constexpr bool PAIR = EE && (K16_EE_PAIRS != 0);
  constexpr int NBUF = PAIR ? 4 : 3;
  __shared__ __attribute__((aligned(16))) _Float16 lds_ring[NBUF * LDS_TILE_H];
  constexpr int TILE_LDS_BYTES = LDS_TILE_H * 2;
  __shared__ float lds_sd[K16_NWAVE][K16_CAPMAX];
  __shared__ int lds_si[K16_NWAVE][K16_CAPMAX];
  // per-wave live-step masks of the pruning window: [0/1] by step parity, [2] window switch.  Table-driven kernels only: the
  // list-driven ones never read them and let the name alias the ranking scratch -- their 96 bytes are what separates four
  // workgroups of the two-tile pass from the CU's 160 KiB of LDS
  unsigned long long (*lds_wlive)[K16_NWAVE];
  if constexpr (!LIST) {
    __shared__ unsigned long long lds_wlive_own[3][K16_NWAVE];
    lds_wlive = lds_wlive_own;
  } else {
    lds_wlive = reinterpret_cast<unsigned long long (*)[K16_NWAVE]>(&lds_sd[0][0]);
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform by construction; the compiler cannot tell)
  const int jq = lane & 31;
  const int h = lane >> 5;
  // Query block of this workgroup.  Workgroups are dispatched in index order as slots free up, and with pruning their
  // work differs by up to 12x: block_order (optional) lists the query blocks by decreasing work, so that the longest
  // start first and the chip does not wait for a heavy block that was dispatched last (meld_knn16_block_work).
  // (grid = slices x query blocks: the slices of a block are dispatched together, so "longest blocks first" holds across slices)
  const int bx = a.block_order ? __builtin_amdgcn_readfirstlane(a.block_order[blockIdx.y]) : (int)blockIdx.y;
  const int q_base = bx * K16_BQ + wave * 64;  // first query of this wave
  // reference slices (gridDim.x > 1): slice y scans the tiles y, y + S, y + 2 S, ... (interleaved: with the cells in
  // locality order the tiles a query block cannot rule out sit in a few runs, and contiguous slices would leave most of a
  // heavy block's work in one of them) and writes its own candidate rows; meld_knn16_merge_slices combines them.  Used to
  // spread few query blocks (the re-search, a row shard, a mid-sized data set) over the whole chip.
  int sl_n_ = (int)gridDim.x;
#ifndef K16_NO_SLN_PIN
  // (opaque to the rematerialiser: otherwise gridDim.x is re-read from the dispatch packet -- a scalar load and its
  // s_waitcnt lgkmcnt(0) -- in tile_of() of every iteration)
  asm volatile("" : "+s"(sl_n_));
#endif
  const int sl_n = sl_n_, sl_y = (int)blockIdx.x;
  // LIST: the block's steps = the entries of its list (tile | waves that cannot rule the tile out << 24), in scan order;
  // read with scalar loads (constant address space), two steps ahead of their use
  typedef const __attribute__((address_space(4))) unsigned* k16_list_ptr;
  const k16_list_ptr my_list = LIST ? (k16_list_ptr)(uintptr_t)(a.step_list + (size_t)bx * (size_t)a.list_stride) : (k16_list_ptr)0;
  // (reference slices of a LIST launch: slice y takes the entries y, y + S, ... of the block's list -- every slice gets its share
  // of the nearest tiles, where the thresholds tighten, and the slices of a block are equally long)
  const int n_scan = ((LIST ? __builtin_amdgcn_readfirstlane(a.step_cnt[bx]) : n_tiles) - sl_y + sl_n - 1) / sl_n;
  auto list_entry = [&](int s_) __attribute__((always_inline)) { return s_ < n_scan ? my_list[sl_y + sl_n * s_] : 0u; };
  auto entry_live = [&](unsigned e) __attribute__((always_inline)) { return ((e >> (24 + wave)) & 1u) != 0u; };
  const int row_base = (int)(blockIdx.x * (gridDim.y * K16_BQ)) + q_base;  // first candidate row of this wave
  float* const wave_d2 = cand_d2 + (size_t)row_base * cap;                // (wave-uniform: SGPR pairs)
  int* const wave_idx = cand_idx + (size_t)row_base * cap;

  // B fragments of both query groups: [g][kb] hi / lo, 8 halves each
  f16x8 bhi[2][KB], blo[2][KB];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const f16x8* qrow = reinterpret_cast<const f16x8*>(Q16 + (size_t)(q_base + g * 32 + jq) * (KB * 32));
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      bhi[g][kb] = qrow[(kb * 2 + h) * 2 + 0];
      if (NPROD == 3) blo[g][kb] = qrow[(kb * 2 + h) * 2 + 1];
    }
  }
  // |q|^2 is needed in cold code only (the thresholds in registers are raw, thrp = thr - |q|^2): it is re-read from
  // memory there instead of occupying two registers across the scan (the kernel sits at its register budget)
  const float* const wave_qn = Qn + q_base;

  int cnt[2] = {0, 0};  // entries this lane has appended to its half of the rows of its two queries
  const int half = cap >> 1;
  if (!LIST && lane == 0) lds_wlive[0][wave] = lds_wlive[1][wave] = lds_wlive[2][wave] = ~0ull;
  // thr_init (optional, scaled units): a bound the caller knows every wanted neighbour to lie below (the
  // re-search of rows whose first-pass list could not be certified knows one); it starts the thresholds
  // there instead of at +inf, so that only a handful of candidates per query ever take the slow path
  // (only thrp = thr - |q|^2 lives in registers: the kernel is compiled for a fixed register budget)
  auto thr_start = [&](int g) __attribute__((always_inline)) { return a.thr_init ? a.thr_init[q_base + g * 32 + jq] : INFINITY; };
  float thrp[2] = {thr_start(0) - wave_qn[jq], thr_start(1) - wave_qn[32 + jq]};
  // EE: what the K blocks behind the first can take away from an accumulator at most, per query: |q_hiB|^2 (the fp16 values the
  // MFMAs multiply, summed here from the fragments in registers: the two half-lanes of a query hold the two halves of every K
  // block) rounded up, plus the fp32 rounding of the three later accumulations (<= 2^-18 n_max with room to spare: the partial
  // sums stay below 3 n_max).  A block is dropped when every partial value is >= thrp + ee_c: v_full < thrp implies
  // v_partial = v_full - (rest) < thrp + |q_hiB|^2 + rounding.
  float ee_c[2] = {0.0f, 0.0f};
  if constexpr (EE) {
    const int ee_hi = a.ee_hi;
    const float s_ = a.scale_info[0];
    const float delta = 3.814697265625e-06f * (a.norm2_max[0] * s_ * s_) + 1e-30f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float qb = 0.0f;
#pragma unroll
      for (int kb = 1; kb < KB; ++kb) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = (16 * kb + 8 * h + e < ee_hi) ? (float)bhi[g][kb][e] : 0.0f;
          qb = fmaf(v, v, qb);
        }
      }
      qb += __shfl_xor(qb, 32, 64);
      ee_c[g] = qb * 1.0001f + delta;
    }
  }
  float wmax = INFINITY;  // max threshold over this wave's 64 queries (wave-uniform)
  if (a.thr_init) {  // seeded thresholds also seed the pruning bound (and let the timing ablations prune realistically)
    float w = fmaxf(thr_start(0), thr_start(1));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w = fmaxf(w, __shfl_xor(w, off, 64));
    wmax = w;
  }
  unsigned st_go = 0;  // (EE, ABL == 2: blocks of 32 references that went on past K block 0)
  unsigned st_slow = 0, st_app = 0, st_sq = 0;  // profiling (ABL == 2 only): slow-path entries, appends (per lane), compactions
  // ... and where a wave's cycles go (s_memtime stamps at points where the LDS / scalar counter is drained anyway): MFMA segments
  // of a live step, slow path, control tail up to the tile wait, the tile wait, the tile barrier; steps sat out
  unsigned tm_seg = 0, tm_sel = 0, tm_tail = 0, tm_dma = 0, tm_bar = 0, tm_idle = 0, tm_mark = 0;
  auto tm_now = [&]() __attribute__((always_inline)) { return ABL == 2 ? (unsigned)__builtin_readcyclecounter() : 0u; };
  auto tm_lap = [&](unsigned& acc) __attribute__((always_inline)) {
    if (ABL == 2) {
      const unsigned n = tm_now();
      acc += n - tm_mark;
      tm_mark = n;
    }
  };

  // Tiles are visited starting at the workgroup's own position (its spatial neighbourhood when
  // the cells are in locality order) and wrapping around: step s -> tile (t0 + s) mod n_tiles.
  // lb2 (optional) holds, per (workgroup, tile), a lower bound on the squared distance between
  // any query of the workgroup and any reference of the tile; a tile whose bound exceeds every
  // threshold of the workgroup cannot contribute a candidate and is skipped without being loaded.
  // search-error allowance in the scaled space: a skipped tile must fail `d2_approx < thr` for sure
  const float prune_margin = lb2 ? a.err_coef * a.norm2_max[0] * a.scale_info[0] * a.scale_info[0] : 0.0f;
  const int t0 = (int)((((long long)a.tile_origin + (long long)bx * (K16_BQ / K16_TS)) / sl_n) % n_scan);
  const __half* my_lb = lb2 ? lb2 + (size_t)(bx * K16_NWAVE + wave) * n_tiles : nullptr;
  // (A "convoy" order -- all resident workgroups sweeping the same tiles at the same time so that all but
  // the first find them in L2 -- was tried: the base loop gained 7 %, but the thresholds converge later and
  // the net was +3 %; its progress counter was also a returning atomic in the loop, whose pending result made
  // the compiler guard the first ds_read of every iteration with s_waitcnt vmcnt(0).  Removed.)
  // (LIST kernels never call it: their steps are list entries)
  auto tile_of = [&](int s) { return k16_scan_tile(s, t0, n_scan, two_sided) * sl_n + sl_y; };
  // Pruning window (64 steps): lane i holds THIS WAVE's bound for step win_base + i (lb2 has one row per wave:
  // the wave's 64 queries against every tile).  my_live = steps of the window whose tile may still hold a
  // candidate for the wave (its bound <= the wave's largest threshold + the search-error allowance); the waves
  // publish their masks in LDS before every tile barrier, the union decides which tile the workgroup stages
  // next, and a wave whose own bit is clear sits the tile out (no MFMAs, no vote).  win_rem = steps not yet
  // passed.  The hot loop touches no memory for this: the table is read once per 64 steps in a cold region at
  // the loop top, where no tile load is in flight.  Thresholds only decrease, so a cleared bit stays clear.
  int win_base = 0;
  float win_lb = 0.0f;
  unsigned long long win_rem = 0, my_live = ~0ull;
  auto load_window = [&](int base) __attribute__((always_inline)) {
    const int ss = base + lane;
    win_base = base;
    win_lb = ss < n_scan ? __half2float(my_lb[tile_of(ss)]) : INFINITY;
    win_rem = __ballot(ss < n_scan);
    asm volatile("" : "+v"(win_lb));
  };
  if (my_lb) {
    load_window(0);
    win_rem &= ~1ull;  // step 0 is the first tile
  }
  auto lds_union = [&](int buf) __attribute__((always_inline)) {
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int w = 0; w < K16_NWAVE; ++w) {
      const uint2 v = *reinterpret_cast<const uint2*>(&lds_wlive[buf][w]);
      lo |= v.x;
      hi |= v.y;
    }
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)hi) << 32) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)lo);
  };

  const float4* R4 = reinterpret_cast<const float4*>(Rt16);
  // The base of the reference tiles is needed once per iteration (the copy's buffer descriptor).  Under the kernel's SGPR
  // pressure the compiler re-read it from the kernel-argument segment each time -- a scalar load and its wait in front of
  // every copy request -- so it rides in two VGPRs instead (opaque to the rematerialiser) and comes back with
  // v_readfirstlane.
  int r4_lo_v, r4_hi_v;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
               : "=v"(r4_lo_v), "=v"(r4_hi_v)
               : "s"((int)(unsigned)reinterpret_cast<size_t>(R4)), "s"((int)(reinterpret_cast<size_t>(R4) >> 32)));
  // Staging: global -> LDS directly (`buffer_load_dwordx4 ... lds`, gfx950's 16-byte LDS DMA): lane l of a wave
  // writes LDS slot M0-base + 16 l, so a wave copies 64 consecutive LDS vectors per instruction from per-lane
  // global offsets; no staging registers (8 VGPRs at d = 50 that now hold A fragments in flight), no ds_write,
  // and the data never passes through the VALU.  NPROD == 3 copies the whole tile (vector tid + 256 u).
  // NPROD == 1 never reads the lo planes, so only the hi vectors are copied: the j-th of them,
  // j = tid + 256 u, sits at ((j >> 6) << 7) + (j & 63) in the tile (one 64-vector plane out of every 128)
  // and lands at LDS slot j -- half the staging traffic and half the LDS.
  constexpr int N_STAGE = (NPROD == 3) ? KB * 256 : KB * 128;           // vectors to copy per tile
  constexpr int NS = (N_STAGE + K16_THREADS - 1) / K16_THREADS;         // rounds
  constexpr bool STAGE_TAIL = (N_STAGE % K16_THREADS) != 0;             // last round half empty (odd KB, NPROD == 1)
  static_assert(NS <= 9, "tile too large for the staging rounds");
  const int stage_off = (NPROD == 3) ? tid : (((tid >> 6) << 7) + (tid & 63));
  constexpr int STAGE_STRIDE = (NPROD == 3) ? K16_THREADS : 2 * K16_THREADS;  // offset step per round
  const bool stage_last = !STAGE_TAIL || wave < (N_STAGE % K16_THREADS) / 64;  // (wave-uniform)
  // NPROD == 1: the last 64-vector plane of a tile (K slots 8 p .. 8 p + 7, one wave's copy of the last round) may hold nothing but
  // padding -- d = 50: 56 of 64 slots used -- and is then never copied: its LDS slots are zeroed once (the queries' slots there are
  // zero too, but 0 x stale bits could be a NaN).  An eighth of the tile stream at d = 50, and the stream is what bounds the
  // kernel once the partial test has thinned the MFMAs (105 GB per launch at the fabric's copy rate).
  // (the plane must exist: with an odd number of K blocks the last round has planes for half of the waves only -- the others
  // would zero the head of the NEXT ring buffer under its first copy: one build in twenty lost a block's candidates that way)
  const int tail_plane = (NS - 1) * (K16_THREADS / 64) + wave;
  const bool skip_tail = NPROD == 1 && NS >= 2 && tail_plane < 2 * KB && tail_plane >= a.planes_used;  // (wave-uniform)
  if (skip_tail) {
#pragma unroll
    for (int b = 0; b < NBUF; ++b)
      reinterpret_cast<float4*>(lds_ring + b * LDS_TILE_H)[((NS - 1) * (K16_THREADS / 64) + wave) * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#define K16_ROUND_OK(U) (((U) + 1 < NS || stage_last) && !((U) + 1 == NS && skip_tail))
  // The descriptor (SGPRs) carries the tile base, the per-thread offset is one loop-invariant VGPR and the
  // round offset an SGPR.
  const int stage_voff = stage_off * 16;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef int k16_i32x4 __attribute__((ext_vector_type(4)));
  const unsigned ring_addr = (unsigned)(size_t)(lds_ptr_t)lds_ring + (unsigned)wave * 1024u;  // LDS byte address of this wave's slots
  // (M0 = LDS base of the copy; the compiler treats M0 as scratch and sets it before each of its own uses)
#define K16_DMA(U, DSTB)                                                                                              \
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(    \
                   (int)(ring_addr + (DSTB) + (unsigned)((U) * K16_THREADS * 16)))),                                   \
               "v"(stage_voff), "s"(rsrc), "s"((U) * STAGE_STRIDE * 16)                                                \
               : "memory")
#define K16_LOAD(TILE, DSTB)                                                                                       \
  do {                                                                                                             \
    const size_t src_ = (((size_t)(unsigned)__builtin_amdgcn_readfirstlane(r4_hi_v) << 32) |                       \
                         (size_t)(unsigned)__builtin_amdgcn_readfirstlane(r4_lo_v)) +                              \
                        (size_t)(TILE) * (TILE_V4 * 16);                                                           \
    const k16_i32x4 rsrc = {(int)(unsigned)src_, (int)((src_ >> 32) & 0xffffu), TILE_V4 * 16, 0x00020000};         \
    if constexpr (NS > 0) if (K16_ROUND_OK(0)) K16_DMA(0, DSTB);                                                   \
    if constexpr (NS > 1) if (K16_ROUND_OK(1)) K16_DMA(1, DSTB);                                                   \
    if constexpr (NS > 2) if (K16_ROUND_OK(2)) K16_DMA(2, DSTB);                                                   \
    if constexpr (NS > 3) if (K16_ROUND_OK(3)) K16_DMA(3, DSTB);                                                   \
    if constexpr (NS > 4) if (K16_ROUND_OK(4)) K16_DMA(4, DSTB);                                                   \
    if constexpr (NS > 5) if (K16_ROUND_OK(5)) K16_DMA(5, DSTB);                                                   \
    if constexpr (NS > 6) if (K16_ROUND_OK(6)) K16_DMA(6, DSTB);                                                   \
    if constexpr (NS > 7) if (K16_ROUND_OK(7)) K16_DMA(7, DSTB);                                                   \
    if constexpr (NS > 8) if (K16_ROUND_OK(8)) K16_DMA(8, DSTB);                                                   \
  } while (0)
  // the copies of this wave have landed in LDS (then the tile barrier makes them visible to the others)
#define K16_STAGED() __builtin_amdgcn_s_waitcnt(0x0F70)
  // ... all but the copies of the request issued last (every wave issues at least NS - 1 copies per tile)
  constexpr int N_INFLIGHT = NS - (STAGE_TAIL ? 1 : 0);
  // (a wave that skips its copy of the last round has one request fewer in flight per tile)
  constexpr int N_INFLIGHT_SKIP = (!STAGE_TAIL && NS >= 2) ? N_INFLIGHT - 1 : N_INFLIGHT;
#define K16_STAGED_BUT_LAST()                                                                                      \
  do {                                                                                                             \
    if (skip_tail) __builtin_amdgcn_s_waitcnt(0x0F70 | (N_INFLIGHT_SKIP & 15) | ((N_INFLIGHT_SKIP >> 4) << 14));   \
    else __builtin_amdgcn_s_waitcnt(0x0F70 | (N_INFLIGHT & 15) | ((N_INFLIGHT >> 4) << 14));                       \
  } while (0)
  // tile barrier that leaves the vector-memory counter alone (LDS traffic of this wave done, then s_barrier)
#define K16_TILE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  if constexpr (!PAIR) K16_LOAD(__builtin_amdgcn_readfirstlane(LIST ? (int)(list_entry(0) & 0xFFFFFFu) : tile_of(0)), 0u);

  // The query fragments / norms must have landed BEFORE the loop: otherwise the compiler sinks
  // their loads past the first barrier and then has to guard their first use inside the loop with
  // s_waitcnt vmcnt(N) -- which, on every later iteration, also waits for the tile loads that were
  // just issued (vmcnt is in-order) and exposes their latency.  An asm use forces the wait here.
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      asm volatile("" : "+v"(bhi[g][kb]));
      if (NPROD == 3) asm volatile("" : "+v"(blo[g][kb]));
    }
    asm volatile("" : "+v"(thrp[g]));  // (its start value may come from thr_init)
  }

  // ---- software-pipelined scan -------------------------------------------------------------
  // A block = 32 references x the wave's 64 queries = one pair of 32x32 accumulator tiles.  A wave
  // issues in order, so the only way to keep the matrix pipe busy while a block is voted on is to
  // interleave, in program order, the MFMAs of block b with the (straight-line) vote on block b-1:
  // two accumulator pairs alternate, accA = sub-tile 0 of the current tile, accB = sub-tile 1 (of the
  // previous tile at the loop top).  The vote is plain C++ (no inline asm) so that the compiler
  // schedules it between the MFMAs and supplies the MFMA -> VALU wait states itself.
  f32x16 accA0, accA1, accB0, accB1;
#pragma unroll
  for (int r = 0; r < 16; ++r) accB0[r] = accB1[r] = INFINITY;  // nothing to vote on before the first tile
  int refB = 0;

  // |r|^2 - 2 q.r of sub-tile `sub` of the tile at `tile` (the norm rides in K slots d .. d+2,
  // |q|^2 is folded into the threshold: thrp = thr - |q|^2)
  auto mfma_block = [&](const _Float16* tile, int sub, f32x16& c0, f32x16& c1) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.0f;
    // tile layout [kb][h][plane][i][8 halves]: lane reads 16 B at ((kb*2+h)*2+plane)*TS + i
    const f16x8* a8 = reinterpret_cast<const f16x8*>(tile) + sub * 32 + jq;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      // hi parts alone (NPROD == 1, error <= 2^-9 |x~||y~|, see meld_knn16_error_coef; LDS layout
      // [kb][h][ref]) or the full hi/lo split (LDS layout = tile layout [kb][h][plane][ref])
      const f16x8 ahi = NPROD == 3 ? a8[((kb * 2 + h) * 2 + 0) * K16_TS] : a8[(kb * 2 + h) * K16_TS];
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[0][kb], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[1][kb], c1, 0, 0, 0);
      if (NPROD == 3) {
        const f16x8 alo = a8[((kb * 2 + h) * 2 + 1) * K16_TS];
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo[0][kb], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo[1][kb], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi[0][kb], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi[1][kb], c1, 0, 0, 0);
      }
    }
  };
  // minimum over the lane's 16 references of each query group (straight-line VALU)
  auto vote = [&](const f32x16& c0, const f32x16& c1, float& m0, float& m1) __attribute__((always_inline)) {
    m0 = min16(c0);
    m1 = min16(c1);
  };
  // compact row j of query group g (wave-uniform arguments) and refresh its threshold
  auto squeeze = [&](int g, int j) __attribute__((always_inline)) {
    const int cg = g ? cnt[1] : cnt[0];
    const int n0 = __shfl(cg, j, 64), n1 = __shfl(cg, j + 32, 64);
    const size_t ro = (size_t)(g * 32 + j) * cap;  // (within the wave's rows)
    int m0, m1;
    float rcut = INFINITY, err = 0.0f;
    const int knn1 = K16_COLD(knn1);
    const float nq_row = wave_qn[g * 32 + j];
    if (knn1 > 0) {  // search-error allowance of this row, scaled units (refine.hip's E, rounded up)
      const float* sinfo = K16_COLD(scale_info);
      const float nmax_s = K16_COLD(norm2_max)[0] * sinfo[0] * sinfo[0];
      err = (K16_COLD(err_c) * nmax_s + K16_COLD(err_l) * sqrtf(nq_row * nmax_s)) * 1.001f;
    }
    float nt = knn16_squeeze_row(n0, n1, half, ksel, knn1, K16_COLD(rf2), err, nq_row, wave_d2 + ro, wave_idx + ro, lane, &m0, &m1, &rcut);
    if (m0 > half - 32) {
      // pathological ties at the threshold: rank the row down to exactly ksel entries
      knn16_rank_row(m0, m1, half, ksel, 1.0f, 0.0f, (min(m0 + m1, ksel) + 1) >> 1, wave_d2 + ro, wave_idx + ro, lds_sd[wave],
                     lds_si[wave], lane);
      const int tot = min(m0 + m1, ksel);
      m0 = (tot + 1) >> 1;
      m1 = tot - m0;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      nt = ld_sc1_f(wave_d2 + ro + knn16_split_pos(tot - 1, m0, half));
    }
    if (jq == j) {
      const int mine = h ? m1 : m0;
      if (g) cnt[1] = mine; else cnt[0] = mine;
      // list threshold (the ksel-th smallest once the row has ksel entries) or the radius cut, whichever is
      // lower: the row holds every reference seen so far whose d2 is below it
      // (a row shorter than ksel keeps its threshold -- the start value -- unless the radius cut lowers it)
      const float t_list = (n0 + n1 >= ksel) ? nt : INFINITY;
      thrp[g] = fminf(thrp[g], fminf(t_list, rcut));  // (raw, like the row's entries)
    }
    // every load of this cold region has landed when it ends: a result still pending at the join with
    // the hot loop would make the compiler put s_waitcnt vmcnt(0) in front of the next pipeline segment
    asm volatile("" : "+v"(thrp[0]), "+v"(thrp[1]), "+v"(cnt[0]), "+v"(cnt[1]));
  };
  auto refresh_wmax = [&]() __attribute__((always_inline)) {
    if (my_lb == nullptr) return;  // only the pruning test reads it
#ifdef K16_NO_WMAX_REFRESH
    return;  // (experiment: the live sets stay what the start thresholds make them)
#endif
    float w = fmaxf(thrp[0] + wave_qn[jq], thrp[1] + wave_qn[32 + jq]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) w = fmaxf(w, __shfl_xor(w, off, 64));
    wmax = w;
  };
  // slow path: append the entries below the thresholds to the candidate rows, compact full rows.
  // Hits are sparse (well under one per block on average), so the 16 values of a lane are not walked
  // one by one: the minimum of each triple is tested wave-wide first and only a triple with a hit
  // is opened (padding references have |r|^2 = +inf and never pass, so no index test is needed).
  auto select = [&](const f32x16& c0, const f32x16& c1, float m0, float m1, int ref_base) __attribute__((always_inline)) {
    unsigned long long need = 0;  // rows to compact: bit 32 g + j
    if (ABL == 2) ++st_slow;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const f32x16 acc = g ? c1 : c0;
      if (!__any((g ? m1 : m0) < thrp[g])) continue;
      // byte offset of the lane's half-row within the wave's 64 rows: 32 bits, so that a store is one address
      // instruction (wave base in SGPRs + 32-bit lane offset) instead of a 64-bit multiply-add per append
      const unsigned rowoff4 = (unsigned)((g * 32 + jq) * cap + h * half) * 4u;
      auto append = [&](int r) __attribute__((always_inline)) {
        const float v = acc[r];
        if (v < thrp[g]) {
          const unsigned off4 = rowoff4 + 4u * (unsigned)(g ? cnt[1]++ : cnt[0]++);
          if (ABL == 2) ++st_app;
          if (ABL == 5) {  // (timing-only ablation: the append without its stores)
            asm volatile("" ::"v"(off4));
          } else {
            *reinterpret_cast<float*>(reinterpret_cast<char*>(wave_d2) + off4) = v;  // raw: |q|^2 is added when the row is ranked
            *reinterpret_cast<int*>(reinterpret_cast<char*>(wave_idx) + off4) = ref_base + (r & 3) + 8 * (r >> 2);
          }
        }
      };
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        if (__any(min3f(acc[3 * t], acc[3 * t + 1], acc[3 * t + 2]) < thrp[g])) {
          append(3 * t);
          append(3 * t + 1);
          append(3 * t + 2);
        }
      }
      if (__any(acc[15] < thrp[g])) append(15);
      // a half-row is compacted as soon as fewer than 16 free slots remain: a block adds at most 16
      // entries to it (the lane's 16 references), so an append never runs out of room
      const unsigned long long full = __ballot((g ? cnt[1] : cnt[0]) > half - 16);
      need |= ((full | (full >> 32)) & 0xffffffffull) << (32 * g);
    }
    if (__builtin_expect(need != 0, 0)) {
      while (need) {
        const int j = __ffsll((long long)need) - 1;
        need &= need - 1;
        squeeze(j >> 5, j & 31);
        if (ABL == 2) ++st_sq;
      }
      refresh_wmax();
      K16_COLD_REGION_END();
    }
  };
  // Issue order of one pipeline segment (the MFMAs of a block + the vote on the previous one, all in
  // one basic block): the A-fragment reads first, a few vote instructions to cover their latency,
  // then every MFMA followed by the VALU instructions that fit into its 32-cycle pipe slot.
  constexpr int N_MFMA = 2 * KB * (NPROD == 3 ? 3 : 1);
  constexpr int N_DSR = KB * (NPROD == 3 ? 2 : 1);
  constexpr int VALU_PER_MFMA = (N_MFMA >= 16) ? 1 : ((N_MFMA >= 8) ? 2 : 4);
  auto pipeline_order = [&]() __attribute__((always_inline)) {
    if (NPROD == 3) {
      __builtin_amdgcn_sched_group_barrier(0x100, N_DSR, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#pragma unroll
      for (int i = 0; i < N_MFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
      }
    } else {
      // A_AHEAD A fragments in flight: the fragment of K block kb + A_AHEAD is requested right after the MFMAs of
      // block kb have issued (with |q|^2 out of the registers the budget of three waves per SIMD has room for four)
      constexpr int A_AHEAD = K16_A_AHEAD;  // A fragments in flight
      __builtin_amdgcn_sched_group_barrier(0x100, KB < A_AHEAD ? KB : A_AHEAD, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);
        if (kb + A_AHEAD < KB) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
  };
  // MFMAs of (buf, sub) into (n0, n1) interleaved with the vote on the finished block (c0, c1);
  // then the slow path if any lane of the finished block has a candidate
  // (`fresh`, wave-uniform: (c0, c1) hold a block that has not been voted on yet.  A wave that sat tiles out comes back
  // with a block it has already voted on: gating the slow path is enough -- overwriting the 32 accumulators with +inf
  // instead made the register allocator copy them at the loop head, 36 v_mov per wave and tile on the hot path)
  auto segment = [&](const _Float16* tile, int sub, f32x16& n0, f32x16& n1, const f32x16& c0, const f32x16& c1, int ref_base, bool issue,
                     bool fresh) __attribute__((always_inline)) {
    if (issue) mfma_block(tile, sub, n0, n1);
    if (ABL == 3 || ABL == 4 || ABL == 8 || ABL == 9 || ABL == 10) {  // profiling ablation: MFMAs only, accumulators kept live
      asm volatile("" ::"v"(c0[0]), "v"(c0[15]), "v"(c1[0]), "v"(c1[15]));
      return;
    }
    float m0, m1;
    vote(c0, c1, m0, m1);
#ifndef K16_NO_VOTE_PIN
    // The minima are needed only when the block is `fresh`, and where that is not a constant (the first segment of a step)
    // the compiler sinks the whole vote behind the branch on it -- out of the basic block of the MFMAs, i.e. the 8 MFMAs
    // issue back to back and the ~20 VALU instructions of the vote afterwards, instead of two per MFMA slot.  An asm use
    // pins them to this block.
    if (issue) asm volatile("" : "+v"(m0), "+v"(m1));
#endif
    const bool hit = m0 < thrp[0] || m1 < thrp[1];
    if (issue) pipeline_order();
    if (ABL == 1) {  // profiling ablation: distances + minimum, selection removed
      asm volatile("" ::"v"(m0), "v"(m1));
      return;
    }
    if (fresh && __any(hit)) {
      unsigned t0s = tm_now();
      select(c0, c1, m0, m1, ref_base);
      if (ABL == 2) {
        const unsigned d = tm_now() - t0s;
        tm_sel += d;
        tm_mark += d;  // (not counted as segment time)
      }
    }
  };

  // Which step follows `after`: the first one of the pruning window that some wave of the workgroup still
  // needs (union of the masks the waves published before the last barrier; readfirstlane because the compiler
  // cannot see that an LDS value is wave-uniform and would do the mask arithmetic in vector registers).
  // Sets *live to whether THIS wave takes part in it.
  auto next_step = [&](int after, int par_, bool* live) __attribute__((always_inline)) {
    if (!my_lb) {
      *live = true;
      return after + 1;
    }
    unsigned long long any = lds_union(par_) & win_rem;
    if (__builtin_expect(any == 0, 0)) {
      // nothing left in this window for any wave: move on (every wave takes this branch together)
      while (any == 0 && win_base + 64 < n_scan) {
        load_window(win_base + 64);
        my_live = __ballot(win_lb <= wmax + prune_margin);
        if (lane == 0) lds_wlive[2][wave] = my_live;
        __syncthreads();
        any = lds_union(2) & win_rem;
        __syncthreads();
      }
      K16_COLD_REGION_END();
    }
    if (any == 0) {
      *live = false;
      return n_scan;
    }
    const int i = (int)__ffsll((long long)any) - 1;
    *live = (my_live >> i) & 1ull;
    win_rem &= ~((2ull << i) - 1ull);
    return win_base + i;
  };

  // The control flow runs one step ahead of the data: at the top of an iteration the next step (s_next,
  // chosen during the previous iteration) is already known, so its tile loads go out at once and the
  // LDS round trip + scalar work that picks the step after it overlaps with the MFMA segments instead of
  // sitting in front of the loads.
  // one tile of the partial-test pass (EE): the wave's 64 queries against the 64 references of the tile at tile_r
  int n_done = 0;         // tiles this wave has taken part in
  auto ee_tile = [&](const _Float16* tile_r, int t) __attribute__((always_inline)) {
    // Per sub-tile (32 references): K block 0 (2 MFMAs), the partial test, and only a block that passes it goes on: the other
    // K blocks, the vote and -- if some value is below its row's threshold -- the slow path.  No block is left pending across
    // tiles; both A fragments of K block 0 are requested before the first MFMA.
    const f16x8* a8 = reinterpret_cast<const f16x8*>(tile_r) + jq;
    const f16x8 aA = a8[h * K16_TS], aB = a8[32 + h * K16_TS];
    const float tA0 = thrp[0] + ee_c[0], tA1 = thrp[1] + ee_c[1];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int r = 0; r < 16; ++r) accA0[r] = accA1[r] = 0.0f;
      accA0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(sub ? aB : aA, bhi[0][0], accA0, 0, 0, 0);
      accA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(sub ? aB : aA, bhi[1][0], accA1, 0, 0, 0);
      if (__any(min16(accA0) < tA0 || min16(accA1) < tA1)) {
        const f16x8* as = a8 + sub * 32;
#pragma unroll
        for (int kb = 1; kb < KB; ++kb) {
          const f16x8 ahi = as[(kb * 2 + h) * K16_TS];
          accA0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[0][kb], accA0, 0, 0, 0);
          accA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[1][kb], accA1, 0, 0, 0);
        }
        const float m0 = min16(accA0), m1 = min16(accA1);
        if (__any(m0 < thrp[0] || m1 < thrp[1])) {
          const unsigned t0s = tm_now();
          select(accA0, accA1, m0, m1, t * K16_TS + 32 * sub + 4 * h);
          if (ABL == 2) {
            const unsigned dsel = tm_now() - t0s;
            tm_sel += dsel;
            tm_mark += dsel;  // (not counted as segment time)
          }
        }
        ++st_go;
      }
    }
    ++n_done;
  };
  int s_cur = 0;
  int par = 0;
  int it = 0;             // tiles the workgroup has staged so far
  bool live_cur = true;   // does this wave take part in the current tile
  bool pend = false;      // accB holds a block that has not been voted on yet
  unsigned e_pref = 0u;   // LIST: the entry of the step after s_next
  int t_cur = tile_of(0);
  bool live_next = true;
  int s_next = next_step(0, 0, &live_next);
  int t_next = s_next < n_scan ? tile_of(s_next) : t_cur;
  if constexpr (LIST) {
    const unsigned e0 = list_entry(0), e1 = list_entry(1);
    e_pref = list_entry(2);
    t_cur = (int)(e0 & 0xFFFFFFu);
    live_cur = entry_live(e0);
    s_next = 1;
    t_next = s_next < n_scan ? (int)(e1 & 0xFFFFFFu) : t_cur;
    live_next = entry_live(e1);
  }
  // tiles of steps 0 and s_next requested; the first one has to be there
  if constexpr (!PAIR) {
    if (s_next < n_scan && ABL != 9) {
      K16_LOAD(__builtin_amdgcn_readfirstlane(ABL == 8 ? (t_next & 63) : t_next), (unsigned)TILE_LDS_BYTES);
      K16_STAGED_BUT_LAST();
    } else {
      K16_STAGED();
    }
    K16_TILE_BARRIER();
  }
  // ring positions (byte offsets, wave-uniform): the tile being read, the next one (requested), the one to request
  unsigned rd_b = 0u, nx_b = (unsigned)TILE_LDS_BYTES, wr_b = 2u * (unsigned)TILE_LDS_BYTES;
  auto scan_step = [&]() __attribute__((always_inline)) {
    const int t = t_cur;
    const _Float16* tile_r = reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(lds_ring) + rd_b);
    const bool tm_live = live_cur;
    unsigned e_far = 0u;  // LIST: the entry two steps past s_next, requested a whole iteration before it is needed
    if constexpr (LIST) e_far = list_entry(s_next + 2);
#ifdef K16_LIST_LOAD_TOP
    // LIST: the step after next is known here, so its tile is requested before the MFMA segments (the buffer it goes to was
    // read by the previous iteration and every wave has passed the barrier behind that)
    int s_nn_top = 0, t_nn_top = 0;
    if constexpr (LIST) {
      s_nn_top = min(s_next + 1, n_scan);
      t_nn_top = s_nn_top < n_scan ? (int)(e_pref & 0xFFFFFFu) : t_next;
      if (s_nn_top < n_scan) K16_LOAD(__builtin_amdgcn_readfirstlane(t_nn_top), wr_b);
    }
#endif
    if (EE && live_cur) {
      ee_tile(tile_r, t);
    } else if (live_cur) {
#ifdef K16_SETPRIO
      __builtin_amdgcn_s_setprio(K16_SETPRIO);
#endif
      // sub-tile 0 on the pipe while sub-tile 1 of the previous tile is voted on
      segment(tile_r, 0, accA0, accA1, accB0, accB1, refB, true, pend);
      // sub-tile 1 on the pipe while sub-tile 0 is voted on
      segment(tile_r, 1, accB0, accB1, accA0, accA1, t * K16_TS + 4 * h, true, true);
#ifdef K16_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      refB = t * K16_TS + 32 + 4 * h;
      pend = true;
      ++n_done;
    } else if (pend) {
      // this wave sits the tile out (pruned for its 64 queries): only the vote it still owes
      segment(nullptr, 0, accA0, accA1, accB0, accB1, refB, false, true);
      pend = false;
    }

    if (tm_live) tm_lap(tm_seg); else tm_lap(tm_idle);
    if (batch_every > 0 && (it & (batch_every - 1)) == batch_every - 1) {
      // (Placed here, in front of the staging store whose vmcnt(0) wait follows anyway: a cold region with
      // memory operations in front of a pipeline segment makes the compiler wait for ALL outstanding loads at
      // the join -- the tile loads just issued -- on every iteration.)
      // Batched compaction, at the same step in every wave of the workgroup (the waves meet at a
      // barrier per tile, so a compaction at a random moment in one wave stalls all four; done
      // together the stalls overlap): every row that has gathered more than batch_slack entries beyond
      // ksel.  Fresher thresholds also mean fewer candidates that cannot survive.
      const int limit = ksel + K16_COLD(batch_slack);
      unsigned long long todo = 0;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int cg = g ? cnt[1] : cnt[0];
        const int tot = cg + __shfl_xor(cg, 32, 64);
        todo |= (__ballot(tot > limit) & 0xffffffffull) << (32 * g);
      }
      if (todo) {
        while (todo) {
          const int j = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          squeeze(j >> 5, j & 31);
          if (ABL == 2) ++st_sq;
        }
        refresh_wmax();
        K16_COLD_REGION_END();
      }
    }

    // the step after s_next (masks published before the last barrier; the window may move on here)
    bool live_nn = true;
    int s_nn, t_nn;
    if constexpr (LIST) {
      s_nn = min(s_next + 1, n_scan);
      t_nn = s_nn < n_scan ? (int)(e_pref & 0xFFFFFFu) : t_next;
      live_nn = entry_live(e_pref);
      e_pref = e_far;
    } else {
      s_nn = s_next < n_scan ? next_step(s_next, par, &live_nn) : n_scan;
      t_nn = s_nn < n_scan ? tile_of(s_nn) : t_next;
    }

    // request the tile of the step after next (into the buffer the previous iteration read: every wave has passed
    // the barrier behind it), then wait for the tile of the next step, requested one iteration ago
    if (ABL != 9) {  // (8 / 9 = timing-only ablations: tiles from a 64-tile hot set / no tile loads)
      if (s_nn < n_scan) {
#ifdef K16_LIST_LOAD_TOP
        if constexpr (!LIST)
#endif
        K16_LOAD(__builtin_amdgcn_readfirstlane(ABL == 8 ? (t_nn & 63) : t_nn), wr_b);
        tm_lap(tm_tail);
        K16_STAGED_BUT_LAST();
      } else {
        tm_lap(tm_tail);
        K16_STAGED();
      }
      tm_lap(tm_dma);
    }
    if (my_lb) {
      my_live = __ballot(win_lb <= wmax + prune_margin);
      if (lane == 0) lds_wlive[par ^ 1][wave] = my_live;
    }
    tm_lap(tm_tail);
    if (ABL != 10 && (ABL != 4 || (s_cur & 1))) K16_TILE_BARRIER();  // (4 / 10 = timing-only ablations: MFMAs only, a barrier every other tile / none)
    tm_lap(tm_bar);
    s_cur = s_next;
    t_cur = t_next;
    live_cur = live_next;
    s_next = s_nn;
    t_next = t_nn;
    live_next = live_nn;
    par ^= 1;
    ++it;
    const unsigned free_b = rd_b;
    rd_b = nx_b;
    nx_b = wr_b;
    wr_b = free_b;
  };
  tm_mark = tm_now();
  if constexpr (PAIR) {
    // Two list entries per step: buffer (p & 1) holds the tiles of pair p, the copies of pair p + 1 go into the other one at the
    // top of the step (every wave has passed the barrier behind the step that read it) and have the step to land; entries are
    // read two pairs ahead (scalar loads).  A wave waits for ALL its copies at the end of a step (vmcnt(0): nothing to count).
    constexpr unsigned PB = 2u * (unsigned)TILE_LDS_BYTES;
    const int n_pairs = (n_scan + 1) >> 1;
    unsigned c0 = list_entry(0), c1 = list_entry(1), x0 = list_entry(2), x1 = list_entry(3);
    if (n_scan > 0) K16_LOAD(__builtin_amdgcn_readfirstlane((int)(c0 & 0xFFFFFFu)), 0u);
    if (n_scan > 1) K16_LOAD(__builtin_amdgcn_readfirstlane((int)(c1 & 0xFFFFFFu)), (unsigned)TILE_LDS_BYTES);
    K16_STAGED();
    K16_TILE_BARRIER();
    for (int p = 0; p < n_pairs; ++p) {
      const unsigned buf = (p & 1) ? PB : 0u, other = PB - buf;
      const unsigned f0 = list_entry(2 * p + 4), f1 = list_entry(2 * p + 5);
      if (2 * p + 2 < n_scan) K16_LOAD(__builtin_amdgcn_readfirstlane((int)(x0 & 0xFFFFFFu)), other);
      if (2 * p + 3 < n_scan) K16_LOAD(__builtin_amdgcn_readfirstlane((int)(x1 & 0xFFFFFFu)), other + (unsigned)TILE_LDS_BYTES);
      tm_lap(tm_tail);  // (MELD_KNN16_STATS: the copy requests of the next pair)
      const char* ring = reinterpret_cast<const char*>(lds_ring);
      const bool live0 = entry_live(c0), live1 = 2 * p + 1 < n_scan && entry_live(c1);
      if (live0) ee_tile(reinterpret_cast<const _Float16*>(ring + buf), (int)(c0 & 0xFFFFFFu));
      if (live1) ee_tile(reinterpret_cast<const _Float16*>(ring + buf + TILE_LDS_BYTES), (int)(c1 & 0xFFFFFFu));
      if (live0 || live1) tm_lap(tm_seg); else tm_lap(tm_idle);
      if (batch_every > 0 && (p & (batch_every / 2 - 1)) == batch_every / 2 - 1) {  // (batched compaction: as in the single-tile loop)
        const int limit = ksel + K16_COLD(batch_slack);
        unsigned long long todo = 0;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int cg = g ? cnt[1] : cnt[0];
          const int tot = cg + __shfl_xor(cg, 32, 64);
          todo |= (__ballot(tot > limit) & 0xffffffffull) << (32 * g);
        }
        while (todo) {
          const int j = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          squeeze(j >> 5, j & 31);
          if (ABL == 2) ++st_sq;
        }
      }
      tm_lap(tm_tail);
      K16_STAGED();
      tm_lap(tm_dma);
      K16_TILE_BARRIER();
      tm_lap(tm_bar);
      c0 = x0;
      c1 = x1;
      x0 = f0;
      x1 = f1;
      ++it;
    }
  }
  while (!PAIR && s_cur < n_scan) scan_step();
  if (pend) segment(nullptr, 0, accA0, accA1, accB0, accB1, refB, false, true);  // drain: sub-tile 1 of the last tile
  {
    unsigned long long* tiles_done = K16_COLD(tiles_done);
    if (tiles_done && lane == 0) {
      atomicAdd(tiles_done, (unsigned long long)n_done);
      if (EE && K16_COLD(count_go)) atomicAdd(tiles_done + 1, (unsigned long long)st_go);  // (blocks of 32 references that went on past K block 0)
    }
  }

  unsigned long long* stats = ABL == 2 ? K16_COLD(stats) : nullptr;
  if (ABL == 2 && stats) {  // profiling counters requested (MELD_KNN16_STATS): wave-blocks, slow-path entries, appends, compactions
    unsigned a = st_app;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
    if (lane == 0) {
      atomicAdd(stats + 0, (unsigned long long)(2 * n_done));
      atomicAdd(stats + 1, (unsigned long long)st_slow);
      atomicAdd(stats + 2, (unsigned long long)a);
      atomicAdd(stats + 3, (unsigned long long)st_sq);
      const int it_tiles = PAIR ? n_scan : it;  // (the two-tile loop counts pairs)
      if (wave == 0) atomicAdd(stats + 4, (unsigned long long)it_tiles);
      atomicAdd(stats + 5, (unsigned long long)tm_seg);
      atomicAdd(stats + 6, (unsigned long long)tm_sel);
      atomicAdd(stats + 7, (unsigned long long)tm_tail);
      atomicAdd(stats + 8, (unsigned long long)tm_dma);
      atomicAdd(stats + 9, (unsigned long long)tm_bar);
      atomicAdd(stats + 10, (unsigned long long)tm_idle);
      atomicAdd(stats + 11, (unsigned long long)it_tiles);
      if (EE) atomicAdd(stats + 12, (unsigned long long)st_go);
    }
  }
  // final: sort every row, convert back to input units, publish its length and its threshold (the row
  // holds every reference whose approximate d2 is below it)
  const float out_scale = K16_COLD(scale_info)[1];  // 1 / s^2
  float* cand_thr = K16_COLD(cand_thr);
  int* cand_cnt = K16_COLD(cand_cnt);
  if (cand_thr && h == 0) {
    cand_thr[row_base + jq] = (thrp[0] + wave_qn[jq]) * out_scale;
    cand_thr[row_base + 32 + jq] = (thrp[1] + wave_qn[32 + jq]) * out_scale;
  }
  // The rows leave the kernel as they are (raw values, two half-rows each); knn16_finish_rows_kernel, launched right
  // behind, sorts them.  Ranking them here -- 64 rows per wave one after the other, at three waves per SIMD and
  // with every workgroup reaching its epilogue at about the same time -- took 4.8 of the search's 51 ms at 1M cells.
  // cand_cnt carries the two half-row lengths to it.
  {
    const int o0 = __shfl_xor(cnt[0], 32, 64), o1 = __shfl_xor(cnt[1], 32, 64);
    if (h == 0) {
      cand_cnt[row_base + jq] = cnt[0] | (o0 << 16);
      cand_cnt[row_base + 32 + jq] = cnt[1] | (o1 << 16);
    }
  }
#undef K16_LOAD
#undef K16_DMA
#undef K16_STAGED
#undef K16_STAGED_BUT_LAST
#undef K16_TILE_BARRIER
#undef K16_ROUND_OK
}

// ---- the partial-distance test as a pass of its own (round 6) ------------------------------------------------------------
// knn16_partial_filter_kernel: the step lists of the first pass, thinned by the distances themselves over the coordinates of K
// block 0.  For every list entry (tile | waves that cannot rule the tile out << 24) of a query block, every listed wave
// multiplies K block 0 of its 64 queries with K block 0 of the tile's 64 references (4 MFMAs) and keeps its bit only if some
// partial value lies below its row's START threshold + |q_hiB|^2 (+ the rounding allowance): the rule of the EE search kernel
// above, against thresholds that never rise during the search -- so a (wave, tile) pair dropped here could not have produced a
// candidate, and the search over the thinned lists appends exactly what the search over the full ones would.  What it buys: in
// the cells' principal frame 80 % of the pairs fall away, and the search stages whole tiles (7 KiB at d = 50) for the remaining
// fifth only; this pass stages 2 KiB per tile (the two hi planes of K block 0), has no selection, no slow path and no
// data-dependent control -- eight tiles per barrier.  (The one-kernel form -- test and go on inside the search kernel -- staged
// all K blocks of every listed tile for the 18 % of the blocks that went on: 94 GB per launch at 1M cells, at the fabric's
// copy rate.  Staging K block 0 only and letting a block that goes on fetch the rest by itself was built first: 16.7 and, with
// the fetch deferred across the barrier, 19.5 ms against 14.9 -- the blocks that go on come in bursts, their fetch latency is
// exposed to all four waves at the next barrier.)
// Entries are rewritten IN PLACE (an entry is written at or in front of where it was read; no list position is read twice),
// emptied entries dropped, cnt_out = the new length.  One workgroup per query block = the search kernel's four waves.
// (eight tiles per step at four waves per SIMD; 4 tiles x 6 waves, 6 x 5, 2 x 8 measured 5.8 ... 6.9 against 5.4-5.8 ms: the SIMDs are
// busy issuing -- an LDS-DMA piece costs ~180 cycles of issue, the test ~300 per (wave, tile) -- not waiting; the pieces through
// registers instead (a 16-byte load at the top of the step, a ds_write behind the tests, __syncthreads): 5.7-5.8 against 5.4-5.5;
// no staging at all -- every wave walking the list on its own, its A fragments by 16-byte loads straight from global memory, four
// live entries and eight loads in flight, no barrier before the rewrite: 7.4 ms at four waves per SIMD, 7.9-8.1 at six; one copy
// request behind every other tile's test instead of the four of a wave back to back at the top of the step: 5.6 against 5.6)
template <int TPS, int WPE>  // tiles per step (and barrier); waves per SIMD the kernel is compiled for
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void knn16_partial_filter_kernel(const _Float16* __restrict__ Q16, const float* __restrict__ Qn,
                                                                   const _Float16* __restrict__ Rt16, const float* __restrict__ scale_info,
                                                                   const float* __restrict__ norm2_max, const float* __restrict__ thr_init,
                                                                   unsigned* step_list, const int* __restrict__ step_cnt, long long list_stride,
                                                                   int* __restrict__ cnt_out, unsigned long long* __restrict__ tested,
                                                                   const int* __restrict__ block_order, int KB, int ee_hi, int abl) {
  // (abl: timing-only ablations of -DK16_PROFILING builds, MELD_KNN_FILTER_ABL: 1 = no staging behind the first step, 2 = no tests)
  static_assert(TPS >= 2 && TPS <= 32 && (TPS & 1) == 0, "tiles per step");
  __shared__ __attribute__((aligned(16))) _Float16 ring[2][TPS][2 * K16_TS * 8];  // K block 0, hi planes: [k-half][ref][8 halves]
  __shared__ unsigned wmask[2][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jq = lane & 31, h = lane >> 5;
  const int bx = block_order ? __builtin_amdgcn_readfirstlane(block_order[blockIdx.x]) : (int)blockIdx.x;
  const int q_base = bx * K16_BQ + wave * 64;
  // K block 0 of both query groups, and the most the other K blocks can take away from an accumulator (as in the EE kernel)
  f16x8 b0[2];
  float tA[2];
  {
    const float s_ = scale_info[0];
    const float delta = 3.814697265625e-06f * (norm2_max[0] * s_ * s_) + 1e-30f;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int q = q_base + g * 32 + jq;
      const f16x8* qrow = reinterpret_cast<const f16x8*>(Q16 + (size_t)q * (size_t)(KB * 32));
      b0[g] = qrow[h * 2];
      float qb = 0.0f;
      for (int kb = 1; kb < KB; ++kb) {
        const f16x8 v8 = qrow[(kb * 2 + h) * 2];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = (16 * kb + 8 * h + e < ee_hi) ? (float)v8[e] : 0.0f;
          qb = fmaf(v, v, qb);
        }
      }
      qb += __shfl_xor(qb, 32, 64);
      tA[g] = (thr_init[q] - Qn[q]) + (qb * 1.0001f + delta);
    }
  }
  unsigned* const my_list_w = step_list + (size_t)bx * (size_t)list_stride;
  const int n = __builtin_amdgcn_readfirstlane(step_cnt[bx]);
  const int n_steps = (n + TPS - 1) / TPS;
  // K block 0 of a tile = its first two hi planes (tile = KB x [k-half][plane][ref][8 halves]: bytes 0 and 2048).  (The same planes
  // gathered into a contiguous array of their own first, 2 KiB per tile: 5.48 against 5.46 ms -- the pass is not short of
  // bandwidth: without its staging it takes 5.2 ms, without its arithmetic 2.7, with neither 1.0.)
  const size_t tile_bytes = (size_t)KB * 4096;
  const int plane_stride = 2048;
  const size_t src_base = reinterpret_cast<size_t>(Rt16);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const unsigned ring_base = (unsigned)(size_t)(lds_ptr_t)&ring[0][0][0];
  const int lane16 = lane * 16;
  // The entries of a step ride in ONE register: lane j < TPS holds entry j (0 past the end of the list: no wave listed), loaded
  // two steps ahead of their use by one vector load; a tile's entry reaches the scalar side by v_readlane.  (First version: a
  // scalar load and its wait per tile -- 1.5 of the pass's 5.4 ms at 1M cells were spent waiting for list entries.)
  auto load_entries = [&](int step) __attribute__((always_inline)) {
    const int idx = step * TPS + lane;
    return (lane < TPS && idx < n) ? my_list_w[idx] : 0u;
  };
  // staging: 2 TPS pieces of 1 KiB per step, TPS / 2 per wave: piece k = plane (k & 1) of the step's tile (k >> 1)
  auto stage = [&](unsigned ev, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < TPS / 2; ++u) {
      const int k = wave * (TPS / 2) + u;
      const unsigned e = (unsigned)__builtin_amdgcn_readlane((int)ev, k >> 1);
      if (e >> 24) {
        const size_t src = src_base + (size_t)(e & 0xFFFFFFu) * tile_bytes;
        const i32x4 rsrc = {(int)(unsigned)src, (int)((src >> 32) & 0xffffu), 4096, 0x00020000};
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane(
                         (int)(ring_base + (unsigned)((buf * TPS + (k >> 1)) * 2048 + (k & 1) * 1024)))),
                     "v"(lane16), "s"(rsrc), "s"((k & 1) * plane_stride)
                     : "memory");
      }
    }
  };
  int out_n = 0;  // (wave 0: entries written so far)
  unsigned long long n_tested = 0;
  // the masks of a step -> the list, by wave 0 at the top of the next step (lane j: entry j of the step, still in its register)
  auto emit = [&](unsigned ev, int par) __attribute__((always_inline)) {
    unsigned nm = 0u;
#pragma unroll
    for (int w = 0; w < 4; ++w) nm |= ((wmask[par][w] >> lane) & 1u) << w;
    nm &= (ev >> 24);
    const bool keep = nm != 0u;
    const unsigned long long b = __ballot(keep);
    if (keep) my_list_w[out_n + __popcll(b & ((1ull << lane) - 1ull))] = (ev & 0xFFFFFFu) | (nm << 24);
    out_n += __popcll(b);
  };
  unsigned ev_prev = 0u, ev_cur = load_entries(0), ev_nxt = load_entries(1);
  if (n_steps > 0) stage(ev_cur, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int s = 0; s < n_steps; ++s) {
    const int buf = s & 1;
    const unsigned ev_nn = load_entries(s + 2);
    if (s + 1 < n_steps && !(abl & 1)) stage(ev_nxt, buf ^ 1);
    if (wave == 0 && s > 0) emit(ev_prev, buf ^ 1);
    unsigned mybits = 0u;
#pragma unroll
    for (int j = 0; j < TPS; ++j) {
      const unsigned e = (unsigned)__builtin_amdgcn_readlane((int)ev_cur, j);
      if (((e >> (24 + wave)) & 1u) && !(abl & 2)) {
        const f16x8* a8 = reinterpret_cast<const f16x8*>(&ring[buf][j][0]) + jq;
        const f16x8 aA = a8[h * K16_TS], aB = a8[h * K16_TS + 32];
        // (four independent accumulators; one pair after the other -- 55 instead of 70 registers, up to eight waves per SIMD --
        // measured 5.9 against 5.4 ms: the pass is not short of waves)
        f32x16 c00, c01, c10, c11;
#pragma unroll
        for (int r = 0; r < 16; ++r) c00[r] = c01[r] = c10[r] = c11[r] = 0.0f;
        c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aA, b0[0], c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aA, b0[1], c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aB, b0[0], c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aB, b0[1], c11, 0, 0, 0);
        const bool hit = __builtin_fminf(min16(c00), min16(c10)) < tA[0] || __builtin_fminf(min16(c01), min16(c11)) < tA[1];
        if (__any(hit)) mybits |= 1u << j;
        ++n_tested;
      }
    }
    if (lane == 0) wmask[buf][wave] = mybits;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // (this wave's copies of the next step have landed)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    ev_prev = ev_cur;
    ev_cur = ev_nxt;
    ev_nxt = ev_nn;
  }
  if (wave == 0) {
    if (n_steps > 0) emit(ev_prev, (n_steps - 1) & 1);
    if (lane == 0) cnt_out[bx] = out_n;
  }
  if (tested && lane == 0) atomicAdd(tested, n_tested);
}

// Final form of the candidate rows: one wave per row sorts its entries by (value, index) with a bitonic network on
// 64-bit keys (ordered value bits : index; element i = 64 e + lane, so exchanges at distance >= 64 stay inside a
// lane and the others are one __shfl_xor), keeps the ksel smallest, converts them to input units
// ((v + |q|^2) / s^2) and writes the row's length.  cnt: in = half-row lengths n0 | n1 << 16, out = min(n, ksel).
// 1M rows of <= 128 entries: 0.8 ms (the in-kernel ranking it replaces: 4.8 ms).
template <int SL>
__device__ __forceinline__ void k16_bitonic_sort(unsigned long long (&key)[SL], int lane) {
  constexpr int NE = 64 * SL;
#pragma unroll
  for (int k = 2; k <= NE; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {
        const int je = j >> 6;
#pragma unroll
        for (int e = 0; e < SL; ++e) {
          if ((e & je) == 0) {
            const int i = 64 * e + lane;
            const bool asc = (i & k) == 0;
            const unsigned long long a = key[e], b = key[e | je];
            const bool sw = asc ? (b < a) : (a < b);
            key[e] = sw ? b : a;
            key[e | je] = sw ? a : b;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < SL; ++e) {
          const int i = 64 * e + lane;
          const unsigned long long o = __shfl_xor(key[e], j, 64);
          const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
          key[e] = keep_min ? (o < key[e] ? o : key[e]) : (o < key[e] ? key[e] : o);
        }
      }
    }
  }
}
// (the network is as small as the row allows: most rows hold fewer than 128 entries, many fewer than 64)
template <int SL>
__device__ __forceinline__ void k16_finish_row(float* __restrict__ drow, int* __restrict__ irow, int n0, int half, int n_all, int n,
                                               float add, float out_scale, int lane) {
  unsigned long long key[SL];
#pragma unroll
  for (int e = 0; e < SL; ++e) {
    const int i = lane + 64 * e;                        // i-th entry of the packed row: first half-row, then second
    const int p = i < n0 ? i : half + (i - n0);
    key[e] = ~0ull;
    if (i < n_all) key[e] = ((unsigned long long)ordered_bits(drow[p] + 0.0f) << 32) | (unsigned)irow[p];  // (+0: -0 and +0 tie)
  }
  k16_bitonic_sort<SL>(key, lane);
#pragma unroll
  for (int e = 0; e < SL; ++e) {
    const int i = 64 * e + lane;
    if (i < n) {
      drow[i] = (from_ordered_bits((unsigned)(key[e] >> 32)) + add) * out_scale;
      irow[i] = (int)(unsigned)key[e];
    }
  }
}
__global__ __launch_bounds__(256) void knn16_finish_rows_kernel(float* __restrict__ d2, int* __restrict__ idx,
                                                                int* __restrict__ cnt, const float* __restrict__ Qn,
                                                                const float* __restrict__ scale_info, int64_t n_rows,
                                                                int64_t q_pad, int cap, int ksel) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int packed = cnt[row];
  const int n0 = packed & 0xffff, n1 = (packed >> 16) & 0xffff, half = cap >> 1;
  float* drow = d2 + (size_t)row * cap;
  int* irow = idx + (size_t)row * cap;
  const int n_all = n0 + n1, n = min(n_all, ksel);
  const float add = Qn[row % q_pad], out_scale = scale_info[1];
  // (every load of the row happens before its first store: all lanes of the wave pass the sort's shuffles in between)
  if (n_all <= 64)
    k16_finish_row<1>(drow, irow, n0, half, n_all, n, add, out_scale, lane);
  else if (n_all <= 128)
    k16_finish_row<2>(drow, irow, n0, half, n_all, n, add, out_scale, lane);
  else
    k16_finish_row<K16_SLOTS>(drow, irow, n0, half, n_all, n, add, out_scale, lane);
  if (lane == 0) cnt[row] = n;
}

// Merge the per-slice candidate rows of one query (n_slices * q_pad rows of stride cap) into its
// final row: the ksel smallest of the union, sorted by (d2, idx).  One wave per query; at most
// K16_CAPMAX entries in the union.
constexpr int K16_MERGE_MAX = 4096;                 // entries in the union of the slice lists of one query
constexpr int K16_MERGE_SLOTS = K16_MERGE_MAX / 64;  // per lane

__global__ __launch_bounds__(64) void knn16_merge_slices_kernel(const int* __restrict__ s_idx,
                                                                const float* __restrict__ s_d2,
                                                                const int* __restrict__ s_cnt, int q_count, int q_pad,
                                                                int ksel, int cap, int n_slices,
                                                                int* __restrict__ out_idx, float* __restrict__ out_d2,
                                                                int* __restrict__ out_cnt) {
  // (dynamic LDS, n_slices * ksel entries: sized for the largest union it allowed five waves per CU -- 2.3 ms for 300k
  // queries of two slices)
  extern __shared__ __attribute__((aligned(16))) unsigned char merge_lds[];
  float* sd = reinterpret_cast<float*>(merge_lds);
  int* si = reinterpret_cast<int*>(merge_lds) + n_slices * ksel;
  const int lane = threadIdx.x;
  const int q = blockIdx.x;
  if (q >= q_count) return;
  // gather the union into LDS (slice lists are short: <= ksel each)
  int n = 0;
  for (int s = 0; s < n_slices; ++s) {
    const int row = s * q_pad + q;
    const int c = min(s_cnt[row], ksel);
    for (int e = lane; e < c; e += 64) {
      sd[n + e] = s_d2[(size_t)row * cap + e];
      si[n + e] = s_idx[(size_t)row * cap + e];
    }
    n += c;
  }
  __syncthreads();
  // rank every entry by (d2, idx) against the whole union; the ksel smallest go out sorted
  for (int p = lane; p < n; p += 64) {
    const float dp = sd[p];
    const int ip = si[p];
    int rk = 0;
    for (int e = 0; e < n; ++e) rk += (sd[e] < dp || (sd[e] == dp && si[e] < ip)) ? 1 : 0;
    if (rk < ksel) {
      out_d2[(size_t)q * cap + rk] = dp;
      out_idx[(size_t)q * cap + rk] = ip;
    }
  }
  if (lane == 0) out_cnt[q] = min(n, ksel);
}

// ---------------------------------------------------------------------------------------------
// pruning bounds: lb2[b][t] = lower bound on the squared distance (scaled space) between any query of
// workgroup b (K16_BQ consecutive cells) and any reference of tile t (K16_TS consecutive cells):
//     ( min_{p in b} |p - c_t|  -  rho_t )^2        c_t = centroid of tile t, rho_t = max_{r in t} |r - c_t|
// (triangle inequality; holds for any centre as long as rho_t is measured from the same one).  The
// min over the cells of b is taken point by point -- a distance GEMM cells x centroids, 1/64 of the
// search itself, on the same MFMA path with the centroids in the query role -- not from a bounding
// sphere of b: in a 10-d geometry a sphere around 256 cells of a leaf is as wide as the distance to
// the neighbouring leaves and prunes nothing (2 % of the pairs), the pointwise minimum prunes ~35 %.
// ---------------------------------------------------------------------------------------------
// One workgroup per reference tile: centroid (scaled fp32) written as a query-operand row
// ([kb][h][plane][8], 1.0 in K slots d .. d+2), its squared norm, and the tile radius.
__global__ __launch_bounds__(256) void tile_spheres_kernel(const double* __restrict__ X, int64_t N, int d,
                                                           const double* __restrict__ mean,
                                                           const float* __restrict__ scale_info, int KB,
                                                           _Float16* __restrict__ cent16, float* __restrict__ cent_n,
                                                           float* __restrict__ cent_r, int tile0, int dA) {
  // the tile's cells, row stride ld = d | 1 floats (odd: conflict-free column walks; sized by d, not by the largest d the
  // library takes -- 13 KB instead of 37 at d = 50, three times the workgroups per CU of a kernel that is all latency)
  extern __shared__ float xs_dyn[];
  const int ld = d | 1;
#define XS(r, k) xs_dyn[(r) * ld + (k)]
  __shared__ float cs[K16_DMAX + 19];
  const int tid = threadIdx.x;
  const int tile = tile0 + (int)blockIdx.x;  // (a row-sharded build computes the spheres of a range of tiles per rank)
  const int64_t row0 = (int64_t)tile * K16_TS;
  const int cnt = (int)min((int64_t)K16_TS, N - row0);
  const float s = scale_info[0];
  for (int u = tid; u < K16_TS * d; u += 256) {
    const int r = u / d, k = u - r * d;
    const int64_t i = row0 + r;
    XS(r, k) = i < N ? s * (float)(X[i * d + k] - mean[k]) : 0.0f;
  }
  __syncthreads();
  if (tid < KB * 16) {
    float acc = 0.0f;
    if (tid < d)
      for (int r = 0; r < cnt; ++r) acc += XS(r, tid);
    cs[tid] = tid < d ? acc / (float)cnt : 0.0f;
  }
  __syncthreads();
  // The centre the bounds are measured from is any point; the radius is what the pruning test pays for.  Starting at the
  // centroid, a few steps of Badoiu-Clarkson (move 1 / (i + 1) of the way towards the farthest cell) approach the centre of
  // the smallest enclosing ball; the best centre met is kept.  One wave, the cells one per lane.
  __shared__ float cb[K16_DMAX + 19];
  __shared__ float s_best_r2;
  if (tid < 64) {
    const int lane = tid;
    auto far2 = [&](int* who) {
      float r2 = -1.0f;
      if (lane < cnt) {
        // (eight LDS reads in flight: the plain loop waited out one LDS round trip per coordinate, 25 times per tile in ONE wave
        // of the workgroup -- 0.76 ms at 1M cells for a kernel that reads 400 MB; same summation order)
        r2 = 0.0f;
        int k = 0;
        for (; k + 8 <= d; k += 8) {
          float t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = XS(lane, k + u) - cs[k + u];
#pragma unroll
          for (int u = 0; u < 8; ++u) r2 = fmaf(t[u], t[u], r2);
        }
        for (; k < d; ++k) {
          const float t = XS(lane, k) - cs[k];
          r2 = fmaf(t, t, r2);
        }
      }
      // the farthest cell, lowest index among equals: a DPP maximum (in-row butterflies, then lane 15 / 31 handed to the rows
      // above: six vector instructions) and a ballot -- the (value, index) butterfly over ds_bpermute was twelve dependent LDS
      // round trips per step, 25 steps per tile, in the one wave of the workgroup that works here
      auto dppf = [](float old, float v, auto ctrl, auto rows) __attribute__((always_inline)) {
        return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), decltype(ctrl)::value, decltype(rows)::value, 0xF, false));
      };
      using std::integral_constant;
      float m = r2;
      m = fmaxf(m, dppf(m, m, integral_constant<int, 0xB1>{}, integral_constant<int, 0xF>{}));   // quad_perm [1, 0, 3, 2]
      m = fmaxf(m, dppf(m, m, integral_constant<int, 0x4E>{}, integral_constant<int, 0xF>{}));   // quad_perm [2, 3, 0, 1]
      m = fmaxf(m, dppf(m, m, integral_constant<int, 0x141>{}, integral_constant<int, 0xF>{}));  // row_half_mirror
      m = fmaxf(m, dppf(m, m, integral_constant<int, 0x140>{}, integral_constant<int, 0xF>{}));  // row_mirror: every lane holds its row's maximum
      m = fmaxf(m, dppf(m, m, integral_constant<int, 0x142>{}, integral_constant<int, 0xA>{}));  // row_bcast15 -> rows 1 and 3
      m = fmaxf(m, dppf(m, m, integral_constant<int, 0x143>{}, integral_constant<int, 0xC>{}));  // row_bcast31 -> rows 2 and 3
      m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
      *who = __ffsll((long long)__ballot(r2 == m)) - 1;
      return m;
    };
    float best = INFINITY;
#ifndef K16_MEB_STEPS
#define K16_MEB_STEPS 24
#endif
    constexpr int MEB_STEPS = K16_MEB_STEPS;
    for (int it = 1; it <= MEB_STEPS + 1; ++it) {
      int who;
      const float r2 = far2(&who);
      if (r2 < best) {
        best = r2;
        for (int k = lane; k < KB * 16; k += 64) cb[k] = cs[k];
      }
      if (it <= MEB_STEPS) {
        const float wgt = 1.0f / (float)(it + 1);
        for (int k = lane; k < d; k += 64) cs[k] = fmaf(wgt, XS(who, k) - cs[k], cs[k]);
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // (one wave: its LDS operations complete in order)
      asm volatile("" ::: "memory");
    }
    for (int k = lane; k < KB * 16; k += 64) cs[k] = cb[k];
    if (lane == 0) s_best_r2 = best;
  }
  __syncthreads();
  if (tid < 64) {
    const float r2 = s_best_r2;
    float n = 0.0f;
    for (int k = 0; k < d; ++k) n = fmaf(cs[k], cs[k], n);
    if (tid == 0) {
      cent_r[tile] = sqrtf(r2) * 1.0001f + 1e-30f;
      cent_n[tile] = n;
    }
  } else if (tid - 64 < KB * 2) {
    const int g = tid - 64;
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int cc, pc;
      k16_slot(g * 8 + e, d, dA, &cc, &pc);
      _Float16 h16 = (_Float16)0.0f, l16 = (_Float16)0.0f;
      if (cc >= 0) {
        h16 = (_Float16)cs[cc];
        l16 = (_Float16)(cs[cc] - (float)h16);
      } else if (pc >= 0) {
        h16 = (_Float16)1.0f;
      }
      hi[e] = h16;
      lo[e] = l16;
    }
    f16x8* qrow = reinterpret_cast<f16x8*>(cent16) + (size_t)tile * ((size_t)KB * 4);
    qrow[g * 2 + 0] = hi;
    qrow[g * 2 + 1] = lo;
  }
#undef XS
}

// Workgroup = 256 centroids (4 waves x 2 groups of 32, B fragments, hi parts) x a slice of the query
// workgroups; the cells of a query workgroup are its K16_BQ / K16_TS reference tiles (hi planes, staged
// through LDS as in the search kernel).  err_abs covers the hi-only product error of the distances.
//
// SEEDED (thr_seed = the start thresholds of the search, one per query): a second, per-query test.  The
// bound above compares the NEAREST cell of the wave with the LARGEST threshold of the wave -- two different
// cells as a rule.  Thresholds only fall during the search, so query p can never take a candidate from tile
// t if |p - c_t| - rho_t > s_p, s_p = sqrt(seed_p + search-error allowance); the tile is dead for the wave
// if that holds for all its 64 cells:  min_p [ |p - c_t|^2 - (s_p + rho_t)^2 ] > 0, i.e.
// min_p [ acc_pt - s_p^2 - 2 s_p rho_t ] + |c_t|^2 - rho_t^2 > err (two VALU operations per distance on the
// accumulators the first test needs anyway).  Such a tile gets +inf in the table -- the search kernel needs
// no change -- and at 1M x 50 the blocks computed fall from 35 % to 25 % (tools/sim_prune_rules.py).
// BITS (with SEEDED, queries = all the cells): instead of the fp16 table the kernel writes what the step lists need of it, two bits per
// (wave w, tile t):  A[w][t] = the bound is within reach of wave w's largest start threshold and the per-query test does not rule
// the tile out;  B[w][t] = the bound is within reach of WAVE t's largest start threshold.  The symmetrised table entry of (w, t)
// is max(bound(w, t), bound(t, w)), so (w, t) is listed iff A[w][t] and B[t][w]: a bit-matrix transpose and an AND
// (knn16_bits_transpose_and_kernel) replace the table's 488 MB at 1M cells, the 64 x 64-entry symmetrisation pass over it and the
// list builder's pass over it.  wthr[w] = largest seed of wave w + the search-error allowance (-inf: no cells).
// KU < KB (KU = 1, SPLIT operand layout, cells in their principal frame): only the first K block of the operands is staged and
// multiplied -- |p_A - c_A| over its dA coordinates never exceeds |p - c|, and the radius of a tile in the full space is at least its
// radius in that subspace, so (|p_A - c_A| - rho)^2 is a lower bound too (N_A = |p_A|^2 - m_p only lowers it further); a quarter of
// the staging and two of five MFMAs per accumulator.  |c_A|^2 is summed here from the hi + lo parts of the centroid row.
template <int KB, bool SEEDED, int BT, bool BITS = false, int KU = KB>  // BT threads = BT centroids per workgroup
__global__ __launch_bounds__(BT) void knn16_tile_bounds_kernel(const _Float16* __restrict__ C16, const float* __restrict__ Cn,
                                                                const float* __restrict__ Cr,
                                                                const _Float16* __restrict__ Rt16, int n_tiles,
                                                                int first_tile, int n_blocks, float err_coef,
                                                                const float* __restrict__ norm2_max,
                                                                const float* __restrict__ scale_info,
                                                                const float* __restrict__ thr_seed, int64_t n_seed,
                                                                float seed_err_coef, int mark_sign,
                                                                const float* __restrict__ q_norm2, float err_c, float err_l,
                                                                __half* __restrict__ lb2, const float* __restrict__ wthr = nullptr,
                                                                unsigned long long* __restrict__ bits_a = nullptr,
                                                                unsigned long long* __restrict__ bits_b = nullptr, int wpr = 0, int dA = 0) {
  constexpr int TPB = 1;  // one table row per wave of the search kernel: its 64 queries = one reference tile
  constexpr int HV = KU * 2 * K16_TS;   // hi vectors per tile (of the K blocks that are used)
  constexpr int NS = (HV + BT - 1) / BT;
  __shared__ __attribute__((aligned(16))) float4 lds_a[2][HV];
  // SEEDED: the per-query terms ride on the matrix pipe too.  acc_pt - s_p^2 - 2 s_p rho_t = acc_pt + (s_p^2)(-1) + (s_p)(-2 rho_t):
  // one more K block whose slots 0 and 1 hold (s_p^2, s_p) on the cell side and (-1, -2 rho_t) on the centroid side, i.e.
  // one MFMA per accumulator on top of the plain distances instead of three VALU operations and two LDS reads per distance
  // (the kernel was VALU-bound: ~220 vector instructions against 16 MFMAs per wave and tile).  fp16 products are exact in
  // fp32; s_p^2, s_p and rho_t are rounded UP to fp16, which only keeps more tiles alive.
  __shared__ __attribute__((aligned(16))) f16x8 lds_x[2][2 * K16_TS];  // per cell of the tile: (s_p^2, s_p, 0 ...); second half: zeros
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jq = lane & 31, h = lane >> 5;
  const int c_base = blockIdx.x * BT + wave * 64;
  f16x8 bhi[2][KU];
  float cn[2], cr[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const f16x8* qrow = reinterpret_cast<const f16x8*>(C16 + (size_t)(c_base + g * 32 + jq) * (KB * 32));
#pragma unroll
    for (int kb = 0; kb < KU; ++kb) bhi[g][kb] = qrow[(kb * 2 + h) * 2 + 0];
    cn[g] = Cn[c_base + g * 32 + jq];
    cr[g] = Cr[c_base + g * 32 + jq];
    if (KU < KB) {  // |c_A|^2 over the dA coordinates of K block 0 (the half-lanes of a centroid hold slots 0..7 and 8..15)
      const f16x8 lo = qrow[(0 * 2 + h) * 2 + 1];
      float n = 0.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = (8 * h + e < dA) ? (float)bhi[g][0][e] + (float)lo[e] : 0.0f;
        n = fmaf(v, v, n);
      }
      cn[g] = n + __shfl_xor(n, 32, 64);
    }
  }
  float wcol[2] = {-INFINITY, -INFINITY};  // BITS: start threshold (+ allowance) of the wave whose cells are this centroid's tile
  if (BITS) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
      if (c_base + g * 32 + jq < n_tiles) wcol[g] = wthr[c_base + g * 32 + jq];
  }
  // the closing pass of a step: lane l finishes centroid c_base + l (group h, member jq)
  const int c_l = c_base + lane;
  const float cn_l = h == 0 ? cn[0] : cn[1], cr_l = h == 0 ? cr[0] : cr[1], wcol_l = h == 0 ? wcol[0] : wcol[1];
  f16x8 bx[2];  // centroid side of the extra K block: (-1, -2 rho_t, 0 ...) in the lower half-lanes
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int e = 0; e < 8; ++e) bx[g][e] = (_Float16)0.0f;
    if (h == 0) {
      bx[g][0] = (_Float16)(-1.0f);
      bx[g][1] = (_Float16)(-__half2float(__float2half_ru(2.0f * cr[g])));
    }
  }
  // (KU < KB: + 2^-19 n_max for |c_A|^2 taken from the hi + lo parts)
  const float err_abs = (err_coef + (KU < KB ? 1.9073486328125e-06f : 0.0f)) * norm2_max[0] * scale_info[0] * scale_info[0];
  const int b_lo = (int)((long long)n_blocks * blockIdx.y / gridDim.y);
  const int b_hi = (int)((long long)n_blocks * (blockIdx.y + 1) / gridDim.y);
  const int n_steps = (b_hi - b_lo) * TPB;
  const float4* R4 = reinterpret_cast<const float4*>(Rt16);
  float4 p[NS];
  float seed_v = 0.0f;
  const float nmax_s = norm2_max[0] * scale_info[0] * scale_info[0];
  const float seed_margin = seed_err_coef * nmax_s;
  float seed_nq = -1.0f;  // |q|^2 of the cell (scaled), when the caller has it: the allowance of ITS row instead of the global one
  auto load = [&](int step) __attribute__((always_inline)) {
    const int t = first_tile + (b_lo * TPB + step);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int j = tid + BT * u;  // j-th hi vector: (kb*2+h)*64 + ref  ->  plane-0 slot of the tile
      if (j < HV) p[u] = t < n_tiles ? R4[(size_t)t * (KB * 256) + ((j >> 6) << 7) + (j & 63)] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (SEEDED && tid < K16_TS) {  // (seeds are numbered from the first query: table row b = queries 64 b .. 64 b + 63)
      const int64_t q = (int64_t)(b_lo * TPB + step) * K16_TS + tid;
      seed_v = q < n_seed ? thr_seed[q] : -1.0f;  // (-1: no cell here, the padding of the last tile)
      seed_nq = (q_norm2 != nullptr && q < n_seed) ? q_norm2[q] : -1.0f;
    }
  };
  auto store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int j = tid + BT * u;
      if (j < HV) lds_a[buf][j] = p[u];
    }
    if (SEEDED && tid < K16_TS) {
      // a padding cell (its |r|^2 is +inf, so is every accumulator) must add nothing to +inf; a cell without
      // a seed (+inf) keeps the tile alive: acc - inf = -inf
      // search-error allowance of the cell's own row (refine's E_i = c_const n_max + c_lin sqrt(n_i n_max), rounded up),
      // the global one without the norms
      const float e_row = seed_nq >= 0.0f ? fminf((err_c * nmax_s + err_l * sqrtf(seed_nq * nmax_s)) * 1.001f, seed_margin) : seed_margin;
      const float sp = seed_v < 0.0f ? 0.0f : sqrtf(seed_v + e_row) * 1.0001f;
      f16x8 xv;
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = (_Float16)0.0f;
      xv[0] = (_Float16)__half2float(__float2half_ru(sp * sp));  // (+inf for a cell without a seed: acc - inf keeps the tile alive)
      xv[1] = (seed_v < 0.0f || !(sp < INFINITY)) ? (_Float16)0.0f : (_Float16)__half2float(__float2half_ru(sp));
      lds_x[buf][tid] = xv;
      xv[0] = xv[1] = (_Float16)0.0f;
      lds_x[buf][K16_TS + tid] = xv;
    }
  };
  if (n_steps <= 0) return;
  load(0);
  store(0);
  __syncthreads();
  float m[2] = {INFINITY, INFINITY};
  float mb[2] = {INFINITY, INFINITY};  // SEEDED: min_p [acc_pt - s_p^2 - 2 s_p rho_t]
  for (int step = 0; step < n_steps; ++step) {
    const int buf = step & 1;
    if (step + 1 < n_steps) load(step + 1);
    const int t = first_tile + (b_lo * TPB + step);
    if (t < n_tiles) {  // (tiles past the end of the references: nothing there)
      const f16x8* a8 = reinterpret_cast<const f16x8*>(lds_a[buf]);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        f32x16 c0, c1;
#pragma unroll
        for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.0f;
#pragma unroll
        for (int kb = 0; kb < KU; ++kb) {
          const f16x8 ahi = a8[(kb * 2 + h) * K16_TS + sub * 32 + jq];
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[0][kb], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[1][kb], c1, 0, 0, 0);
        }
        m[0] = fminf(m[0], min16(c0));
        m[1] = fminf(m[1], min16(c1));
        if (SEEDED) {
          // (the extra block goes on top of the plain accumulators, in place: they are not needed afterwards)
          const f16x8 ax = lds_x[buf][(h == 0 ? 0 : K16_TS) + sub * 32 + jq];  // (upper half-lanes: K slots 8 .. 15, all zero)
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ax, bx[0], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ax, bx[1], c1, 0, 0, 0);
          mb[0] = fminf(mb[0], min16(c0));
          mb[1] = fminf(mb[1], min16(c1));
        }
      }
    }
    if ((step % TPB) == TPB - 1) {
      // One pass for both groups: the half-lanes of a group hold the two halves of the tile's cells, so after the exchange every
      // lane knows both minima -- lanes 0 .. 31 finish group 0 (centroids c_base .. c_base + 31), lanes 32 .. 63 group 1
      // (c_base + 32 .. c_base + 63): lane l = centroid c_base + l.  (Two passes with the upper half-lanes idle were 40 % of the
      // kernel's vector instructions, and the kernel is bound by those.)
      const int b = b_lo + step / TPB;
      const float wrow = BITS ? wthr[b] : 0.0f;
      const float m0 = fminf(m[0], __shfl_xor(m[0], 32, 64)), m1 = fminf(m[1], __shfl_xor(m[1], 32, 64));
      const float v = (h == 0 ? m0 : m1) + cn_l;  // min_p |p - c|^2, approximate
      const float dist = sqrtf(fmaxf(v - err_abs, 0.0f)) * 0.9999f - cr_l;
      // (fp16, rounded towards zero: a smaller bound only prunes less)
      __half out = __float2half_rz(dist > 0.0f ? fminf(dist * dist, 60000.0f) : 0.0f);
      const float outv = __half2float(out);  // (the value the table would hold: the tests below are the table-driven ones)
      bool dead = false;
      if (SEEDED) {
        const float b0 = fminf(mb[0], __shfl_xor(mb[0], 32, 64)), b1 = fminf(mb[1], __shfl_xor(mb[1], 32, 64));
        const float vb = (h == 0 ? b0 : b1) + cn_l - cr_l * cr_l;
        if (vb > 1.01f * err_abs + 1e-6f * (cn_l + cr_l * cr_l)) {  // dead for every query of the wave
          // (+inf; or, when the table is symmetrised afterwards, the bound with its sign bit set: the bound itself is
          // still wanted for the transposed entry)
          out = mark_sign ? __ushort_as_half((unsigned short)(__half_as_ushort(out) | 0x8000u)) : __ushort_as_half((unsigned short)0x7C00);
          dead = true;
        }
        mb[0] = mb[1] = INFINITY;
      }
      m[0] = m[1] = INFINITY;
      if (BITS) {
        const unsigned long long ba = __ballot(c_l < n_tiles && !dead && outv <= wrow);
        const unsigned long long bb = __ballot(c_l < n_tiles && outv <= wcol_l);
        if (lane == 0 && c_base < wpr * 64) {
          bits_a[(size_t)b * wpr + (c_base >> 6)] = ba;
          bits_b[(size_t)b * wpr + (c_base >> 6)] = bb;
        }
      } else {
        if (c_l < n_tiles) lb2[(size_t)b * n_tiles + c_l] = out;
      }
    }
    if (step + 1 < n_steps) store(buf ^ 1);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// operand preparation: centre, scale into [-1, 1], augment, split into fp16 hi / lo planes
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_centered_kernel(const double* __restrict__ X, int64_t total, int d,
                                                              const double* __restrict__ mean,
                                                              float* __restrict__ scale_info) {
  float m = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf((float)(X[i] - mean[i % d])));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(scale_info + 2), __float_as_int(m));
}

// absmax from the columns' extremes: x -> (float)(x - mean_c) is monotone, so max_i |(float)(x_ic - mean_c)| is attained at the
// column's minimum or maximum -- the same bits as absmax_centered_kernel's pass over all of X, from 2 d numbers
__global__ __launch_bounds__(256) void absmax_from_extremes_kernel(const double* __restrict__ mean, const double* __restrict__ col_min,
                                                                   const double* __restrict__ col_max, int d, float* __restrict__ scale_info) {
  float m = 0.0f;
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    m = fmaxf(m, fmaxf(fabsf((float)(col_min[c] - mean[c])), fabsf((float)(col_max[c] - mean[c]))));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(scale_info + 2), __float_as_int(m));
}

// scale_info: [0] = s (multiply centred data by s), [1] = 1/s^2, [2] = absmax (input of this kernel)
__global__ void finish_scale_kernel(float* scale_info) {
  const float a = scale_info[2];
  const float s = (a > 0.0f) ? 1.0f / a : 1.0f;
  scale_info[0] = s;
  scale_info[1] = 1.0f / (s * s);
}

// Operand preparation, one workgroup per 64 rows (a reference tile / 64 queries).  The rows are
// read with coalesced loads into LDS as centred, scaled fp32 values (a thread-per-row version gathers
// 400-byte-strided rows: 3.5 ms at 1M x 50 against 1 ms here); thread (row, part) then converts every
// fourth group of 8 K slots and stores hi / lo as 16-byte vectors.
//   IS_REF: rows [tile * 64, ...) of X -> tile layout [kb][h][plane][ref][8], values -2 x~ and the three
//           fp16 pieces of |x~|^2 in K slots d .. d+2 (hi plane; padding rows: (+inf, 0, 0));
//           norm2[i] = |x_i - mean|^2 (input units), norm2_max = their maximum.
//   else  : query rows q_begin + (rows ? rows[q] : q) -> [q][kb][h][plane][8], values x~ and 1.0 in K slots
//           d .. d+2; Qn[q] = |x~|^2.

template <bool IS_REF>
__global__ __launch_bounds__(256) void prepare16_kernel(const double* __restrict__ X, int64_t N, int d,
                                                        const double* __restrict__ mean,
                                                        const float* __restrict__ scale_info, int KB, int64_t q_begin,
                                                        int64_t n_rows, const int* __restrict__ rows,
                                                        _Float16* __restrict__ out16, float* __restrict__ out_norm,
                                                        float* __restrict__ norm2_max, int dA) {
  __shared__ float xs[K16_TS][K16_DMAX + 4];  // row stride 145 floats: odd, conflict-free column walks
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * K16_TS;
  const float s = scale_info[0];
  for (int u = tid; u < K16_TS * d; u += 256) {
    const int r = u / d, k = u - r * d;
    const int64_t i = row0 + r;
    float v = 0.0f;
    if (IS_REF) {
      if (i < N) v = s * (float)(X[i * d + k] - mean[k]);
    } else {
      const int64_t qq = i < n_rows ? i : n_rows - 1;  // padding queries repeat the last one
      const int64_t src = q_begin + (rows ? (int64_t)rows[qq] : qq);
      v = s * (float)(X[src * d + k] - mean[k]);
    }
    xs[r][k] = v;
  }
  __syncthreads();
  const int r = tid & 63, part = tid >> 6;
  const int64_t i = row0 + r;
  const bool real = !IS_REF || i < N;
  float n = 0.0f;
  for (int k = 0; k < d; ++k) n = fmaf(xs[r][k], xs[r][k], n);
  if (!real) n = INFINITY;  // padding references are infinitely far
  // |r|^2 = n1 + n2 + n3 (fp16 pieces; residual <= 2^-33 n, or 2^-25 absolute once n3 is subnormal);
  // a padding row is (+inf, 0, 0): inf * 1.0 accumulates to +inf, which never passes `< thr`
  // SPLIT layout (dA > 0): the norm rides in two groups of pieces, N_A = |r_A|^2 - m_r in K block 0 and N_B = |r_B|^2 + m_r behind
  // the other coordinates (r_A = the first dA coordinates).  With m_r >= (|fp16(r_B)|^2 - |r_B|^2)^+ the K blocks behind the
  // first contribute N_B - 2 q_hi.r_hi >= |r_hiB|^2 - 2 q_hiB.r_hiB >= -|q_hiB|^2 to an accumulator whatever r is, which is what
  // the partial test of the list-driven first pass rests on: fp16 rounding is 2^-11 relative (|fp16(x)|^2 <= x^2 (1 + 2^-10 +
  // 2^-22)) or 2^-25 absolute below the normal range (<= 2^-24 |x| + 2^-50 on the square, |x| <= 1), the fp32 sums carry
  // d 2^-24 relative: m_r = 1.03 2^-10 |r_B|^2 + d 2^-22.  The two sums differ from the one chain by <= 2^-22 n (budgeted in
  // k16_const_coef).  PLAIN layout: N_B = |r|^2, no N_A.
  float nA = 0.0f, nB = n;
  if (IS_REF && dA > 0) {
    nB = 0.0f;
    for (int k = 0; k < dA; ++k) nA = fmaf(xs[r][k], xs[r][k], nA);
    for (int k = dA; k < d; ++k) nB = fmaf(xs[r][k], xs[r][k], nB);
    const float m_r = 1.03f * 0.0009765625f * nB + (float)d * 2.384185791015625e-07f;
    nA -= m_r;
    nB += m_r;
    if (!real) {  // (+inf rides in K block 0: a padding reference fails the partial test too)
      nA = INFINITY;
      nB = 0.0f;
    }
  }
  _Float16 npc[6];
  {
    const float src2[2] = {nA, nB};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float nn = src2[q];
      const _Float16 n1 = (_Float16)nn;
      const float r1 = isinf(nn) ? 0.0f : nn - (float)n1;
      const _Float16 n2 = (_Float16)r1;
      npc[3 * q + 0] = n1;
      npc[3 * q + 1] = n2;
      npc[3 * q + 2] = (_Float16)(r1 - (float)n2);
    }
  }
  for (int g = part; g < KB * 2; g += 4) {
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int cc, pc;
      k16_slot(g * 8 + e, d, dA, &cc, &pc);
      _Float16 h16 = (_Float16)0.0f, l16 = (_Float16)0.0f;
      if (cc >= 0) {
        const float v = IS_REF ? -2.0f * xs[r][cc] : xs[r][cc];
        h16 = (_Float16)v;
        l16 = (_Float16)(v - (float)h16);
      } else if (pc >= 0) {
        h16 = IS_REF ? npc[pc] : (_Float16)1.0f;  // exact against 1.0: hi plane only
      }
      hi[e] = h16;
      lo[e] = l16;
    }
    if (IS_REF) {
      f16x8* tile = reinterpret_cast<f16x8*>(out16) + (size_t)blockIdx.x * ((size_t)KB * 2 * 2 * K16_TS);
      tile[(size_t)(g * 2 + 0) * K16_TS + r] = hi;
      tile[(size_t)(g * 2 + 1) * K16_TS + r] = lo;
    } else {
      f16x8* qrow = reinterpret_cast<f16x8*>(out16) + (size_t)i * ((size_t)KB * 4);
      qrow[g * 2 + 0] = hi;
      qrow[g * 2 + 1] = lo;
    }
  }
  if (part == 0) {
    if (IS_REF) {
      float n_orig = 0.0f;
      if (real) {
        n_orig = n * scale_info[1];
        out_norm[i] = n_orig;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) n_orig = fmaxf(n_orig, __shfl_xor(n_orig, off, 64));
      if (r == 0) atomicMax(reinterpret_cast<int*>(norm2_max), __float_as_int(n_orig));
    } else {
      out_norm[i] = n;
    }
  }
}

// Threshold seeds.  The scan starts with every threshold at +inf, so the first tiles of a workgroup -- its own
// 256 cells -- are appended wholesale (256 appends and ~3 compactions per row, a third of what learning the
// thresholds costs the search).  A valid start value is cheap: the (knn+1)-th smallest distance of a cell *within its
// own block of 256* (self included) bounds its bandwidth from above, hence
//     thr_init = rf^2 (A (1 + 1e-5) + 2^-20 n_max) + 1.01 E_row
// bounds the approximate d2 of every reference the kernel radius can reach (A is computed here by direct fp32
// differences of the same centred, scaled coordinates the search uses: relative error ~3e-6 plus the fp32 rounding of
// the inputs, 2^-21 n_max).  One workgroup per block: the 256 cells in LDS, thread i scans them keeping its knn+1
// smallest values in registers.
constexpr int SEED_KMAX = 64;  // largest knn + 1 a register list holds (longer: no seed)
// (rows are zero-padded to SEED_D coordinates -- 60 / 104 / 144: 60 / 104 / 144 KiB of dynamic LDS -- and read as
// broadcast float4, the thread's own row once into registers)
template <int SEED_K, int SEED_D>  // register list length >= knn + 1; padded dimension >= d
__global__ __launch_bounds__(256) void knn16_seed_kernel(const double* __restrict__ X, int64_t N, int d,
                                                         const double* __restrict__ mean,
                                                         const float* __restrict__ scale_info,
                                                         const float* __restrict__ norm2_max, int64_t q_begin,
                                                         int64_t q_count, int knn1, float rf2, float err_c, float err_l,
                                                         float* __restrict__ thr_init) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [256][SEED_D]: rows zero-padded to SEED_D coordinates
  const int tid = threadIdx.x;
  constexpr int ldx = SEED_D;
  const int64_t row0 = q_begin + (int64_t)blockIdx.x * K16_BQ;
  const int n_here = (int)max((int64_t)0, min((int64_t)K16_BQ, q_begin + q_count - row0));
  const float s = scale_info[0];
  for (int u = tid; u < K16_BQ * SEED_D; u += 256) xs[u] = 0.0f;
  __syncthreads();
  for (int u = tid; u < K16_BQ * d; u += 256) {
    const int r = u / d, k = u - r * d;
    if (r < n_here) xs[r * ldx + k] = s * (float)(X[(row0 + r) * d + k] - mean[k]);
  }
  __syncthreads();
  // the SEED_K smallest values seen so far, ascending; inserting v and dropping the largest is one median per slot:
  // new[e] = med3(old[e - 1], v, old[e])
  float best[SEED_K];
#pragma unroll
  for (int e = 0; e < SEED_K; ++e) best[e] = INFINITY;
  // own row in registers (padded coordinates are zero on both sides and add nothing)
  float xr[SEED_D];
#pragma unroll
  for (int k = 0; k < SEED_D; ++k) xr[k] = xs[tid * ldx + k];
  float nq = 0.0f;
#pragma unroll
  for (int k = 0; k < SEED_D; ++k) nq = fmaf(xr[k], xr[k], nq);
  for (int j = 0; j < n_here; ++j) {
    const float4* xj = reinterpret_cast<const float4*>(xs + j * ldx);  // (same address in every lane: LDS broadcast)
    float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
    for (int k4 = 0; k4 < SEED_D / 4; ++k4) {
      const float4 c = xj[k4];
      const float t0 = xr[4 * k4] - c.x, t1 = xr[4 * k4 + 1] - c.y, t2 = xr[4 * k4 + 2] - c.z, t3 = xr[4 * k4 + 3] - c.w;
      acc0 = fmaf(t0, t0, acc0);
      acc1 = fmaf(t1, t1, acc1);
      acc0 = fmaf(t2, t2, acc0);
      acc1 = fmaf(t3, t3, acc1);
    }
    const float acc = acc0 + acc1;
    if (acc < best[SEED_K - 1]) {
#pragma unroll
      for (int e = SEED_K - 1; e > 0; --e) best[e] = __builtin_amdgcn_fmed3f(best[e - 1], acc, best[e]);
      best[0] = fminf(best[0], acc);
    }
  }
  float worst = INFINITY;  // the knn1-th smallest
#pragma unroll
  for (int e = 0; e < SEED_K; ++e)
    if (e == knn1 - 1) worst = best[e];
  if (tid < K16_BQ) {
    float out = INFINITY;
    if (tid < n_here && n_here >= knn1 && worst < INFINITY) {
      const float nmax_s = norm2_max[0] * s * s;
      const float e_row = (err_c * nmax_s + err_l * sqrtf(nq * nmax_s)) * 1.01f;
      out = (rf2 * (worst * 1.00001f + 9.5367431640625e-07f * nmax_s) + e_row) * 1.000001f + 1e-30f;
    }
    thr_init[(int64_t)blockIdx.x * K16_BQ + tid] = out;
  }
}

// The same seeds from a wider neighbourhood on the matrix pipe: a workgroup takes its 256 queries (B fragments, as in
// the search) against its own four reference tiles and the four on either side in index order (768 cells: in locality
// order the next leaves along the chain) -- 192 MFMAs per wave -- and every lane keeps the SEED_K smallest approximate
// d2 of its two queries sorted in registers (one v_med3 per slot and insertion); the two half-lanes of a query then
// merge their lists.  The approximate distance may fall short of the exact one by the row's search-error allowance, so
//     thr_init = rf^2 (A~ + 1.01 E_row + 2^-20 n_max) + 1.01 E_row.
// More cells than the fp32 kernel above looks at (768 vs 256: a tighter bound) for a tenth of its time.
// `side` = tiles on either side of the workgroup's own K16_BQ / K16_TS tiles.  Tighter seeds pay twice since the pruning
// table tests every query against its own seed (meld_knn16_bounds): at 1M x 50, side 4 / 16 / 32 / 64 cost 0.6 / 1.2 /
// 1.9 / 3.3 ms and leave the search at 45.9 / 41.9 / 40.9 / 38.9 ms.  The cost grows with N, the gain with N^2: the
// automatic choice WAS N^2 / 2e10 tiles, between 4 and 64 (500k cells: 12; 1M: 50).  With the window scanned nearest tiles first a
// far tile costs little more than its MFMAs, and wider windows pay at the mid sizes: N / 12500 tiles, between 8 and 64 -- whole
// step at 100k / 200k / 300k / 500k / 750k cells 7.9 -> 7.7 / 10.4 -> 10.2 / 14.8 -> 14.4 / 24.6 -> 23.8 / 34.1 -> 33.7 ms;
// flat from 50 to 128 at 1M and 2M (what the seeds cost more, the search costs less).
// (SEED_WAVES waves = 64 SEED_WAVES queries per workgroup share every staged tile: the kernel is bound by its tile loads and
// barriers, and a workgroup of 8 waves stages 8 + 2 side tiles where two of 4 waves staged 2 (4 + 2 side))
constexpr int SEED_WAVES = 8;
template <int KB, int SEED_K>
__global__ __launch_bounds__(64 * SEED_WAVES) void knn16_seed_mfma_kernel(const _Float16* __restrict__ Q16, const float* __restrict__ Qn,
                                                              const _Float16* __restrict__ Rt16,
                                                              const float* __restrict__ scale_info,
                                                              const float* __restrict__ norm2_max, int n_tiles,
                                                              int first_tile, int side, int n_qwaves, int knn1, float rf2, float err_c, float err_l,
                                                              float* __restrict__ thr_init) {
  constexpr int HV = KB * 2 * K16_TS;  // hi vectors per tile
  constexpr int ST = 64 * SEED_WAVES;
  constexpr int NS = (HV + ST - 1) / ST;
  __shared__ __attribute__((aligned(16))) float4 lds_a[HV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, jq = lane & 31, h = lane >> 5;
  const int q_base = (blockIdx.x * SEED_WAVES + wave) * 64;
  const bool has_q = blockIdx.x * SEED_WAVES + wave < n_qwaves;  // (the last workgroup may reach past the padded query rows)
  f16x8 bhi[2][KB];
  float nq[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const f16x8* qrow = reinterpret_cast<const f16x8*>(Q16 + (size_t)((has_q ? q_base : 0) + g * 32 + jq) * (KB * 32));
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) bhi[g][kb] = qrow[(kb * 2 + h) * 2 + 0];
    nq[g] = Qn[(has_q ? q_base : 0) + g * 32 + jq];
  }
  float best0[SEED_K], best1[SEED_K];
#pragma unroll
  for (int e = 0; e < SEED_K; ++e) best0[e] = best1[e] = INFINITY;
  auto insert = [&](float (&best)[SEED_K], float v) __attribute__((always_inline)) {
    if (v < best[SEED_K - 1]) {
#pragma unroll
      for (int e = SEED_K - 1; e > 0; --e) best[e] = __builtin_amdgcn_fmed3f(best[e - 1], v, best[e]);
      best[0] = fminf(best[0], v);
    }
  };
  const float4* R4 = reinterpret_cast<const float4*>(Rt16);
  const int t_own = first_tile + blockIdx.x * SEED_WAVES;
  // (the next tile is requested before the MFMAs of the current one: loading, storing and computing one after the
  // other left the kernel at a sixth of what its MFMAs need)
  const int t_lo = max(t_own - side, 0), t_hi = min(t_own + SEED_WAVES + side, n_tiles);
  float4 stage[NS];
  auto fetch = [&](int t) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int j = tid + ST * u;  // j-th hi vector: (kb*2+h)*64 + ref  ->  plane-0 slot of the tile
      if (j < HV) stage[u] = R4[(size_t)t * (KB * 256) + ((j >> 6) << 7) + (j & 63)];
    }
  };
  // Nearest tiles first: the workgroup's own tiles, then outwards on both sides in turn.  The k smallest of a set do not depend on
  // the order it is scanned in, the work does: walking the window from one end to the other kept the thresholds loose for half of
  // it, with the distances falling tile after tile -- insertions (one lane's insertion is paid by the wave) in most of them.
  const int own_hi = min(t_own + SEED_WAVES, t_hi);
  int n_below = 0, n_above = 0, t_next;  // tiles taken so far on either side; the tile after the one in hand
  auto advance = [&](int t) __attribute__((always_inline)) {  // (uniform) the tile that follows t in the scan, t_hi when none is left
    if (t >= t_own && t + 1 < own_hi) return t + 1;
    const bool more_below = t_own - 1 - n_below >= t_lo, more_above = own_hi + n_above < t_hi;
    if (more_below && (n_below <= n_above || !more_above)) return t_own - 1 - n_below++;
    if (more_above) return own_hi + n_above++;
    return t_hi;
  };
  t_next = t_own < own_hi ? t_own : advance(t_hi);  // (a workgroup past the last tile has no tiles of its own)
  if (t_next < t_hi) fetch(t_next);
  for (int t = t_next; t < t_hi; t = t_next) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int j = tid + ST * u;
      if (j < HV) lds_a[j] = stage[u];
    }
    __syncthreads();
    t_next = advance(t);
    if (t_next < t_hi) fetch(t_next);
    const f16x8* a8 = reinterpret_cast<const f16x8*>(lds_a);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 c0, c1;
#pragma unroll
      for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.0f;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const f16x8 ahi = a8[(kb * 2 + h) * K16_TS + sub * 32 + jq];
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[0][kb], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[1][kb], c1, 0, 0, 0);
      }
      // (one test per 16 values first: beyond the nearest few tiles no lane has anything to insert)
      if (__any(min16(c0) + nq[0] < best0[SEED_K - 1])) {
#pragma unroll
        for (int r = 0; r < 16; ++r) insert(best0, c0[r] + nq[0]);
      }
      if (__any(min16(c1) + nq[1] < best1[SEED_K - 1])) {
#pragma unroll
        for (int r = 0; r < 16; ++r) insert(best1, c1[r] + nq[1]);
      }
    }
  }
  // the other half-lane holds the other half of each query's references: merge its list
  float other[SEED_K];
#pragma unroll
  for (int e = 0; e < SEED_K; ++e) other[e] = __shfl_xor(best0[e], 32, 64);
#pragma unroll
  for (int e = 0; e < SEED_K; ++e) insert(best0, other[e]);
#pragma unroll
  for (int e = 0; e < SEED_K; ++e) other[e] = __shfl_xor(best1[e], 32, 64);
#pragma unroll
  for (int e = 0; e < SEED_K; ++e) insert(best1, other[e]);
  float a0 = INFINITY, a1 = INFINITY;  // the knn1-th smallest
#pragma unroll
  for (int e = 0; e < SEED_K; ++e) {
    if (e == knn1 - 1) {
      a0 = best0[e];
      a1 = best1[e];
    }
  }
  if (h == 0 && has_q) {
    const float s = scale_info[0];
    const float nmax_s = norm2_max[0] * s * s;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float a = g ? a1 : a0;
      float out = INFINITY;
      if (a < INFINITY) {
        const float e_row = (err_c * nmax_s + err_l * sqrtf(nq[g] * nmax_s)) * 1.01f;
        out = (rf2 * (fmaxf(a, 0.0f) + e_row + 9.5367431640625e-07f * nmax_s) + e_row) * 1.000001f + 1e-30f;
      }
      thr_init[q_base + g * 32 + jq] = out;
    }
  }
}

}  // namespace meld

using namespace meld;

// SPLIT operand layout (k16_slot): wherever it fits.  A function of d alone: no process-wide switch (the boundary's contract is
// "no global state"); the plain layout for A-B measurements is a compile-time choice of a development build (-DK16_PLAIN_LAYOUT).
#ifdef K16_PLAIN_LAYOUT
static int k16_dA(int, int) { return 0; }
#else
static int k16_dA(int d, int KB) { return k16_split_dims_of(d, KB, 1); }
#endif
extern "C" int meld_knn16_kblocks(int d);
// coordinates K block 0 holds under the SPLIT layout (the list-driven first pass tests its accumulators behind that block: the
// caller does well to hand the cells over in a frame whose leading coordinates carry the distances); 0: plain layout
extern "C" int meld_knn16_split_dims(int d) {
  const int kb = meld_knn16_kblocks(d);
  return kb < 0 ? 0 : k16_dA(d, kb);
}
extern "C" int meld_knn16_kblocks(int d) {
  if (d < 1) return MELD_ERR_INVALID;
  const int kb = (d + 3 + 15) / 16;  // d coordinates + three K slots for the reference norm
  if (kb > 9) {
    set_err("meld_knn16_kblocks: d=%d exceeds the largest instantiated distance kernel (d <= 141)", d);
    return MELD_ERR_UNSUPPORTED;
  }
  return kb;
}
// bytes of one reference tile / one query row of the fp16 operand arrays
extern "C" size_t meld_knn16_tile_bytes(int d) {
  const int kb = meld_knn16_kblocks(d);
  return kb < 0 ? 0 : (size_t)kb * 2 * 2 * K16_TS * 16;
}
extern "C" size_t meld_knn16_query_bytes(int d) {
  const int kb = meld_knn16_kblocks(d);
  return kb < 0 ? 0 : (size_t)kb * 64;
}
extern "C" int meld_knn16_tile_refs(void) { return K16_TS; }
extern "C" int meld_knn16_block_queries(void) { return K16_BQ; }
extern "C" int meld_knn16_row_capacity(int ksel) {
  if (ksel < 1 || ksel > K16_CAPMAX - K16_SLACK) {
    set_err("meld_knn16_row_capacity: ksel=%d outside [1, %d]", ksel, K16_CAPMAX - K16_SLACK);
    return MELD_ERR_UNSUPPORTED;
  }
  return ksel + K16_SLACK;
}
// Bound on |d2_approx - d2_exact| / max_i |x~_i|^2 (n_max) that meld_knn_refine budgets for, as a function
// of the dimension (the number of terms an accumulator really sums).  Per result the kernel accumulates
// |r|^2 - 2 q.r, T = nprod * d + 3 non-zero products (zero K slots add exactly), sum of |terms| <= |r|^2 +
// 2 |q||r| <= 3 n_max:
//   fp32 accumulation          <= 1.01 T u * 3 n_max with u = 2^-23 (one ulp per addition: also covers a
//                                 truncating adder tree inside the MFMA)
//   scaled operands in fp32    points are rounded to fp32 before the split: <= 2^-21 n_max on d2
//   |q|^2 and the threshold    fp32 FMA chain over d terms + one subtraction: (d + 2) 2^-24 n_max
//   |r|^2 pieces               three exact fp16 pieces, residual <= 2^-33 n_max + 2^-25 <= 2^-24 n_max
//   nprod = 3 only             operands are hi + lo: |v - hi - lo| <= 2^-22 |v| + 2^-25 (fp16 subnormal floor;
//                              the data are scaled to max |coordinate| = 1, so n_max >= 1): <= (2^-20 + 3 * 2^-25
//                              sqrt(d)) n_max; dropped lo.lo products <= 2^-21 n_max
//   nprod = 1 only             coordinate blocks on the fp16 hi parts: |q.r - qhi.rhi| <= |qlo.r| + |qhi.rlo|
//                              <= 2 * 2^-11 |x~_q| |2 x~_r| = 2^-9 |x~_q||x~_r| (Cauchy-Schwarz) -- the "lin" part,
//                              charged per row through its own norm
// d = 50: 6.1e-5 (nprod 3), 2.3e-5 + 2^-9 |q||r| (nprod 1); d = 2: 5.6e-6 -- what lets a million cells in the plane
// be certified (neighbour distances^2 ~ 1e-5 n_max) instead of going through the exact sweep row by row.
// Measured worst error: 7.6e-7 n_max (nprod 3), 5.4e-4 n_max (nprod 1) at d = 50.
static double k16_const_coef(int nprod, int d) {
  const double T = (double)(nprod == 3 ? 3 * d : d) + 6.0;  // (+ the norm pieces: six under the SPLIT layout)
  double c = 1.01 * T * 1.1920928955078125e-07 * 3.0  // accumulation, u = 2^-23
             + 4.76837158203125e-07                    // 2^-21: fp32 rounding of the scaled points
             + (double)(d + 2) * 5.9604644775390625e-08  // (d + 2) 2^-24
             + 5.9604644775390625e-08                  // 2^-24: norm pieces
             + 2.98023223876953125e-07;                // 2^-22 + 2^-24: SPLIT layout -- |r_A|^2 and |r_B|^2 summed apart, two sets of pieces
  if (nprod == 3) c += 9.5367431640625e-07 + 3.0 * 2.98023223876953125e-08 * sqrt((double)d) + 4.76837158203125e-07;
  return c;
}
extern "C" double meld_knn16_error_coef(int nprod, int d) {
  return k16_const_coef(nprod, d) + (nprod == 1 ? 0.001953125 : 0.0);  // + 2^-9 (global worst case of the lin part)
}
// The same bound split for a per-row allowance  E_i = c_const max|x~|^2 + c_lin |x~_i| max|x~|
// (the nprod = 1 term is |qlo.r| + |qhi.rlo| <= 2^-9 |x~_q| |x~_r|): rows near the centre of
// the data get a tighter allowance than the global worst case.
extern "C" double meld_knn16_error_coef_const(int nprod, int d) { return k16_const_coef(nprod, d); }
extern "C" double meld_knn16_error_coef_lin(int nprod) { return nprod == 1 ? 0.001953125 : 0.0; }

static int k16_prepare_impl(const double* X, int64_t N, int d, const double* mean, int64_t q_begin,
                                  int64_t q_count, void* Rt16, void* Q16, float* Qn, float* norm2, float* norm2_max,
                                  float* scale_info, meld_stream_t stream, const double* col_min, const double* col_max) {
  MELD_CHECK_ARG(X && mean && Rt16 && Q16 && Qn && norm2 && norm2_max && scale_info && N > 0,
                 "meld_knn16_prepare: null/empty argument");
  MELD_CHECK_ARG(q_count > 0 && q_begin >= 0 && q_begin + q_count <= N, "meld_knn16_prepare: bad query range");
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  hipStream_t st = S(stream);
  MELD_HIP_CALL(hipMemsetAsync(scale_info, 0, 4 * sizeof(float), st));
  MELD_HIP_CALL(hipMemsetAsync(norm2_max, 0, sizeof(float), st));
  if (col_min != nullptr && col_max != nullptr)
    hipLaunchKernelGGL(absmax_from_extremes_kernel, dim3(1), dim3(256), 0, st, mean, col_min, col_max, d, scale_info);
  else
    hipLaunchKernelGGL(absmax_centered_kernel, dim3(2048), dim3(256), 0, st, X, N * (int64_t)d, d, mean, scale_info);
  hipLaunchKernelGGL(finish_scale_kernel, dim3(1), dim3(1), 0, st, scale_info);
  const int64_t n_pad = ceil_div(N, K16_TS) * K16_TS;
  hipLaunchKernelGGL((prepare16_kernel<true>), dim3((unsigned)(n_pad / K16_TS)), dim3(256), 0, st, X, N, d, mean, scale_info,
                     KB, (int64_t)0, N, (const int*)nullptr, reinterpret_cast<_Float16*>(Rt16), norm2, norm2_max, k16_dA(d, KB));
  const int64_t q_pad = ceil_div(q_count, K16_BQ) * K16_BQ;
  hipLaunchKernelGGL((prepare16_kernel<false>), dim3((unsigned)(q_pad / K16_TS)), dim3(256), 0, st, X, N, d, mean, scale_info,
                     KB, q_begin, q_count, (const int*)nullptr, reinterpret_cast<_Float16*>(Q16), Qn, (float*)nullptr, k16_dA(d, KB));
  MELD_LAUNCH_CHECK("meld_knn16_prepare");
  return MELD_OK;
}
extern "C" int meld_knn16_prepare(const double* X, int64_t N, int d, const double* mean, int64_t q_begin,
                                  int64_t q_count, void* Rt16, void* Q16, float* Qn, float* norm2, float* norm2_max,
                                  float* scale_info, meld_stream_t stream) {
  return k16_prepare_impl(X, N, d, mean, q_begin, q_count, Rt16, Q16, Qn, norm2, norm2_max, scale_info, stream, nullptr, nullptr);
}
// The same with the scale taken from the columns' minima / maxima (meld_col_stats_f64: one pass over X gives the mean, the
// finite check and these), instead of a pass of its own over all of X; identical operands.
extern "C" int meld_knn16_prepare_scaled(const double* X, int64_t N, int d, const double* mean, const double* col_min, const double* col_max,
                                         int64_t q_begin, int64_t q_count, void* Rt16, void* Q16, float* Qn, float* norm2, float* norm2_max,
                                         float* scale_info, meld_stream_t stream) {
  MELD_CHECK_ARG(col_min && col_max, "meld_knn16_prepare_scaled: null column extremes");
  return k16_prepare_impl(X, N, d, mean, q_begin, q_count, Rt16, Q16, Qn, norm2, norm2_max, scale_info, stream, col_min, col_max);
}

// The same for a search between two point sets (the blocks of graphtools' MNN kernel between two samples): X holds
// n_total rows, the references are its rows [0, n_refs), the queries any range of it (the caller puts the query set
// behind the references).  Scaling over all n_total rows; norm2 / norm2_max cover the references (the caller adds the
// queries' from Qn).
extern "C" int meld_knn16_prepare_cross(const double* X, int64_t n_refs, int64_t n_total, int d, const double* mean,
                                        int64_t q_begin, int64_t q_count, void* Rt16, void* Q16, float* Qn, float* norm2,
                                        float* norm2_max, float* scale_info, meld_stream_t stream) {
  MELD_CHECK_ARG(X && mean && Rt16 && Q16 && Qn && norm2 && norm2_max && scale_info && n_refs > 0 && n_total >= n_refs,
                 "meld_knn16_prepare_cross: null/empty argument");
  MELD_CHECK_ARG(q_count > 0 && q_begin >= 0 && q_begin + q_count <= n_total, "meld_knn16_prepare_cross: bad query range");
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  hipStream_t st = S(stream);
  MELD_HIP_CALL(hipMemsetAsync(scale_info, 0, 4 * sizeof(float), st));
  MELD_HIP_CALL(hipMemsetAsync(norm2_max, 0, sizeof(float), st));
  hipLaunchKernelGGL(absmax_centered_kernel, dim3(2048), dim3(256), 0, st, X, n_total * (int64_t)d, d, mean, scale_info);
  hipLaunchKernelGGL(finish_scale_kernel, dim3(1), dim3(1), 0, st, scale_info);
  const int64_t n_pad = ceil_div(n_refs, K16_TS) * K16_TS;
  hipLaunchKernelGGL((prepare16_kernel<true>), dim3((unsigned)(n_pad / K16_TS)), dim3(256), 0, st, X, n_refs, d, mean, scale_info,
                     KB, (int64_t)0, n_refs, (const int*)nullptr, reinterpret_cast<_Float16*>(Rt16), norm2, norm2_max, k16_dA(d, KB));
  const int64_t q_pad = ceil_div(q_count, K16_BQ) * K16_BQ;
  hipLaunchKernelGGL((prepare16_kernel<false>), dim3((unsigned)(q_pad / K16_TS)), dim3(256), 0, st, X, n_total, d, mean, scale_info,
                     KB, q_begin, q_count, (const int*)nullptr, reinterpret_cast<_Float16*>(Q16), Qn, (float*)nullptr, k16_dA(d, KB));
  MELD_LAUNCH_CHECK("meld_knn16_prepare_cross");
  return MELD_OK;
}

// Query operands for a list of rows (second search stage); scale_info / mean as produced by
// meld_knn16_prepare for the same X.
extern "C" int meld_knn16_prepare_rows(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                                       int64_t q_begin, const int32_t* rows, int64_t n_rows, void* Q16, float* Qn,
                                       meld_stream_t stream) {
  MELD_CHECK_ARG(X && mean && scale_info && rows && Q16 && Qn && n_rows > 0 && N > 0,
                 "meld_knn16_prepare_rows: bad arguments");
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  const int64_t q_pad = ceil_div(n_rows, K16_BQ) * K16_BQ;
  hipLaunchKernelGGL((prepare16_kernel<false>), dim3((unsigned)(q_pad / K16_TS)), dim3(256), 0, S(stream), X, N, d, mean,
                     scale_info, KB, q_begin, n_rows, rows, reinterpret_cast<_Float16*>(Q16), Qn, (float*)nullptr, k16_dA(d, KB));
  MELD_LAUNCH_CHECK("meld_knn16_prepare_rows");
  return MELD_OK;
}

namespace meld {
// Start thresholds of the re-search (scaled units): the first pass found ksel references with approximate d2 <= tau, so the true
// ksel-th distance is <= tau + E1 and its full-precision approximation <= tau + E1 + E3 -- nothing above that can enter the
// list; a row whose first list is shorter than ksel starts at +inf.  Padding rows copy the last row's value.
__global__ __launch_bounds__(256) void knn16_research_thr_kernel(const int* __restrict__ rows, long long n_rows, long long q_pad,
                                                                 long long q_begin, const int* __restrict__ cand_cnt,
                                                                 const float* __restrict__ cand_d2, int cap, int ksel,
                                                                 const float* __restrict__ norm2, const float* __restrict__ nmax,
                                                                 double err_coef, double err_lin, double err3,
                                                                 const float* __restrict__ scale_info, float* __restrict__ thr) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= q_pad) return;
  const long long r = rows[i < n_rows ? i : n_rows - 1];
  const int c = cand_cnt[r];
  const double tau = (double)cand_d2[r * cap + (c > 0 ? c - 1 : 0)];
  const double nmx = (double)nmax[0];
  const double e1 = err_coef * nmx + err_lin * sqrt((double)norm2[q_begin + r] * nmx);
  const double e3 = err3 * nmx;
  const double s = (double)scale_info[0];
  const float b = (float)((tau + e1 + e3) * (s * s) * (1.0 + 1e-5));
  thr[i] = c >= ksel ? b : INFINITY;
}
}  // namespace meld

// thr[q_pad] for meld_knn16_topk(thr_init = thr) over the rows meld_knn16_prepare_rows gathered: see the kernel.
// cand_cnt / cand_d2 (row stride cap): the first pass's lists; err_coef / err_lin: its error coefficients, err3: the
// re-search's (meld_knn16_error_coef(3, d)); norm2 is indexed by global row (q_begin + rows[i]).
extern "C" int meld_knn16_research_thresholds(const int32_t* rows, int64_t n_rows, int64_t q_begin, const int32_t* cand_cnt,
                                              const float* cand_d2, int cap, int ksel, const float* norm2, const float* norm2_max,
                                              double err_coef, double err_lin, double err3, const float* scale_info, float* thr,
                                              meld_stream_t stream) {
  MELD_CHECK_ARG(rows && cand_cnt && cand_d2 && norm2 && norm2_max && scale_info && thr && n_rows > 0 && cap > 0,
                 "meld_knn16_research_thresholds: bad arguments");
  const int64_t q_pad = ceil_div(n_rows, K16_BQ) * K16_BQ;
  knn16_research_thr_kernel<<<(unsigned)ceil_div(q_pad, 256), 256, 0, S(stream)>>>(rows, n_rows, q_pad, q_begin, cand_cnt, cand_d2, cap, ksel,
                                                                                   norm2, norm2_max, err_coef, err_lin, err3, scale_info, thr);
  MELD_LAUNCH_CHECK("meld_knn16_research_thresholds");
  return MELD_OK;
}

namespace meld {
// The bound of (wave w, tile t) is a lower bound on the distance between ANY cell of w and ANY cell of t, and so is the bound
// of (wave t, tile w) -- the cells of tile w against the sphere of tile t, and the other way round: when the queries are all
// the cells, the table is symmetrised to the larger of the two (1M cells: the live blocks of the end state fall from 15.8
// to 13.9 %).  Entries the per-query test marked dead arrive with their sign bit set (the bound itself is still needed by
// the transposed entry) and leave as +inf.  64 x 64 entries per workgroup, pairs (bi, bj) with bi <= bj.
__global__ __launch_bounds__(256) void knn16_bounds_symmetrize_kernel(unsigned short* __restrict__ lb, int n, int ld) {
  __shared__ unsigned short ta[64][66], tb[64][66];
  // linear block index -> (bi, bj), bi <= bj, over the upper triangle of nb x nb blocks
  const int nb = (n + 63) / 64;
  int bi = 0, rem = blockIdx.x;
  while (rem >= nb - bi) {
    rem -= nb - bi;
    ++bi;
  }
  const int bj = bi + rem;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int wa = bi * 64 + r, ca = bj * 64 + tx;  // block A = rows of bi, columns of bj
    ta[r][tx] = (wa < n && ca < n) ? lb[(size_t)wa * ld + ca] : (unsigned short)0;
    const int wb = bj * 64 + r, cb = bi * 64 + tx;  // block B = rows of bj, columns of bi
    tb[r][tx] = (wb < n && cb < n) ? lb[(size_t)wb * ld + cb] : (unsigned short)0;
  }
  __syncthreads();
  auto merge = [](unsigned short mine, unsigned short other) {
    const unsigned short a = mine & 0x7fffu, b = other & 0x7fffu;  // (non-negative halves order like their bits)
    return (mine & 0x8000u) ? (unsigned short)0x7C00 : (a > b ? a : b);
  };
  for (int r = ty; r < 64; r += 4) {
    const int wa = bi * 64 + r, ca = bj * 64 + tx;
    if (wa < n && ca < n) lb[(size_t)wa * ld + ca] = merge(ta[r][tx], tb[tx][r]);
    if (bi != bj) {
      const int wb = bj * 64 + r, cb = bi * 64 + tx;
      if (wb < n && cb < n) lb[(size_t)wb * ld + cb] = merge(tb[r][tx], ta[tx][r]);
    }
  }
}
}  // namespace meld

extern "C" size_t meld_knn16_bounds_bytes(int64_t n_ref, int64_t q_count) {
  return (size_t)ceil_div(q_count, K16_BQ) * K16_NWAVE * (size_t)ceil_div(n_ref, K16_TS) * sizeof(__half);
}
extern "C" size_t meld_knn16_bounds_temp_bytes(int64_t n_ref, int d, int64_t q_count) {
  (void)q_count;
  const int kb = meld_knn16_kblocks(d);
  if (kb < 0) return 0;
  const size_t n_c = (size_t)ceil_div(ceil_div(n_ref, K16_TS), K16_BOUNDS_THREADS) * K16_BOUNDS_THREADS;  // centroid rows, padded to whole workgroups
  return n_c * ((size_t)kb * 64 + 2 * sizeof(float)) + 256;
}

// The tile spheres alone, for the tiles [tile_begin, tile_begin + tile_count) (centres as query-operand rows, |c|^2, radii ->
// their slots of temp, which the caller has zeroed): they depend on the cells only, so the ranks of a row-sharded build
// compute a share each and all-gather the three arrays (meld_knn16_sphere_layout) before meld_knn16_bounds_from_spheres.
extern "C" int meld_knn16_tile_spheres(const double* X, int64_t N, int d, const double* mean, const float* scale_info, void* temp,
                                       int64_t tile_begin, int64_t tile_count, meld_stream_t stream) {
  MELD_CHECK_ARG(X && mean && scale_info && temp && N > 0 && tile_begin >= 0 && tile_count >= 0, "meld_knn16_tile_spheres: bad arguments");
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  const int n_t = (int)ceil_div(N, K16_TS);
  MELD_CHECK_ARG(tile_begin + tile_count <= n_t, "meld_knn16_tile_spheres: tile range beyond the %d tiles", n_t);
  if (tile_count == 0) return MELD_OK;
  const size_t n_c = (size_t)ceil_div(n_t, K16_BOUNDS_THREADS) * K16_BOUNDS_THREADS;
  _Float16* c16 = reinterpret_cast<_Float16*>(temp);
  float* cn = reinterpret_cast<float*>(reinterpret_cast<char*>(temp) + n_c * (size_t)KB * 64);
  float* cr = cn + n_c;
  hipLaunchKernelGGL(tile_spheres_kernel, dim3((unsigned)tile_count), dim3(256), sizeof(float) * K16_TS * (size_t)(d | 1), S(stream), X, N, d, mean, scale_info, KB, c16, cn, cr,
                     (int)tile_begin, k16_dA(d, KB));
  MELD_LAUNCH_CHECK("tile_spheres_kernel");
  return MELD_OK;
}
// rows (= tiles, padded to whole workgroups of the table kernel) and bytes per row of the three arrays in temp:
// [rows][row_bytes] centres, then [rows] fp32 |c|^2, then [rows] fp32 radii
extern "C" int meld_knn16_sphere_layout(int64_t n_ref, int d, int64_t* rows, int64_t* row_bytes) {
  MELD_CHECK_ARG(rows && row_bytes && n_ref > 0, "meld_knn16_sphere_layout: bad arguments");
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  *rows = (int64_t)(ceil_div(ceil_div(n_ref, K16_TS), K16_BOUNDS_THREADS) * K16_BOUNDS_THREADS);
  *row_bytes = (int64_t)KB * 64;
  return MELD_OK;
}

static int k16_bounds_impl(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                           const float* norm2_max, const void* Rt16, int64_t q_begin, int64_t q_count,
                           const float* thr_seed, const float* q_norm2, int nprod, void* temp, void* lb2,
                           meld_stream_t stream, bool spheres_ready) {
  MELD_CHECK_ARG(nprod == 1 || nprod == 3, "meld_knn16_bounds: nprod must be 1 or 3");
  MELD_CHECK_ARG(X && mean && scale_info && norm2_max && Rt16 && temp && lb2 && N > 0 && q_count > 0 && q_begin >= 0 &&
                     q_begin + q_count <= N,
                 "meld_knn16_bounds: bad arguments");
  MELD_CHECK_ARG(q_begin % K16_TS == 0, "meld_knn16_bounds: q_begin must be a multiple of the reference tile (64 cells)");
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  const int n_t = (int)ceil_div(N, K16_TS), n_q = (int)ceil_div(q_count, K16_BQ) * K16_NWAVE;  // table rows = waves
  const size_t n_c = (size_t)ceil_div(n_t, K16_BOUNDS_THREADS) * K16_BOUNDS_THREADS;
  hipStream_t st = S(stream);
  _Float16* c16 = reinterpret_cast<_Float16*>(temp);
  float* cn = reinterpret_cast<float*>(reinterpret_cast<char*>(temp) + n_c * (size_t)KB * 64);
  float* cr = cn + n_c;
  if (!spheres_ready) {
    MELD_HIP_CALL(hipMemsetAsync(temp, 0, n_c * ((size_t)KB * 64 + 2 * sizeof(float)), st));
    hipLaunchKernelGGL(tile_spheres_kernel, dim3(n_t), dim3(256), sizeof(float) * K16_TS * (size_t)(d | 1), st, X, N, d, mean, scale_info, KB, c16, cn, cr, 0, k16_dA(d, KB));
  }
  const int bt = KB <= 4 ? K16_BOUNDS_THREADS : K16_BOUNDS_THREADS / 2;
  const int gx = (int)(n_c / bt);
  const int gy = std::max(1, std::min(n_q, (int)ceil_div(2048, gx)));
  const float ec = (float)meld_knn16_error_coef(1, d);
  const float es = (float)meld_knn16_error_coef(nprod, d);  // the search's allowance: a tile is skipped only if d2_approx < thr fails for sure
  // queries = all the cells: wave w is tile w and the table can be symmetrised (see knn16_bounds_symmetrize_kernel)
  const bool symmetric = q_begin == 0 && q_count == N && meld_dev_getenv("MELD_KNN_SYMMETRIC_BOUNDS_OFF") == nullptr;
#define K16_BOUNDS_LAUNCH(KBV, SD, BTV)                                                                                   \
  hipLaunchKernelGGL((knn16_tile_bounds_kernel<KBV, SD, BTV>), dim3(gx, gy), dim3(BTV), 0, st, c16, cn, cr,               \
                     reinterpret_cast<const _Float16*>(Rt16), n_t, (int)(q_begin / K16_TS), n_q, ec, norm2_max,           \
                     scale_info, thr_seed, q_count, es, symmetric ? 1 : 0, q_norm2,                                       \
                     (float)meld_knn16_error_coef_const(nprod, d), (float)meld_knn16_error_coef_lin(nprod),               \
                     reinterpret_cast<__half*>(lb2))
#define K16_BOUNDS_CASE(KBV)                                                                                              \
  case KBV:                                                                                                               \
    if (thr_seed)                                                                                                         \
      K16_BOUNDS_LAUNCH(KBV, true, (KBV <= 4 ? K16_BOUNDS_THREADS : K16_BOUNDS_THREADS / 2));                             \
    else                                                                                                                  \
      K16_BOUNDS_LAUNCH(KBV, false, (KBV <= 4 ? K16_BOUNDS_THREADS : K16_BOUNDS_THREADS / 2));                            \
    break;
  switch (KB) {
    K16_BOUNDS_CASE(4)
#ifndef K16_DEV_KB4
    K16_BOUNDS_CASE(1)
    K16_BOUNDS_CASE(2)
    K16_BOUNDS_CASE(3)
    K16_BOUNDS_CASE(5)
    K16_BOUNDS_CASE(6)
    K16_BOUNDS_CASE(7)
    K16_BOUNDS_CASE(8)
    K16_BOUNDS_CASE(9)
#endif
    default:
      set_err("meld_knn16_bounds: no kernel for %d K blocks", KB);
      return MELD_ERR_UNSUPPORTED;
  }
#undef K16_BOUNDS_CASE
#undef K16_BOUNDS_LAUNCH
  MELD_LAUNCH_CHECK("meld_knn16_bounds");
  if (symmetric) {
    const int nbk = (n_t + 63) / 64;
    hipLaunchKernelGGL(knn16_bounds_symmetrize_kernel, dim3((unsigned)((int64_t)nbk * (nbk + 1) / 2)), dim3(256), 0, st,
                       reinterpret_cast<unsigned short*>(lb2), n_t, n_t);
    MELD_LAUNCH_CHECK("knn16_bounds_symmetrize_kernel");
  }
  return MELD_OK;
}
extern "C" int meld_knn16_bounds(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                                 const float* norm2_max, const void* Rt16, int64_t q_begin, int64_t q_count,
                                 const float* thr_seed, const float* q_norm2, int nprod, void* temp, void* lb2,
                                 meld_stream_t stream) {
  return k16_bounds_impl(X, N, d, mean, scale_info, norm2_max, Rt16, q_begin, q_count, thr_seed, q_norm2, nprod, temp, lb2, stream, false);
}
extern "C" int meld_knn16_bounds_from_spheres(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                                              const float* norm2_max, const void* Rt16, int64_t q_begin, int64_t q_count,
                                              const float* thr_seed, const float* q_norm2, int nprod, void* temp, void* lb2,
                                              meld_stream_t stream) {
  return k16_bounds_impl(X, N, d, mean, scale_info, norm2_max, Rt16, q_begin, q_count, thr_seed, q_norm2, nprod, temp, lb2, stream, true);
}

// Tiles a search workgroup will stage at most: those some wave of it cannot rule out at its start thresholds (the
// table entry <= the wave's largest seed + the search-error allowance).  One workgroup per query block.
__global__ __launch_bounds__(256) void knn16_block_work_kernel(const __half* __restrict__ lb2, const float* __restrict__ thr_seed,
                                                               int n_tiles, float err_coef, const float* __restrict__ norm2_max,
                                                               const float* __restrict__ scale_info, int* __restrict__ work) {
  __shared__ float wm[K16_NWAVE];
  __shared__ int part[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = thr_seed ? thr_seed[(size_t)blockIdx.x * K16_BQ + tid] : INFINITY;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s = fmaxf(s, __shfl_xor(s, off, 64));
  if (lane == 0) wm[wave] = s + err_coef * norm2_max[0] * scale_info[0] * scale_info[0];
  __syncthreads();
  const float w0 = wm[0], w1 = wm[1], w2 = wm[2], w3 = wm[3];
  const __half* row = lb2 + (size_t)blockIdx.x * K16_NWAVE * n_tiles;
  int c = 0;
  for (int t = tid; t < n_tiles; t += 256) {
    const bool live = __half2float(row[t]) <= w0 || __half2float(row[(size_t)n_tiles + t]) <= w1 ||
                      __half2float(row[(size_t)2 * n_tiles + t]) <= w2 || __half2float(row[(size_t)3 * n_tiles + t]) <= w3;
    c += live ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if (lane == 0) part[wave] = c;
  __syncthreads();
  if (tid == 0) work[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

extern "C" int meld_knn16_block_work(const void* lb2, const float* thr_seed, int64_t n_ref, int d, int64_t q_count, int nprod,
                                     const float* norm2_max, const float* scale_info, int32_t* work, meld_stream_t stream) {
  MELD_CHECK_ARG(lb2 && norm2_max && scale_info && work && n_ref > 0 && q_count > 0 && (nprod == 1 || nprod == 3),
                 "meld_knn16_block_work: bad arguments");
  static_assert(K16_NWAVE == 4 && K16_BQ == 256, "knn16_block_work_kernel is written for 4 waves of 64 queries");
  const int n_tiles = (int)ceil_div(n_ref, K16_TS);
  hipLaunchKernelGGL(knn16_block_work_kernel, dim3((unsigned)ceil_div(q_count, K16_BQ)), dim3(256), 0, S(stream),
                     reinterpret_cast<const __half*>(lb2), thr_seed, n_tiles, (float)meld_knn16_error_coef(nprod, d), norm2_max,
                     scale_info, work);
  MELD_LAUNCH_CHECK("knn16_block_work_kernel");
  return MELD_OK;
}

// Step lists of the first pass.  With the seeds the search starts at thresholds that are final for half the rows, and what
// the waves learn during the scan removes only 1.5 % of the (wave, tile) blocks the START thresholds leave (measured at 1M
// cells: 69.3 M wave-blocks with the per-step test against the current thresholds, 70.3 M with the start thresholds alone)
// -- while testing, publishing and merging the per-wave masks every step is half of what a wave does between two tiles.
// So the steps of a query block are fixed before the search: one workgroup per block walks the scan order, keeps the tiles
// some wave of the block cannot rule out at its start thresholds (table entry <= the wave's largest seed + the search-error
// allowance, as in knn16_block_work_kernel) and writes  tile | (waves that need it) << 24  in scan order; cnt[b] = the
// block's steps (>= 1: step 0 is always listed).  The search kernel (LIST instantiation) just walks the list.
__global__ __launch_bounds__(256) void knn16_step_list_kernel(const __half* __restrict__ lb2, const float* __restrict__ thr_seed,
                                                              int n_tiles, float err_coef, const float* __restrict__ norm2_max,
                                                              const float* __restrict__ scale_info, int tile_origin, int two_sided,
                                                              unsigned* __restrict__ list, long long stride, int* __restrict__ cnt) {
  __shared__ float wm[K16_NWAVE];
  __shared__ int wtot[2][K16_NWAVE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bx = blockIdx.x;
  float sd = thr_seed ? thr_seed[(size_t)bx * K16_BQ + tid] : INFINITY;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sd = fmaxf(sd, __shfl_xor(sd, off, 64));
  if (lane == 0) wm[wave] = sd + err_coef * norm2_max[0] * scale_info[0] * scale_info[0];
  __syncthreads();
  const float w0 = wm[0], w1 = wm[1], w2 = wm[2], w3 = wm[3];
  const __half* row = lb2 + (size_t)bx * K16_NWAVE * n_tiles;
  unsigned* out = list + (size_t)bx * (size_t)stride;
  const int t0 = (int)(((long long)tile_origin + (long long)bx * (K16_BQ / K16_TS)) % n_tiles);
  int base = 0, par = 0;
  for (int s0 = 0; s0 < n_tiles; s0 += 256, par ^= 1) {
    const int sidx = s0 + tid;
    unsigned mask = 0u;
    int t = 0;
    if (sidx < n_tiles) {
      t = k16_scan_tile(sidx, t0, n_tiles, two_sided);
      mask = (__half2float(row[t]) <= w0 ? 1u : 0u) | (__half2float(row[(size_t)n_tiles + t]) <= w1 ? 2u : 0u) |
             (__half2float(row[(size_t)2 * n_tiles + t]) <= w2 ? 4u : 0u) | (__half2float(row[(size_t)3 * n_tiles + t]) <= w3 ? 8u : 0u);
    }
    const bool keep = mask != 0u || sidx == 0;
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wtot[par][wave] = __popcll(bal);
    __syncthreads();  // (the two parities alternate: the next round's writes cannot overtake this round's reads)
    int before = base;
    for (int w = 0; w < wave; ++w) before += wtot[par][w];
    if (keep) out[before + __popcll(bal & (((unsigned long long)1 << lane) - 1ull))] = (unsigned)t | (mask << 24);
    base += wtot[par][0] + wtot[par][1] + wtot[par][2] + wtot[par][3];
  }
  if (tid == 0) cnt[bx] = base;
}

// wthr[w] = largest start threshold of wave w (its 64 queries) + the search-error allowance, -inf for a wave without cells
__global__ __launch_bounds__(256) void knn16_wave_thresholds_kernel(const float* __restrict__ thr_seed, int64_t n_seed, int n_rows,
                                                                    float err_coef, const float* __restrict__ norm2_max,
                                                                    const float* __restrict__ scale_info, float* __restrict__ wthr) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_rows) return;
  const int64_t q = (int64_t)w * 64 + lane;
  float sd = q < n_seed ? thr_seed[q] : -INFINITY;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sd = fmaxf(sd, __shfl_xor(sd, off, 64));
  if (lane == 0) wthr[w] = sd + err_coef * norm2_max[0] * scale_info[0] * scale_info[0];
}

// live[w][.] = A[w][.] & transpose(B)[w][.]: one wave per 64 x 64 bit tile (lane l loads word R of row 64 C + l of B; bit j of
// the 64 loaded words, gathered by a ballot, is the transposed word of row 64 R + j, columns 64 C ...)
__global__ __launch_bounds__(256) void knn16_bits_transpose_and_kernel(const unsigned long long* __restrict__ bits_a,
                                                                       const unsigned long long* __restrict__ bits_b, int n_rows, int wpr,
                                                                       unsigned long long* __restrict__ live) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int R = tile / wpr, C = tile % wpr;  // output rows 64 R ..., output word C
  if (R >= (n_rows + 63) / 64) return;
  const int src_row = C * 64 + lane;
  const unsigned long long x = (src_row < n_rows && R < wpr) ? bits_b[(size_t)src_row * wpr + R] : 0ull;
  unsigned long long mine = 0ull;
#pragma unroll 8
  for (int j = 0; j < 64; ++j) {
    const unsigned long long t = __ballot((x >> j) & 1ull);
    if (lane == j) mine = t;
  }
  const int out_row = R * 64 + lane;
  if (out_row < n_rows) live[(size_t)out_row * wpr + C] = mine & bits_a[(size_t)out_row * wpr + C];
}

// knn16_step_list_kernel on the bit form of the table (live[w][t]: wave w cannot rule tile t out)
__global__ __launch_bounds__(256) void knn16_step_list_bits_kernel(const unsigned long long* __restrict__ live, int wpr, int n_tiles,
                                                                   int tile_origin, int two_sided, unsigned* __restrict__ list,
                                                                   long long stride, int* __restrict__ cnt) {
  __shared__ int wtot[2][K16_NWAVE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bx = blockIdx.x;
  const unsigned long long* row = live + (size_t)bx * K16_NWAVE * wpr;
  unsigned* out = list + (size_t)bx * (size_t)stride;
  const int t0 = (int)(((long long)tile_origin + (long long)bx * (K16_BQ / K16_TS)) % n_tiles);
  int base = 0, par = 0;
  for (int s0 = 0; s0 < n_tiles; s0 += 256, par ^= 1) {
    const int sidx = s0 + tid;
    unsigned mask = 0u;
    int t = 0;
    if (sidx < n_tiles) {
      t = k16_scan_tile(sidx, t0, n_tiles, two_sided);
      const int wd = t >> 6, bt = t & 63;
#pragma unroll
      for (int w = 0; w < K16_NWAVE; ++w) mask |= (unsigned)((row[(size_t)w * wpr + wd] >> bt) & 1ull) << w;
    }
    const bool keep = mask != 0u || sidx == 0;
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wtot[par][wave] = __popcll(bal);
    __syncthreads();
    int before = base;
    for (int w = 0; w < wave; ++w) before += wtot[par][w];
    if (keep) out[before + __popcll(bal & (((unsigned long long)1 << lane) - 1ull))] = (unsigned)t | (mask << 24);
    base += wtot[par][0] + wtot[par][1] + wtot[par][2] + wtot[par][3];
  }
  if (tid == 0) cnt[bx] = base;
}

static int k16_two_sided() {  // scan order: own tiles, then alternately forwards / backwards (0 = forwards only; profiling hook)
  const char* e = meld_dev_getenv("MELD_KNN16_TWO_SIDED");
  return e ? (atoi(e) != 0) : 1;
}

extern "C" int meld_knn16_step_lists(const void* lb2, const float* thr_seed, int64_t n_ref, int d, int64_t q_count, int nprod,
                                     const float* norm2_max, const float* scale_info, int64_t q_begin, uint32_t* list,
                                     int64_t list_stride, int32_t* cnt, meld_stream_t stream) {
  MELD_CHECK_ARG(lb2 && norm2_max && scale_info && list && cnt && n_ref > 0 && q_count > 0 && (nprod == 1 || nprod == 3) && q_begin >= 0,
                 "meld_knn16_step_lists: bad arguments");
  static_assert(K16_NWAVE == 4 && K16_BQ == 256, "knn16_step_list_kernel is written for 4 waves of 64 queries");
  const int n_tiles = (int)ceil_div(n_ref, K16_TS);
  MELD_CHECK_ARG(n_tiles < (1 << 24) && list_stride >= n_tiles, "meld_knn16_step_lists: list_stride must hold the %d tiles of a block", n_tiles);
  const int tile_origin = (int)((q_begin / K16_TS) % n_tiles);
  hipLaunchKernelGGL(knn16_step_list_kernel, dim3((unsigned)ceil_div(q_count, K16_BQ)), dim3(256), 0, S(stream),
                     reinterpret_cast<const __half*>(lb2), thr_seed, n_tiles, (float)meld_knn16_error_coef(nprod, d), norm2_max, scale_info,
                     tile_origin, k16_two_sided(), list, (long long)list_stride, cnt);
  MELD_LAUNCH_CHECK("knn16_step_list_kernel");
  return MELD_OK;
}

// Step lists straight from the cells (queries = all the cells, start thresholds known): tile spheres, the cells x centroids
// bounds in their bit form (knn16_tile_bounds_kernel<..., BITS>), transpose-and-AND, lists -- the fp16 table of meld_knn16_bounds
// (2 B per (wave, tile): 488 MB at 1M cells), its symmetrisation pass and the list builder's pass over it never exist.
// Same lists as meld_knn16_bounds + meld_knn16_step_lists (tests/test_gpu_parity.py compares them entry by entry).
extern "C" size_t meld_knn16_list_scratch_bytes(int64_t n_ref) {
  const size_t n_q = (size_t)ceil_div(n_ref, K16_BQ) * K16_NWAVE, wpr = (size_t)ceil_div(ceil_div(n_ref, K16_TS), 64);
  return ((n_q * sizeof(float) + 255) / 256) * 256 + 3 * n_q * wpr * sizeof(unsigned long long);
}

static int k16_step_lists_direct_impl(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                                      const float* norm2_max, const void* Rt16, const float* thr_seed, const float* q_norm2,
                                      int nprod, void* temp, void* scratch, uint32_t* list, int64_t list_stride, int32_t* cnt,
                                      int lead_only, meld_stream_t stream) {
  MELD_CHECK_ARG(nprod == 1 || nprod == 3, "meld_knn16_step_lists_direct: nprod must be 1 or 3");
  MELD_CHECK_ARG(X && mean && scale_info && norm2_max && Rt16 && thr_seed && q_norm2 && temp && scratch && list && cnt && N > 0,
                 "meld_knn16_step_lists_direct: bad arguments");
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  const int n_t = (int)ceil_div(N, K16_TS), n_q = (int)ceil_div(N, K16_BQ) * K16_NWAVE;
  MELD_CHECK_ARG(n_t < (1 << 24) && list_stride >= n_t, "meld_knn16_step_lists_direct: list_stride must hold the %d tiles of a block", n_t);
  const int wpr = (int)ceil_div(n_t, 64);
  const size_t n_c = (size_t)ceil_div(n_t, K16_BOUNDS_THREADS) * K16_BOUNDS_THREADS;
  hipStream_t st = S(stream);
  _Float16* c16 = reinterpret_cast<_Float16*>(temp);
  float* cn = reinterpret_cast<float*>(reinterpret_cast<char*>(temp) + n_c * (size_t)KB * 64);
  float* cr = cn + n_c;
  float* wthr = reinterpret_cast<float*>(scratch);
  unsigned long long* bits_a = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(scratch) + (((size_t)n_q * sizeof(float) + 255) / 256) * 256);
  unsigned long long* bits_b = bits_a + (size_t)n_q * wpr;
  unsigned long long* live = bits_b + (size_t)n_q * wpr;
  MELD_HIP_CALL(hipMemsetAsync(temp, 0, n_c * ((size_t)KB * 64 + 2 * sizeof(float)), st));
  hipLaunchKernelGGL(tile_spheres_kernel, dim3(n_t), dim3(256), sizeof(float) * K16_TS * (size_t)(d | 1), st, X, N, d, mean, scale_info, KB, c16, cn, cr, 0, k16_dA(d, KB));
  const float es = (float)meld_knn16_error_coef(nprod, d);
  hipLaunchKernelGGL(knn16_wave_thresholds_kernel, dim3((unsigned)ceil_div(n_q, 4)), dim3(256), 0, st, thr_seed, N, n_q, es, norm2_max,
                     scale_info, wthr);
  // (rows of waves that no workgroup of the bounds kernel reaches, and the words behind the last tile, must read as zero)
  MELD_HIP_CALL(hipMemsetAsync(bits_a, 0, 2 * (size_t)n_q * wpr * sizeof(unsigned long long), st));
  const int bt = KB <= 4 ? K16_BOUNDS_THREADS : K16_BOUNDS_THREADS / 2;
  const int gx = (int)(n_c / bt);
  const int gy = std::max(1, std::min(n_q, (int)ceil_div(2048, gx)));
  const float ec = (float)meld_knn16_error_coef(1, d);
  // lead_only (the caller's word that the leading coordinates carry the distances, SPLIT layout only): bounds from K block 0 alone
  const int dA = k16_dA(d, KB);
  const bool lead = lead_only != 0 && dA > 0 && KB >= 2 && !(meld_dev_getenv("MELD_KNN16_LEAD_BOUNDS") && atoi(meld_dev_getenv("MELD_KNN16_LEAD_BOUNDS")) == 0);
#define K16_BITS_LAUNCH(KBV, BTV)                                                                                          \
  if (lead)                                                                                                                \
    hipLaunchKernelGGL((knn16_tile_bounds_kernel<KBV, true, BTV, true, (KBV >= 2 ? 1 : KBV)>), dim3(gx, gy), dim3(BTV), 0, st, c16, cn, cr, \
                       reinterpret_cast<const _Float16*>(Rt16), n_t, 0, n_q, ec, norm2_max, scale_info, thr_seed, N, es, 1, q_norm2, \
                       (float)meld_knn16_error_coef_const(nprod, d), (float)meld_knn16_error_coef_lin(nprod), (__half*)nullptr, wthr,  \
                       bits_a, bits_b, wpr, dA);                                                                          \
  else                                                                                                                     \
  hipLaunchKernelGGL((knn16_tile_bounds_kernel<KBV, true, BTV, true>), dim3(gx, gy), dim3(BTV), 0, st, c16, cn, cr,        \
                     reinterpret_cast<const _Float16*>(Rt16), n_t, 0, n_q, ec, norm2_max, scale_info, thr_seed, N, es, 1, q_norm2, \
                     (float)meld_knn16_error_coef_const(nprod, d), (float)meld_knn16_error_coef_lin(nprod), (__half*)nullptr, wthr,  \
                     bits_a, bits_b, wpr)
#define K16_BITS_CASE(KBV) \
  case KBV:                \
    K16_BITS_LAUNCH(KBV, (KBV <= 4 ? K16_BOUNDS_THREADS : K16_BOUNDS_THREADS / 2)); \
    break;
  switch (KB) {
    K16_BITS_CASE(4)
#ifndef K16_DEV_KB4
    K16_BITS_CASE(1)
    K16_BITS_CASE(2)
    K16_BITS_CASE(3)
    K16_BITS_CASE(5)
    K16_BITS_CASE(6)
    K16_BITS_CASE(7)
    K16_BITS_CASE(8)
    K16_BITS_CASE(9)
#endif
    default:
      set_err("meld_knn16_step_lists_direct: no kernel for %d K blocks", KB);
      return MELD_ERR_UNSUPPORTED;
  }
#undef K16_BITS_CASE
#undef K16_BITS_LAUNCH
  MELD_LAUNCH_CHECK("knn16_tile_bounds_kernel(bits)");
  const int64_t n_tiles64 = (int64_t)ceil_div(n_q, 64) * wpr;
  hipLaunchKernelGGL(knn16_bits_transpose_and_kernel, dim3((unsigned)ceil_div(n_tiles64, 4)), dim3(256), 0, st, bits_a, bits_b, n_q, wpr, live);
  hipLaunchKernelGGL(knn16_step_list_bits_kernel, dim3((unsigned)ceil_div(N, K16_BQ)), dim3(256), 0, st, live, wpr, n_t, 0, k16_two_sided(), list,
                     (long long)list_stride, cnt);
  MELD_LAUNCH_CHECK("knn16_step_list_bits_kernel");
  return MELD_OK;
}

extern "C" int meld_knn16_step_lists_direct(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                                            const float* norm2_max, const void* Rt16, const float* thr_seed, const float* q_norm2,
                                            int nprod, void* temp, void* scratch, uint32_t* list, int64_t list_stride, int32_t* cnt,
                                            meld_stream_t stream) {
  return k16_step_lists_direct_impl(X, N, d, mean, scale_info, norm2_max, Rt16, thr_seed, q_norm2, nprod, temp, scratch, list, list_stride, cnt, 0,
                                    stream);
}
// The same with the bounds taken from the first K block of the operands alone (lead_only != 0; SPLIT layout, cells handed over in
// a frame whose leading coordinates carry the distances): lower bounds all the same -- the lists are a superset of the exact ones
// either way -- for a quarter of the tile stream and two of five MFMAs per bound.
extern "C" int meld_knn16_step_lists_direct_lead(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                                                 const float* norm2_max, const void* Rt16, const float* thr_seed, const float* q_norm2,
                                                 int nprod, void* temp, void* scratch, uint32_t* list, int64_t list_stride, int32_t* cnt,
                                                 int lead_only, meld_stream_t stream) {
  return k16_step_lists_direct_impl(X, N, d, mean, scale_info, norm2_max, Rt16, thr_seed, q_norm2, nprod, temp, scratch, list, list_stride, cnt,
                                    lead_only, stream);
}

// Start values for the thresholds of meld_knn16_topk's first pass (thr_init, scaled units, roundup(q_count, BQ)
// floats) from every query's own block of BQ cells; q_begin must be a multiple of BQ.  knn, radius_factor as for the
// radius cut.  Rows whose block holds fewer than knn + 1 cells (or knn + 1 > 64) get +inf.
extern "C" int meld_knn16_seed_thresholds(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                                          const float* norm2_max, int64_t q_begin, int64_t q_count, int knn,
                                          double radius_factor, int nprod, float* thr_init, meld_stream_t stream) {
  MELD_CHECK_ARG(X && mean && scale_info && norm2_max && thr_init && N > 0 && q_count > 0 && q_begin >= 0 &&
                     q_begin + q_count <= N,
                 "meld_knn16_seed_thresholds: bad arguments");
  MELD_CHECK_ARG(q_begin % K16_BQ == 0, "meld_knn16_seed_thresholds: q_begin must be a multiple of the query block (%d)", K16_BQ);
  MELD_CHECK_ARG(knn >= 1 && radius_factor >= 1.0 && (nprod == 1 || nprod == 3), "meld_knn16_seed_thresholds: bad kernel parameters");
  if (meld_knn16_kblocks(d) < 0) return MELD_ERR_UNSUPPORTED;
  const int n_b = (int)ceil_div(q_count, K16_BQ);
  hipStream_t st = S(stream);
  if (knn + 1 > SEED_KMAX || d > 144) {  // no seed: the search starts at +inf as before
    const size_t n = (size_t)n_b * K16_BQ;
    MELD_HIP_CALL(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(thr_init), 0x7f800000, n, st));
    return MELD_OK;
  }
  const float rf2 = (float)(radius_factor * radius_factor * (1.0 + 1e-6));
#define K16_SEED_LAUNCH2(KV, DV)                                                                                          \
  do {                                                                                                                    \
    const size_t lds = sizeof(float) * (size_t)K16_BQ * DV;                                                               \
    if (lds > 64 * 1024)                                                                                                  \
      MELD_HIP_CALL(hipFuncSetAttribute(reinterpret_cast<const void*>(&knn16_seed_kernel<KV, DV>),                        \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
    hipLaunchKernelGGL((knn16_seed_kernel<KV, DV>), dim3(n_b), dim3(256), lds, st, X, N, d, mean, scale_info, norm2_max,   \
                       q_begin, q_count, knn + 1, rf2, (float)meld_knn16_error_coef_const(nprod, d),                      \
                       (float)meld_knn16_error_coef_lin(nprod), thr_init);                                                \
  } while (0)
#define K16_SEED_LAUNCH(KV)        \
  do {                             \
    if (d <= 60)                   \
      K16_SEED_LAUNCH2(KV, 60);    \
    else if (d <= 104)             \
      K16_SEED_LAUNCH2(KV, 104);   \
    else                           \
      K16_SEED_LAUNCH2(KV, 144);   \
  } while (0)
  if (knn + 1 <= 8) {
    K16_SEED_LAUNCH(8);
  } else if (knn + 1 <= 16) {
    K16_SEED_LAUNCH(16);
  } else if (knn + 1 <= 32) {
    K16_SEED_LAUNCH(32);
  } else {
    K16_SEED_LAUNCH(64);
  }
#undef K16_SEED_LAUNCH
#undef K16_SEED_LAUNCH2
  MELD_LAUNCH_CHECK("knn16_seed_kernel");
  return MELD_OK;
}

// The same from the fp16 operands of meld_knn16_prepare, on the matrix pipe, over the query block's own tiles and four
// on either side (see knn16_seed_mfma_kernel).  Q16 / Qn / Rt16 / scale_info / norm2_max as for meld_knn16_topk.
extern "C" int meld_knn16_seed_thresholds_mfma(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info,
                                               const float* norm2_max, int64_t n_ref, int d, int64_t q_begin,
                                               int64_t q_count, int knn, double radius_factor, int nprod, int side_tiles,
                                               float* thr_init, meld_stream_t stream) {
  const int side = side_tiles > 0 ? side_tiles : (int)std::min<double>(64.0, std::max<double>(8.0, (double)n_ref / 12500.0));
  MELD_CHECK_ARG(Q16 && Qn && Rt16 && scale_info && norm2_max && thr_init && n_ref > 0 && q_count > 0 && q_begin >= 0,
                 "meld_knn16_seed_thresholds_mfma: bad arguments");
  MELD_CHECK_ARG(q_begin % K16_BQ == 0, "meld_knn16_seed_thresholds_mfma: q_begin must be a multiple of the query block (%d)", K16_BQ);
  MELD_CHECK_ARG(knn >= 1 && radius_factor >= 1.0 && (nprod == 1 || nprod == 3), "meld_knn16_seed_thresholds_mfma: bad kernel parameters");
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  const int n_b = (int)ceil_div(q_count, K16_BQ);
  hipStream_t st = S(stream);
  if (knn + 1 > SEED_KMAX) {
    MELD_HIP_CALL(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(thr_init), 0x7f800000, (size_t)n_b * K16_BQ, st));
    return MELD_OK;
  }
  const float rf2 = (float)(radius_factor * radius_factor * (1.0 + 1e-6));
  const int n_tiles = (int)ceil_div(n_ref, K16_TS);
  // the hi-only products are what the kernel computes, whatever the search's nprod: charge the hi-only allowance
  const float ec = (float)meld_knn16_error_coef_const(1, d), el = (float)meld_knn16_error_coef_lin(1);
  (void)nprod;
#define K16_SEEDM_LAUNCH(KBV, KV)                                                                                       \
  hipLaunchKernelGGL((knn16_seed_mfma_kernel<KBV, KV>), dim3((unsigned)ceil_div(n_b * K16_NWAVE, SEED_WAVES)),            \
                     dim3(64 * SEED_WAVES), 0, st, reinterpret_cast<const _Float16*>(Q16), Qn,                            \
                     reinterpret_cast<const _Float16*>(Rt16), scale_info, norm2_max, n_tiles, (int)(q_begin / K16_TS),   \
                     side, n_b * K16_NWAVE, knn + 1, rf2, ec, el, thr_init)
#define K16_SEEDM_CASE(KBV)          \
  case KBV:                          \
    if (knn + 1 <= 16)               \
      K16_SEEDM_LAUNCH(KBV, 16);     \
    else                             \
      K16_SEEDM_LAUNCH(KBV, 64);     \
    break;
  switch (KB) {
    K16_SEEDM_CASE(4)
#ifndef K16_DEV_KB4
    K16_SEEDM_CASE(1)
    K16_SEEDM_CASE(2)
    K16_SEEDM_CASE(3)
    K16_SEEDM_CASE(5)
    K16_SEEDM_CASE(6)
    K16_SEEDM_CASE(7)
    K16_SEEDM_CASE(8)
    K16_SEEDM_CASE(9)
#endif
    default:
      set_err("meld_knn16_seed_thresholds_mfma: no kernel for %d K blocks", KB);
      return MELD_ERR_UNSUPPORTED;
  }
#undef K16_SEEDM_CASE
#undef K16_SEEDM_LAUNCH
  MELD_LAUNCH_CHECK("knn16_seed_mfma_kernel");
  return MELD_OK;
}

static int k16_topk_impl(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info,
                         int64_t n_ref, int d,
                         int64_t q_count, int ksel, int nprod, int n_slices, const void* lb2,
                         const float* norm2_max, int64_t q_begin, const float* thr_init, int knn,
                         double radius_factor, int32_t* cand_idx, float* cand_d2, int32_t* cand_cnt,
                         float* cand_thr, uint64_t* tiles_done, const int32_t* block_order, const uint32_t* step_list,
                         const int32_t* step_cnt, int64_t list_stride, meld_stream_t stream, int partial_test = 0, int two_counters = 0) {
  MELD_CHECK_ARG(nprod == 1 || nprod == 3, "meld_knn16_topk: nprod must be 1 or 3");
  MELD_CHECK_ARG(step_list == nullptr || (step_cnt != nullptr && nprod == 1 && lb2 == nullptr && thr_init != nullptr),
                 "meld_knn16_topk_listed: step lists go with the hi-only pass, start thresholds and no table");
  // (reference slices combine with pruning, the radius cut and a dispatch order: a slice's table row is the wave's row, its
  // rows are cut at the radius ITS references imply -- looser than the global one, still valid -- and cand_thr then holds
  // n_slices x q_pad thresholds of which the caller takes the minimum per query)
  MELD_CHECK_ARG(n_slices >= 1 && n_slices * ksel <= K16_MERGE_MAX, "meld_knn16_topk: n_slices=%d must satisfy n_slices * ksel <= %d",
                 n_slices, K16_MERGE_MAX);
  MELD_CHECK_ARG(Q16 && Qn && Rt16 && scale_info && cand_idx && cand_d2 && cand_cnt, "meld_knn16_topk: null pointer");
  MELD_CHECK_ARG(lb2 == nullptr || norm2_max != nullptr, "meld_knn16_topk: pruning needs norm2_max");
  // radius cut (cand_thr != NULL): rows are cut at the kernel radius implied by their knn-th neighbour so far
  // (knn == 0 with cand_thr: the final thresholds are published but no row is cut -- a search whose radii are fixed by the
  // caller's start thresholds, graphtools' `bandwidth=`)
  MELD_CHECK_ARG(cand_thr == nullptr || (norm2_max != nullptr && knn >= 0 && knn < ksel && radius_factor >= 1.0),
                 "meld_knn16_topk: the radius cut needs norm2_max, 0 <= knn < ksel, radius_factor >= 1");
  const int knn1 = (cand_thr && knn >= 1) ? knn + 1 : 0;
  const float rf2 = knn1 > 0 ? (float)(radius_factor * radius_factor * (1.0 + 1e-6)) : 0.0f;
  const float err_c = (float)meld_knn16_error_coef_const(nprod, d), err_l = (float)meld_knn16_error_coef_lin(nprod);
  MELD_CHECK_ARG(n_ref > 0 && n_ref < (int64_t)1 << 31 && q_count > 0, "meld_knn16_topk: bad sizes");
  const int cap = meld_knn16_row_capacity(ksel);
  if (cap < 0) return cap;
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  const int n_tiles = (int)ceil_div(n_ref, K16_TS);
  const dim3 grid((unsigned)n_slices, (unsigned)ceil_div(q_count, K16_BQ));
  MELD_CHECK_ARG(grid.y <= 65535u, "meld_knn16_topk: more than 65535 query blocks in one launch");
  MELD_CHECK_ARG(n_slices <= ceil_div(n_ref, K16_TS), "meld_knn16_topk: more slices than reference tiles");
  const int tile_origin = (int)((q_begin / K16_TS) % n_tiles);  // the scan starts at the queries' own position
  // profiling hooks (never set in production): MELD_KNN16_ABLATION=1 distances without selection,
  // =3 MFMAs only; MELD_KNN16_PADLDS=<bytes> extra dynamic LDS to lower the workgroups per CU
  const char* abl_env = meld_dev_getenv("MELD_KNN16_ABLATION");
  const int abl = abl_env ? atoi(abl_env) : 0;
  const char* pad_env = meld_dev_getenv("MELD_KNN16_PADLDS");
  const size_t pad_lds = pad_env ? (size_t)atoi(pad_env) : 0;
  const _Float16* q = reinterpret_cast<const _Float16*>(Q16);
  const _Float16* r = reinterpret_cast<const _Float16*>(Rt16);
  unsigned long long* stats = nullptr;
  // batched compaction: every 64 tiles, rows with > ksel + 64 entries (with the seeded thresholds the rows fill slowly:
  // 32 / 32, the setting before the seeds, costs 0.9 ms more at 1M cells; none at all 0.4 ms more)
  int batch_every = 64, batch_slack = 64;
  if (const char* e = meld_dev_getenv("MELD_KNN16_BATCH_EVERY")) {  // profiling hooks (a power of two, or 0 = off)
    batch_every = atoi(e);
    MELD_CHECK_ARG(batch_every >= 0 && (batch_every & (batch_every - 1)) == 0, "MELD_KNN16_BATCH_EVERY must be a power of two");
  }
  if (const char* e = meld_dev_getenv("MELD_KNN16_BATCH_SLACK")) batch_slack = std::max(0, atoi(e));
  const int two_sided = k16_two_sided();
  if (meld_dev_getenv("MELD_KNN16_STATS")) {  // profiling hook: selection counters, printed after the launch
    static unsigned long long* counters[64] = {nullptr};  // (a racing first call leaks 256 B at worst)
    int dev = 0;
    MELD_HIP_CALL(hipGetDevice(&dev));
    MELD_CHECK_ARG(dev >= 0 && dev < 64, "meld_knn16_topk: device index out of range");
    if (counters[dev] == nullptr) MELD_HIP_CALL(hipMalloc(reinterpret_cast<void**>(&counters[dev]), 256));
    MELD_HIP_CALL(hipMemsetAsync(counters[dev], 0, 256, S(stream)));
    stats = counters[dev];
  }
  K16Args ka;
  ka.Q16 = q;
  ka.Qn = Qn;
  ka.Rt16 = r;
  ka.scale_info = scale_info;
  ka.n_ref = (int)n_ref;
  ka.n_tiles = n_tiles;
  ka.ksel = ksel;
  ka.cap = cap;
  ka.lb2 = reinterpret_cast<const __half*>(lb2);
  ka.norm2_max = norm2_max;
  ka.err_coef = (float)meld_knn16_error_coef(nprod, d);
  ka.tile_origin = tile_origin;
  ka.batch_every = batch_every;
  ka.batch_slack = batch_slack;
  ka.two_sided = two_sided;
  ka.stats = stats;
  ka.thr_init = thr_init;
  ka.knn1 = knn1;
  ka.rf2 = rf2;
  ka.err_c = err_c;
  ka.err_l = err_l;
  ka.cand_idx = cand_idx;
  ka.cand_d2 = cand_d2;
  ka.cand_cnt = cand_cnt;
  ka.cand_thr = cand_thr;
  ka.tiles_done = reinterpret_cast<unsigned long long*>(tiles_done);
  ka.block_order = block_order;
  ka.step_list = step_list;
  ka.step_cnt = step_cnt;
  ka.list_stride = (long long)list_stride;
  // the list-driven first pass tests its blocks behind K block 0 where the operands carry the SPLIT layout and the caller asks
  // for it (`partial_test`: it knows whether the leading coordinates carry the distances -- where they do not, the test drops
  // nothing and costs 15 %); MELD_KNN16_EE=0 / 1 overrides the caller, for A-B measurements
  const int dA = k16_dA(d, KB);
  const char* ee_env = meld_dev_getenv("MELD_KNN16_EE");
  const char* ee_max = meld_dev_getenv("MELD_KNN16_EE_MAXKB");  // (development: the widest operand the partial-test kernel is taken for)
  const bool ee = step_list != nullptr && dA > 0 && KB >= 2 && KB <= (ee_max ? atoi(ee_max) : 7) && (ee_env ? atoi(ee_env) != 0 : partial_test != 0);
  ka.ee_hi = 16 + d - dA;
  ka.count_go = two_counters;  // (a caller of meld_knn16_topk_listed passes ONE counter, whatever MELD_KNN16_EE forces)
  {
    const int slots_used = dA > 0 ? 16 + (d - dA) + 3 : d + 3;
    const char* sk = meld_dev_getenv("MELD_KNN16_SKIP_PAD");  // (=0: copy the padding plane as before, for A-B measurements)
    ka.planes_used = (sk && atoi(sk) == 0) ? 2 * KB : (slots_used + 7) / 8;
  }
#ifndef K16_PROFILING
  MELD_CHECK_ARG(abl == 0, "MELD_KNN16_ABLATION needs a library built with -DK16_PROFILING");
#endif
#ifdef K16_PROFILING  // the timing-only ablations of the list-driven kernel (8: tiles from a 64-tile hot set, 9: no tile loads, 1: no selection)
#define K16_DEV_LIST_ABL(KBV)                                                                                       \
  if (abl == 9) hipLaunchKernelGGL((knn16_topk_kernel<KBV, 9, 1, true>), grid, dim3(K16_THREADS), pad_lds, S(stream), ka); \
  else if (abl == 8) hipLaunchKernelGGL((knn16_topk_kernel<KBV, 8, 1, true>), grid, dim3(K16_THREADS), pad_lds, S(stream), ka); \
  else if (abl == 1) hipLaunchKernelGGL((knn16_topk_kernel<KBV, 1, 1, true>), grid, dim3(K16_THREADS), pad_lds, S(stream), ka); \
  else
#else
#define K16_DEV_LIST_ABL(KBV)
#endif
#define K16_LAUNCH2(KBV, ABLV, NP) \
  hipLaunchKernelGGL((knn16_topk_kernel<KBV, ABLV, NP>), grid, dim3(K16_THREADS), pad_lds, S(stream), ka)
#define K16_LAUNCH(KBV, ABLV)        \
  do {                               \
    if (step_list != nullptr) {      \
      K16_DEV_LIST_ABL(KBV)          \
      if (ee && KBV >= 2 && KBV <= 7) { /* (seven K blocks: d = 100, the reference's default n_pca; eight and nine are not instantiated) */ \
        if (stats != nullptr)        \
          hipLaunchKernelGGL((knn16_topk_kernel<((KBV >= 2 && KBV <= 7) ? KBV : 2), 2, 1, true, true>), grid, dim3(K16_THREADS), pad_lds, S(stream), ka); \
        else                         \
          hipLaunchKernelGGL((knn16_topk_kernel<((KBV >= 2 && KBV <= 7) ? KBV : 2), 0, 1, true, true>), grid, dim3(K16_THREADS), pad_lds, S(stream), ka); \
      } else if (stats != nullptr)   \
        hipLaunchKernelGGL((knn16_topk_kernel<KBV, 2, 1, true>), grid, dim3(K16_THREADS), pad_lds, S(stream), ka); \
      else                           \
        hipLaunchKernelGGL((knn16_topk_kernel<KBV, 0, 1, true>), grid, dim3(K16_THREADS), pad_lds, S(stream), ka); \
    } else if (nprod == 1)           \
      K16_LAUNCH2(KBV, ABLV, 1);     \
    else                             \
      K16_LAUNCH2(KBV, ABLV, 3);     \
  } while (0)
#ifndef K16_PROFILING  // product builds: the search and its counters (MELD_KNN16_STATS); the timing-only ablations of tools/knn_ablate.py are
                       // compiled with -DK16_PROFILING (tools/build_variant.sh prof knn16.hip -DK16_PROFILING [-DK16_DEV_KB4 = d <= 61 only])
#define K16_CASE(KBV)                  \
  case KBV:                            \
    if (stats != nullptr)              \
      K16_LAUNCH(KBV, 2);              \
    else                               \
      K16_LAUNCH(KBV, 0);              \
    break;
#else
#define K16_CASE(KBV)                  \
  case KBV:                            \
    if (abl == 1)                      \
      K16_LAUNCH(KBV, 1);              \
    else if (abl == 3)                 \
      K16_LAUNCH(KBV, 3);              \
    else if (abl == 4 && KBV == 4)     \
      K16_LAUNCH(4, 4);                \
    else if (abl == 5 && KBV == 4)     \
      K16_LAUNCH(4, 5);                \
    else if (abl == 7 && KBV == 4)     \
      K16_LAUNCH(4, 7);                \
    else if (abl == 10 && KBV == 4)    \
      K16_LAUNCH(4, 10);               \
    else if (abl == 8 && KBV == 4)     \
      K16_LAUNCH(4, 8);                \
    else if (abl == 9 && KBV == 4)     \
      K16_LAUNCH(4, 9);                \
    else if (stats != nullptr)         \
      K16_LAUNCH(KBV, 2);              \
    else if (abl == 6 && KBV <= 4)     \
      K16_LAUNCH((KBV <= 4 ? KBV : 1), 6); \
    else                               \
      K16_LAUNCH(KBV, 0);              \
    break;
#endif
  switch (KB) {
    K16_CASE(4)
#ifndef K16_DEV_KB4
    K16_CASE(1)
    K16_CASE(2)
    K16_CASE(3)
    K16_CASE(5)
    K16_CASE(6)
    K16_CASE(7)
    K16_CASE(8)
    K16_CASE(9)
#endif
    default:
      set_err("meld_knn16_topk: KB=%d is not an instantiated size", KB);
      return MELD_ERR_UNSUPPORTED;
  }
#undef K16_CASE
#undef K16_LAUNCH
#undef K16_LAUNCH2
  MELD_LAUNCH_CHECK("knn16_topk_kernel");
  if (abl == 0 || abl == 2 || abl == 5 || abl == 6) {  // (the timing-only ablations leave no rows to sort)
    const int64_t n_rows = (int64_t)grid.x * K16_BQ * grid.y;
    const dim3 fgrid((unsigned)ceil_div(n_rows, 4));
    hipLaunchKernelGGL(knn16_finish_rows_kernel, fgrid, dim3(256), 0, S(stream), cand_d2, cand_idx, cand_cnt, Qn, scale_info, n_rows,
                       (int64_t)grid.y * K16_BQ, cap, ksel);
    MELD_LAUNCH_CHECK("knn16_finish_rows_kernel");
  }
  if (stats) {
    unsigned long long st[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    MELD_HIP_CALL(hipStreamSynchronize(S(stream)));
    MELD_HIP_CALL(hipMemcpy(st, stats, sizeof(st), hipMemcpyDeviceToHost));
    fprintf(stderr, "[knn16 stats] wave-blocks %llu  slow-path entries %llu (%.1f %%)  appends %llu (%.1f per query)  compactions %llu  tiles staged %llu (%.1f %% of workgroups x tiles)\n",
            st[0], st[1], st[0] ? 100.0 * (double)st[1] / (double)st[0] : 0.0, st[2], (double)st[2] / (double)q_count, st[3], st[4],
            100.0 * (double)st[4] / ((double)grid.y * (double)n_tiles));
    if (ee)
      fprintf(stderr, "[knn16 stats] blocks of 32 references that went on past K block 0: %llu of %llu (%.1f %%)\n", st[12], st[0],
              st[0] ? 100.0 * (double)st[12] / (double)st[0] : 0.0);
    {
      const double tot = (double)(st[5] + st[6] + st[7] + st[8] + st[9] + st[10]);
      fprintf(stderr, "[knn16 stats] wave cycles (s_memtime units, summed over waves; %% of the scan loop): MFMA segments of live steps %.1f  slow path %.1f  "
                      "control tail %.1f  tile wait %.1f  barrier %.1f  steps sat out %.1f   -- per wave and staged tile: %.0f units; live wave-steps %.1f %%\n",
              100.0 * st[5] / tot, 100.0 * st[6] / tot, 100.0 * st[7] / tot, 100.0 * st[8] / tot, 100.0 * st[9] / tot, 100.0 * st[10] / tot,
              tot / (double)st[11], st[11] ? 50.0 * (double)st[0] / (double)st[11] : 0.0);
    }
  }
  return MELD_OK;
}

extern "C" int meld_knn16_topk(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info,
                               int64_t n_ref, int d,
                               int64_t q_count, int ksel, int nprod, int n_slices, const void* lb2,
                               const float* norm2_max, int64_t q_begin, const float* thr_init, int knn,
                               double radius_factor, int32_t* cand_idx, float* cand_d2, int32_t* cand_cnt,
                               float* cand_thr, uint64_t* tiles_done, const int32_t* block_order, meld_stream_t stream) {
  return k16_topk_impl(Q16, Qn, Rt16, scale_info, n_ref, d, q_count, ksel, nprod, n_slices, lb2, norm2_max, q_begin, thr_init, knn,
                       radius_factor, cand_idx, cand_d2, cand_cnt, cand_thr, tiles_done, block_order, nullptr, nullptr, 0, stream);
}

// The hi-only first pass over precomputed step lists (meld_knn16_step_lists) instead of the pruning table: same rows,
// counts and thresholds as meld_knn16_topk(nprod = 1, n_slices, lb2, thr_init) up to the 1.5 % of blocks the per-step
// test against the CURRENT thresholds would have skipped (they hold no candidate: the lists are a superset).
extern "C" int meld_knn16_topk_listed(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info, int64_t n_ref,
                                      int d, int64_t q_count, int ksel, const uint32_t* step_list, const int32_t* step_cnt,
                                      int64_t list_stride, const float* norm2_max, int64_t q_begin, const float* thr_init, int knn,
                                      double radius_factor, int32_t* cand_idx, float* cand_d2, int32_t* cand_cnt, float* cand_thr,
                                      uint64_t* tiles_done, const int32_t* block_order, int n_slices, meld_stream_t stream) {
  MELD_CHECK_ARG(step_list && step_cnt && list_stride > 0, "meld_knn16_topk_listed: null step lists");
  return k16_topk_impl(Q16, Qn, Rt16, scale_info, n_ref, d, q_count, ksel, 1, n_slices, nullptr, norm2_max, q_begin, thr_init, knn,
                       radius_factor, cand_idx, cand_d2, cand_cnt, cand_thr, tiles_done, block_order, step_list, step_cnt, list_stride,
                       stream);
}

// The same with the partial test of the SPLIT layout (meld_knn16_split_dims(d) > 0): partial_test != 0 lets the pass drop a block
// of 32 references behind its first K block when no partial value is within reach of its row -- same rows, counts and
// thresholds; worth asking for when the leading coordinates of the operands carry the distances (principal coordinates).
// tiles_done (optional) then counts two things: [0] the (wave, tile) pairs computed, [1] the blocks of 32 references that went on.
extern "C" int meld_knn16_topk_listed_partial(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info, int64_t n_ref,
                                              int d, int64_t q_count, int ksel, const uint32_t* step_list, const int32_t* step_cnt,
                                              int64_t list_stride, const float* norm2_max, int64_t q_begin, const float* thr_init, int knn,
                                              double radius_factor, int32_t* cand_idx, float* cand_d2, int32_t* cand_cnt, float* cand_thr,
                                              uint64_t* tiles_done, const int32_t* block_order, int n_slices, int partial_test,
                                              meld_stream_t stream) {
  MELD_CHECK_ARG(step_list && step_cnt && list_stride > 0, "meld_knn16_topk_listed_partial: null step lists");
  return k16_topk_impl(Q16, Qn, Rt16, scale_info, n_ref, d, q_count, ksel, 1, n_slices, nullptr, norm2_max, q_begin, thr_init, knn,
                       radius_factor, cand_idx, cand_d2, cand_cnt, cand_thr, tiles_done, block_order, step_list, step_cnt, list_stride,
                       stream, partial_test, 1);
}

// The partial-distance test as a pass over the step lists (knn16_partial_filter_kernel): lists built for operands in the SPLIT
// layout (meld_knn16_split_dims(d) > 0), cells in a frame whose leading coordinates carry the distances.  step_list is rewritten
// in place (entries of waves that fail the test cleared, empty entries dropped), cnt_out[block] = the new length (may alias
// step_cnt), tested (optional) += the (wave, tile) pairs tested.  The search over the thinned lists (meld_knn16_topk_listed)
// returns the rows the search over the full ones would.  thr_init: the start thresholds the lists were built for (scaled units).
extern "C" int meld_knn16_partial_filter(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info, const float* norm2_max,
                                         int d, int64_t q_count, const float* thr_init, uint32_t* step_list, const int32_t* step_cnt,
                                         int64_t list_stride, int32_t* cnt_out, uint64_t* tested, const int32_t* block_order,
                                         meld_stream_t stream) {
  MELD_CHECK_ARG(Q16 && Qn && Rt16 && scale_info && norm2_max && thr_init && step_list && step_cnt && cnt_out && list_stride > 0 && q_count > 0,
                 "meld_knn16_partial_filter: null pointer or bad sizes");
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  const int dA = k16_dA(d, KB);
  MELD_CHECK_ARG(dA > 0 && KB >= 2, "meld_knn16_partial_filter: d = %d has no split operand layout", d);
  const unsigned n_blocks = (unsigned)ceil_div(q_count, K16_BQ);
  int abl = 0;
#ifdef K16_PROFILING
  if (const char* e = meld_dev_getenv("MELD_KNN_FILTER_ABL")) abl = atoi(e);
#endif
  hipLaunchKernelGGL((knn16_partial_filter_kernel<8, 4>), dim3(n_blocks), dim3(256), 0, S(stream), reinterpret_cast<const _Float16*>(Q16), Qn,
                     reinterpret_cast<const _Float16*>(Rt16), scale_info, norm2_max, thr_init, step_list, step_cnt, (long long)list_stride, cnt_out,
                     reinterpret_cast<unsigned long long*>(tested), block_order, KB, 16 + d - dA, abl);
  MELD_LAUNCH_CHECK("knn16_partial_filter_kernel");
  return MELD_OK;
}

// Workgroups of the search kernel that are resident on the device at once (occupancy x CUs): the
// host uses it to avoid a nearly empty last wave of workgroups (it hands the remainder to a sliced
// launch instead).
extern "C" int meld_knn16_resident_blocks(int d, int nprod) {
  MELD_CHECK_ARG(nprod == 1 || nprod == 3, "meld_knn16_resident_blocks: nprod must be 1 or 3");
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  int per_cu = 0;
#define K16_OCC(KBV)                                                                                              \
  case KBV:                                                                                                       \
    if (nprod == 1) {                                                                                             \
      MELD_HIP_CALL(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, knn16_topk_kernel<KBV, 0, 1>, K16_THREADS, 0)); \
    } else {                                                                                                      \
      MELD_HIP_CALL(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, knn16_topk_kernel<KBV, 0, 3>, K16_THREADS, 0)); \
    }                                                                                                             \
    break;
  switch (KB) {
    K16_OCC(4)
#ifndef K16_DEV_KB4
    K16_OCC(1)
    K16_OCC(2)
    K16_OCC(3)
    K16_OCC(5)
    K16_OCC(6)
    K16_OCC(7)
    K16_OCC(8)
    K16_OCC(9)
#endif
    default:
      return MELD_ERR_UNSUPPORTED;
  }
#undef K16_OCC
  int dev = 0, cus = 0;
  MELD_HIP_CALL(hipGetDevice(&dev));
  MELD_HIP_CALL(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  return per_cu * cus;
}

extern "C" int meld_knn16_max_slices(int ksel) { return ksel < 1 ? MELD_ERR_INVALID : std::max(1, K16_MERGE_MAX / ksel); }

extern "C" int meld_knn16_merge_slices(const int32_t* s_idx, const float* s_d2, const int32_t* s_cnt, int64_t q_count,
                                       int ksel, int n_slices, int32_t* out_idx, float* out_d2, int32_t* out_cnt,
                                       meld_stream_t stream) {
  MELD_CHECK_ARG(s_idx && s_d2 && s_cnt && out_idx && out_d2 && out_cnt && q_count > 0, "meld_knn16_merge_slices: null");
  const int cap = meld_knn16_row_capacity(ksel);
  if (cap < 0) return cap;
  MELD_CHECK_ARG(n_slices >= 1 && n_slices * ksel <= K16_MERGE_MAX, "meld_knn16_merge_slices: too many slices");
  const int q_pad = (int)(ceil_div(q_count, K16_BQ) * K16_BQ);
  hipLaunchKernelGGL(knn16_merge_slices_kernel, dim3((unsigned)q_count), dim3(64), (size_t)n_slices * ksel * 8, S(stream), s_idx, s_d2,
                     s_cnt, (int)q_count, q_pad, ksel, cap, n_slices, out_idx, out_d2, out_cnt);
  MELD_LAUNCH_CHECK("knn16_merge_slices_kernel");
  return MELD_OK;
}
