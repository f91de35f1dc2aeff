// spmm_tiled.hip -- the graph-Laplacian recurrence step on a panel-tiled copy of W.
//
// Same operator as spmm.hip (one fused step of [UPSTREAM pygsp cheby_op], called from reference
// meld/filter.py:59, and the SpMV of the lmax estimate, meld/filter.py:39):
//     y = alpha * (dw .* x - W x) + beta * x + gamma * z ;   r += coef * y
// but the iterate is never gathered through the vector cache.  The CSR-stream kernel of spmm.hip is
// bound by L1 -> L2 *requests*: one 16-byte gather per nonzero, two thirds of which miss the 32 KiB L1
// (DESIGN section 4.4), 3.5e7 requests per step for 0.6 GB of compulsory traffic.  W is static over
// the ~75 steps of a fit_transform (45 Lanczos + 29 Chebyshev), so it is re-laid out once:
//
//   * rows are cut into nb blocks of <= RMAX rows with balanced nonzero counts, ONE block per CU;
//   * for every block the sorted list of the distinct columns its rows touch is cut into tiles of
//     CP = 2048 columns; a tile's slice of the iterate (32 KiB at p = 2) is staged in LDS by two
//     loader waves while the 14 consumer waves work on the previous tile (double buffered);
//   * the block's nonzeros are stored tile by tile as (fp64 value, 11-bit tile-local column,
//     9-bit row slot): a consumer wave streams ITS OWN contiguous slice of the block (rows are dealt
//     to the waves round-robin, so every wave has the same share of every tile), reads x from LDS
//     (ds_read_b128) and accumulates into LDS accumulators it alone owns (ds_add_f64) -- the order of
//     the additions into any row is fixed by the layout, results are bit-reproducible;
//   * every staged column is used by at least one nonzero of the block (5.8 on average at 1M cells),
//     so the L2 sees one 16-byte gather per DISTINCT column of a block (2.8e4 per block) instead of
//     one per nonzero (1.6e5), plus the coalesced matrix stream.
//
// HBM traffic per step is the 12 B per nonzero of the CSR form plus the vector passes; the kernel is
// bound by that stream (roofline: HBM).  Layout construction (pt_build_kernel) costs about as much as
// two steps and runs once per graph.
#include "common.hpp"

#include <algorithm>

namespace meld {
namespace pt {

constexpr int NW = 12;                    // consumer waves = row owners
constexpr int NL = 4;                     // loader waves
constexpr int THREADS = 64 * (NW + NL);   // 1024: one workgroup per CU
constexpr int SLOTS = 384;                // accumulator rows per consumer wave
constexpr int RMAX = NW * SLOTS;          // 4480 rows per block at most
constexpr int CP = 1024;                  // columns per staged tile
constexpr int CP_BITS = 10;
constexpr int NB = 4;                     // ring of staged tiles in LDS (NB * CP columns of the iterate)
constexpr int TMAX = 63;                  // tiles per block at most (64512 distinct columns)
constexpr int SEGW = TMAX + 1;            // segment offsets per consumer wave: one per lane of a wave
constexpr int BP = 2048;                  // columns per bitmap panel of the builder (independent of the tile size)
constexpr int BP_BITS = 11;
constexpr int NPAN_MAX = 4096;            // bitmap panels the builder can index (n_cols <= 8.4 M)
constexpr int TP_MAX = 384;               // distinct bitmap panels one block may touch (bitmap rows in LDS)
constexpr int DOT_SLOTS = 64;             // == spmm.hip

static_assert(SLOTS <= 512, "row slot must fit 9 bits");
static_assert((NB & (NB - 1)) == 0, "NB");
static_assert((1 << CP_BITS) == CP && (1 << BP_BITS) == BP && BP / 32 == 64, "CP / BP");

// lgkmcnt(0) only: LDS operations of this wave have completed; global loads stay in flight
#define PT_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xC07F)

struct StepArgs {
  const int32_t* blk_row;    // [nb + 1]
  const int32_t* blk_ntile;  // [nb]
  const int32_t* blk_ndist;  // [nb]
  const int32_t* seg;        // [nb][NW][SEGW] offsets relative to the block's first entry
  const int32_t* list_cols;  // [nnz] block b's sorted distinct columns start at rowptr[blk_row[b]]
  const double* pval;        // [nnz] values in (block, wave, tile, row, col) order
  const float* pval32;       // [nnz] the same rounded to fp32 (F32 instantiation: the lmax estimate's SpMV)
  const uint32_t* pidx;      // [nnz] tile-local column | row slot << 11
  const int64_t* rowptr;     // CSR row pointers (entry base of a block)
  const double* dw;
  const double* x_full;
  const double* z;
  double* y;
  double* r;
  double* dots;
  const double* coef_dev;
  int64_t x_row_offset;
  double alpha, beta, gamma, coef;
  int nb;
  int ld, colofs;
  int ablate;  // timing-only modes (results wrong): 4 no panel loads, 8 panel gathers from a 16 KB window of x,
               // 16 consumers do not wait for the panels  (1 / 2, no accumulator updates / no LDS gather, were removed
               // from the hot loop once measured: DESIGN section 4.4)
};

template <int P>
struct V {
  double v[P];
};

template <int P>
__device__ __forceinline__ V<P> ldg(const double* __restrict__ base, int64_t row, int ld, int colofs) {
  V<P> o;
  const double* p = base + row * ld + colofs;
  if constexpr (P == 1) {
    o.v[0] = p[0];
  } else {
    const double2 t = *reinterpret_cast<const double2*>(p);
    o.v[0] = t.x;
    o.v[1] = t.y;
  }
  return o;
}
template <int P>
__device__ __forceinline__ void stg(double* __restrict__ base, int64_t row, int ld, int colofs, const V<P>& a) {
  double* p = base + row * ld + colofs;
  if constexpr (P == 1) {
    p[0] = a.v[0];
  } else {
    *reinterpret_cast<double2*>(p) = make_double2(a.v[0], a.v[1]);
  }
}

// LDS fp64 add without a return value (ds_add_f64)
__device__ __forceinline__ void lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ double& pr0_sink(const double& x) { return const_cast<double&>(x); }

constexpr int U = 8;  // entry chunks (64 nonzeros each) in flight per consumer wave
static_assert(U == 8, "the consumer stream names its 8 slots (v96..v119) and waits with vmcnt(2 (U - 1))");

// LDS words shared by the waves of a workgroup, polled / bumped with plain LDS operations (the CU's LDS is
// coherent for its own waves; a wave's LDS operations complete in order)
__device__ __forceinline__ int lds_peek(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_bump(int* p) { __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// F32: the matrix VALUES are streamed from the fp32 copy (8 instead of 12 bytes per nonzero); vectors, products and
// sums stay fp64.  Only the Lanczos SpMV of the lmax estimate uses it: rounding W to fp32 moves the largest
// eigenvalue by < 1e-7 relative, far inside the tolerance that estimate is computed to (and the 1.01 factor on it).
template <int P, bool F32>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_num_vgpr(96))) void pt_step_kernel(StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* acc = lds;               // [NW][SLOTS][P]
  double* pans = lds + RMAX * P;   // [NB][CP][P] ring of staged tiles of the iterate
  __shared__ double s_dot[2][NW + NL];
  // ring state: s_prod[i] = tiles staged into buffer i so far (monotonic: buffer i holds tile t, t % NB == i,
  // once s_prod[i] == t / NB + 1; a tile is staged by ONE loader wave); s_cons[i] = consumer waves that have left a
  // tile of buffer i (tile t may be overwritten once s_cons[i] == NW * (t / NB + 1))
  __shared__ int s_prod[NB], s_cons[NB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-contiguous row blocks (workgroup b runs on XCD b % 8; placement used for L2 locality only)
  const int nbp = gridDim.x;  // multiple of 8
  const int per = nbp >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= a.nb) return;  // padding workgroup (whole workgroup leaves: no barrier is left hanging)

  double alpha = a.alpha, gamma = a.gamma;
  if (a.coef_dev != nullptr) {  // device-resident Lanczos: scalars written by the previous iteration
    alpha = a.coef_dev[3];
    gamma = a.coef_dev[4];
  }
  const int ab = a.ablate;
  const int row0 = a.blk_row[b];
  const int nrows = a.blk_row[b + 1] - row0;
  if (nrows == 0) return;  // (uniform) an empty block has nothing to stage, accumulate or write
  const int T = a.blk_ntile[b];
  const int64_t ebase = a.rowptr[row0];
  if (tid < NB) {
    s_prod[tid] = 0;
    s_cons[tid] = 0;
  }
  __syncthreads();

  // No workgroup barrier from here to the end of the block's tiles: a consumer wave waits only for the loaders
  // (tile t staged), the loaders only for the slowest consumer NB tiles back.
  if (w < NW) {
    // ------------------------------------------------------------------ consumer wave
    double* myacc = acc + w * SLOTS * P;
    for (int i = lane; i < SLOTS * P; i += 64) myacc[i] = 0.0;
    // the wave's 64 segment offsets live in one VGPR (lane t holds sg[t]); segment bounds are read with
    // v_readlane, so that the cursor arithmetic below is pure SALU and touches no memory
    const int sgv = a.seg[((size_t)b * (NW + 1) + w) * SEGW + lane];
    auto sg = [&](int t) __attribute__((always_inline)) { return __builtin_amdgcn_readlane(sgv, t); };
    const double* pv = a.pval + ebase;
    const float* pv32 = a.pval32 + ebase;
    const uint32_t* pi = a.pidx + ebase;
    const int e_begin = sg(0);
    const int e_end = sg(T);
    // prefetch cursor over the flattened (tile, chunk) sequence; everything here is wave-uniform
    int tp = 0;
    int pe = e_begin;
    int pend = (T > 0) ? sg(1) : e_begin;
    while (tp < T && pe >= pend) {  // skip empty segments
      ++tp;
      pend = (tp < T) ? sg(tp + 1) : pend;
    }
    // ---- entry stream: U chunks (64 entries each) in flight per wave, wait counts by hand -----------------------
    // hipcc's wait-count pass forgets the order of the loads pending over a loop's back edge: whatever the shape of
    // the loop, the first chunk of every unrolled round got s_waitcnt vmcnt(0..2), i.e. the wave drained its queue
    // every U chunks and its own latency chain (not HBM) set the pace (132 us per step).  The stream loads are
    // therefore issued from inline asm, which the pass does not see, and waited for with explicit counts: loads
    // return in order and every slot is re-issued unconditionally (clamped address) right after it has been read,
    // so when chunk k is wanted exactly 2 (U - 1) younger loads are in flight.  A register with a load in flight
    // must never be copied, and the register allocator copies freely (loop phis, tied operands), so the slots are
    // PHYSICAL registers the compiler does not own: the kernel is limited to v0..v95 (amdgpu_num_vgpr) and the
    // asm names v96..v119 itself (values v[96 + 2u : 97 + 2u], packed indices v[112 + u]).
    int ct[U], cn[U];
    const int e_last = max(e_end - 1, e_begin);
#ifndef PT_NT
#define PT_NT " nt"
#endif
#define PT_SLOT_LOAD(VLO, VHI, IX)                                                                          \
  asm volatile("global_load_dwordx2 v[" #VLO ":" #VHI "], %0, %2" PT_NT "\n\tglobal_load_dword v" #IX ", %1, %3" PT_NT \
               :                                                                                            \
               : "v"(off8), "v"(off4), "s"(pv), "s"(pi)                                                     \
               : "memory", "v" #VLO, "v" #VHI, "v" #IX)
#define PT_SLOT_LOAD32(VLO, VHI, IX)                                                                   \
  asm volatile("global_load_dword v" #VLO ", %0, %1" PT_NT "\n\tglobal_load_dword v" #IX ", %0, %2" PT_NT        \
               :                                                                                         \
               : "v"(off4), "s"(pv32), "s"(pi)                                                           \
               : "memory", "v" #VLO, "v" #VHI, "v" #IX)
#define PT_SLOT_TAKE(VLO, VHI, IX)                                                                                     \
  asm volatile("s_waitcnt vmcnt(14)\n\tv_mov_b32 %0, v" #VLO "\n\tv_mov_b32 %1, v" #VHI "\n\tv_mov_b32 %2, v" #IX \
               : "=v"(t_lo), "=v"(t_hi), "=v"(t_ix)                                                                    \
               :                                                                                                       \
               : "memory")
    auto issue = [&](int u) __attribute__((always_inline)) {
      const int n = (tp < T) ? min(64, pend - pe) : 0;
      const int e = min(pe + lane, e_last);
      const unsigned off8 = (unsigned)e * 8u, off4 = (unsigned)e * 4u;
      if constexpr (F32) {
        switch (u) {  // (u is a constant after unrolling)
          case 0: PT_SLOT_LOAD32(96, 97, 112); break;
          case 1: PT_SLOT_LOAD32(98, 99, 113); break;
          case 2: PT_SLOT_LOAD32(100, 101, 114); break;
          case 3: PT_SLOT_LOAD32(102, 103, 115); break;
          case 4: PT_SLOT_LOAD32(104, 105, 116); break;
          case 5: PT_SLOT_LOAD32(106, 107, 117); break;
          case 6: PT_SLOT_LOAD32(108, 109, 118); break;
          default: PT_SLOT_LOAD32(110, 111, 119); break;
        }
      } else {
        switch (u) {
          case 0: PT_SLOT_LOAD(96, 97, 112); break;
          case 1: PT_SLOT_LOAD(98, 99, 113); break;
          case 2: PT_SLOT_LOAD(100, 101, 114); break;
          case 3: PT_SLOT_LOAD(102, 103, 115); break;
          case 4: PT_SLOT_LOAD(104, 105, 116); break;
          case 5: PT_SLOT_LOAD(106, 107, 117); break;
          case 6: PT_SLOT_LOAD(108, 109, 118); break;
          default: PT_SLOT_LOAD(110, 111, 119); break;
        }
      }
      ct[u] = tp;
      cn[u] = n;
      pe += n;
      while (tp < T && pe >= pend) {
        ++tp;
        pend = (tp < T) ? sg(tp + 1) : pend;
      }
    };
    // wait for the chunk in slot u and copy it out of the slot
    auto take = [&](int u, double& v, uint32_t& ix) __attribute__((always_inline)) {
      int t_lo, t_hi;
      uint32_t t_ix;
      switch (u) {
        case 0: PT_SLOT_TAKE(96, 97, 112); break;
        case 1: PT_SLOT_TAKE(98, 99, 113); break;
        case 2: PT_SLOT_TAKE(100, 101, 114); break;
        case 3: PT_SLOT_TAKE(102, 103, 115); break;
        case 4: PT_SLOT_TAKE(104, 105, 116); break;
        case 5: PT_SLOT_TAKE(106, 107, 117); break;
        case 6: PT_SLOT_TAKE(108, 109, 118); break;
        default: PT_SLOT_TAKE(110, 111, 119); break;
      }
      v = F32 ? (double)__int_as_float(t_lo) : __hiloint2double(t_hi, t_lo);
      ix = t_ix;
    };
    int cur = -1;  // the tile this wave holds (-1: none yet); tiles are entered in order, every one exactly once
    // enter tile `to` (> cur): leave the held tile, pass through the tiles in between (no entries of this wave:
    // they are still waited for and signalled, in order, so that the ring counters never mix two uses of a buffer)
    auto advance = [&](int to) __attribute__((always_inline)) {
      while (cur < to) {
        if (cur >= 0) {
          PT_WAIT_LDS();  // my reads of the panel have completed
          if (lane == 0) lds_bump(&s_cons[cur % NB]);
        }
        ++cur;
        if (cur < T && !(ab & 16)) {
          const int need = cur / NB + 1;
          while (lds_peek(&s_prod[cur % NB]) < need) __builtin_amdgcn_s_sleep(1);
        }
      }
    };
    int srow = -1;  // the row whose run this lane is summing (none yet)
    double s0 = 0.0, s1 = 0.0;
    // A lane's consecutive chunks hold consecutive entries of the segment in (row, column) order (the segment is
    // stored transposed over the wave), i.e. runs of the same row: the run is summed in registers and only a change
    // of row goes to the LDS accumulator -- one ds_add per run and lane instead of one per entry, issued with the
    // few lanes whose row changed (hardly any bank conflict).
    auto accumulate = [&](bool act, double v, int sl, const V<P>& xv) __attribute__((always_inline)) {
      if (act) {
        // (the products first: their LDS wait then precedes the flush below instead of also covering it)
        const double pr0 = v * xv.v[0];
        double pr1 = 0.0;
        if constexpr (P == 2) pr1 = v * xv.v[1];
        asm volatile("" : "+v"(pr0_sink(pr0)), "+v"(pr0_sink(pr1)));
        const bool same = sl == srow;
#ifndef PT_ABL
#define PT_ABL 0
#endif
        if (!(PT_ABL & 1) && !same && srow >= 0) {
          lds_add(myacc + P * srow, s0);
          if constexpr (P == 2) lds_add(myacc + 2 * srow + 1, s1);
        }
        s0 = same ? s0 + pr0 : pr0;
        if constexpr (P == 2) s1 = same ? s1 + pr1 : pr1;
        srow = sl;
      }
    };
    if (e_end > e_begin) {
      // (no separate prologue: the loop starts U steps early on empty slots, so that the very same instructions --
      // and registers -- issue the first loads; a prologue of its own gets its own register assignment and the
      // copies into the loop's registers would read slots whose loads are still in flight)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        ct[u] = -1;
        cn[u] = 0;
      }
      // two-stage pipeline over the chunks: stage 1 of chunk k + 1 (wait for its entries, re-issue its slot, enter
      // its tile if it is a new one, request its x values from LDS) runs before stage 2 of chunk k (products, run
      // sums, accumulator updates), so the LDS gather latency is covered by a chunk's worth of arithmetic
      bool h_act = false;  // the chunk waiting in stage 2 (none yet: no active lane)
      double h_v = 0.0;
      int h_sl = 0;
      V<P> h_x;
#pragma unroll
      for (int c = 0; c < P; ++c) h_x.v[c] = 0.0;
      // chunks of this wave (known up front: the loop has a plain trip count and no exit in its body, which keeps the
      // eight slots in eight fixed registers -- see tests/test_kernel_resources.py)
      int K = 0;
      for (int t = 0; t < T; ++t) K += (sg(t + 1) - sg(t) + 63) >> 6;
      for (int k0 = -U; k0 < K; k0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          // ---- stage 1 of the chunk in slot u (past the end of the stream: an empty chunk, n == 0)
          const int n = cn[u], t = ct[u];
          double v;
          uint32_t ix;
          take(u, v, ix);  // (in the first round: whatever the slot registers held; n == 0, nothing is used)
          issue(u);  // the slot is free again: chunk k + U
          // ---- stage 2 of the previous chunk FIRST: its x values were requested a whole step ago, so the LDS wait
          // in front of the products is free; the gather of this chunk goes out at the END of the step and has the
          // next step's entry wait to land (requested before the products, the wait for the previous gather also
          // waited for the one just issued: a full LDS round trip per step)
          accumulate(h_act, h_v, h_sl, h_x);
          if (cur < t) advance(t);
          const double* pan = pans + (size_t)(t & (NB - 1)) * CP * P;
          const int cl = (PT_ABL & 2) ? lane : (int)(ix & (CP - 1));
          const bool act = lane < n;
          V<P> xv;
          if constexpr (P == 1) {
            xv.v[0] = pan[act ? cl : 0];
          } else {
            const double2 t2 = *reinterpret_cast<const double2*>(pan + 2 * (act ? cl : 0));
            xv.v[0] = t2.x;
            xv.v[1] = t2.y;
          }
          h_act = act;
          h_v = v;
          h_sl = (int)(ix >> CP_BITS);
          h_x = xv;
        }
      }
      accumulate(h_act, h_v, h_sl, h_x);
      // the slots still have loads in flight (clamped re-issues past the end of the stream)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef PT_SLOT_LOAD
#undef PT_SLOT_LOAD32
#undef PT_SLOT_TAKE
    if (srow >= 0) {  // the last run of every lane
      lds_add(myacc + P * srow, s0);
      if constexpr (P == 2) lds_add(myacc + 2 * srow + 1, s1);
    }
    advance(T);  // leave the held tile and pass through the rest
  } else {
    // ------------------------------------------------------------------ loader wave
    // Loader wave lw stages the tiles t = lw, lw + NL, ... on its own (64 lanes x PER columns), so NL tiles are in
    // flight besides what the ring holds; the gathered values of its next tile and the column list of the one after
    // are requested before it waits for the ring buffer of the current one.  Loads are unconditional from clamped
    // positions: slots beyond the block's last distinct column are never referenced by an entry.
    const int lw = w - NW;
    const int ndist = a.blk_ndist[b];
    const int32_t* lst = a.list_cols + ebase;
    const int srcv = a.seg[((size_t)b * (NW + 1) + NW) * SEGW + lane];  // lane j: list chunk of the j-th processed tile
    constexpr int PER = CP / 64;  // columns per lane and tile
    int col[PER];
    V<P> xv[PER];
    auto load_list = [&](int t) __attribute__((always_inline)) {
      const int chunk = __builtin_amdgcn_readlane(srcv, t);
      const int n = min(CP, ndist - chunk * CP);
#pragma unroll
      for (int k = 0; k < PER; ++k) col[k] = lst[chunk * CP + min(k * 64 + lane, n - 1)];
    };
    auto gather = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < PER; ++k) xv[k] = ldg<P>(a.x_full, (ab & 8) ? (col[k] & 1023) : col[k], a.ld, a.colofs);
    };
    if (lw < T && !(ab & 4)) {
      load_list(lw);
      gather();
      if (lw + NL < T) load_list(lw + NL);
    }
    for (int t = lw; t < T; t += NL) {
      const int buf = t % NB;
      const int use = t / NB;
      if (use > 0) {  // every consumer has left tile t - NB
        while (lds_peek(&s_cons[buf]) < NW * use) __builtin_amdgcn_s_sleep(1);
      }
      if (!(ab & 4)) {
        double* pan = pans + (size_t)buf * CP * P;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
          const int g = k * 64 + lane;
          if constexpr (P == 1) {
            pan[g] = xv[k].v[0];
          } else {
            *reinterpret_cast<double2*>(pan + 2 * g) = make_double2(xv[k].v[0], xv[k].v[1]);
          }
        }
        if (t + NL < T) gather();                   // my next tile (its list landed one tile ago)
        if (t + 2 * NL < T) load_list(t + 2 * NL);  // and the list of the one after
      }
      PT_WAIT_LDS();  // the panel has landed
      if (lane == 0) lds_bump(&s_prod[buf]);
    }
  }

  // ---------------------------------------------------------------------- epilogue (all 16 waves)
  // operands of the block's rows are requested before the closing barrier, consumed after it
  constexpr int NR = (RMAX + THREADS - 1) / THREADS;
  V<P> xl[NR], zl[NR], rl_[NR];
  double dwi[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int rl = min(tid + q * THREADS, max(nrows - 1, 0));
    const int64_t row = row0 + rl;
    xl[q] = ldg<P>(a.x_full, a.x_row_offset + row, a.ld, a.colofs);
    dwi[q] = a.dw[row];
#pragma unroll
    for (int c = 0; c < P; ++c) zl[q].v[c] = rl_[q].v[c] = 0.0;
    if (gamma != 0.0) zl[q] = ldg<P>(a.z, row, a.ld, a.colofs);
    if (a.r != nullptr) rl_[q] = ldg<P>(a.r, row, a.ld, a.colofs);
  }
  PT_WAIT_LDS();
  __builtin_amdgcn_s_barrier();  // every accumulator is final
  double d_yx = 0.0, d_yy = 0.0;
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int rl = tid + q * THREADS;
    if (rl < nrows) {
      const int64_t row = row0 + rl;
      const int ow = rl % NW, sl = rl / NW;
      const double* ap = acc + (ow * SLOTS + sl) * P;
      V<P> yv;
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const double lx = dwi[q] * xl[q].v[c] - ap[c];  // (L x)_i
        yv.v[c] = alpha * lx + a.beta * xl[q].v[c] + gamma * zl[q].v[c];
        rl_[q].v[c] += a.coef * yv.v[c];
        d_yx += yv.v[c] * xl[q].v[c];
        d_yy += yv.v[c] * yv.v[c];
      }
      stg<P>(a.y, row, a.ld, a.colofs, yv);
      if (a.r != nullptr) stg<P>(a.r, row, a.ld, a.colofs, rl_[q]);
    }
  }
  if (a.dots != nullptr) {
    d_yx = wave_sum(d_yx);
    d_yy = wave_sum(d_yy);
    if (lane == 0) {
      s_dot[0][w] = d_yx;
      s_dot[1][w] = d_yy;
    }
    __syncthreads();
    if (tid < 2) {
      double s = 0.0;
      for (int k = 0; k < NW + NL; ++k) s += s_dot[tid][k];
      atomicAdd(&a.dots[tid * DOT_SLOTS + (b % DOT_SLOTS)], s);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Layout construction
// ------------------------------------------------------------------------------------------------

// Row-block boundaries with balanced nonzero counts: every thread finds the cut of its blocks by bisection
// (a single lane doing nb x 20 dependent loads took 3.6 ms at 1M cells), then one lane applies the caps
// (a block holds at most RMAX rows; what is left has to fit the remaining blocks) in order.
__global__ __launch_bounds__(1024) void pt_plan_kernel(const int64_t* __restrict__ rowptr, int64_t n_rows, int nb,
                                                       int32_t* __restrict__ blk_row) {
  const int64_t nnz = rowptr[n_rows];
  for (int b = threadIdx.x; b < nb; b += blockDim.x) {
    const int64_t target = (nnz / nb) * (b + 1) + ((nnz % nb) * (b + 1)) / nb;
    int64_t lo = 0, hi = n_rows;
    while (lo < hi) {  // first row index whose prefix reaches the target
      const int64_t mid = (lo + hi) >> 1;
      if (rowptr[mid] < target) lo = mid + 1; else hi = mid;
    }
    blk_row[b + 1] = (int32_t)((b == nb - 1) ? n_rows : lo);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t prev = 0;
    blk_row[0] = 0;
    for (int b = 0; b < nb; ++b) {
      int64_t cut = blk_row[b + 1];
      const int64_t must = n_rows - (int64_t)(nb - 1 - b) * RMAX;
      cut = max(cut, must);
      cut = min(cut, prev + RMAX);
      cut = min(max(cut, prev), n_rows);
      blk_row[b + 1] = (int32_t)cut;
      prev = cut;
    }
  }
}

// exclusive scan of one int per thread over the workgroup (THREADS threads); returns the prefix, total in *total
__device__ __forceinline__ int block_exscan(int v, int* s_wave /* [THREADS/64 + 1] */, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  __syncthreads();
  if (lane == 63) s_wave[w] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int k = 0; k < THREADS / 64; ++k) {
      const int t = s_wave[k];
      s_wave[k] = run;
      run += t;
    }
    s_wave[THREADS / 64] = run;
  }
  __syncthreads();
  *total = s_wave[THREADS / 64];
  return s_wave[w] + inc - v;
}

constexpr int GW = 4;                            // bitmap words per prefix group
constexpr int GPP = (BP / 32) / GW;              // groups per bitmap panel (16)
constexpr int GROUPS_MAX = TP_MAX * GPP;
constexpr int GPT = (GROUPS_MAX + THREADS - 1) / THREADS;  // groups per thread in the prefix pass

// One workgroup per row block: distinct-column list, per-(wave, tile) segment offsets, and the block's
// nonzeros re-ordered by (owner wave, tile, row, column).  status[0] = max over blocks of an error code
// (1: a block touches more than TP_MAX column panels, 2: more than TMAX tiles, 3: n_cols too large).
__global__ __launch_bounds__(THREADS) void pt_build_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, const double* __restrict__ val, int64_t n_cols,
    int nb, const int32_t* __restrict__ blk_row, int32_t* __restrict__ blk_ntile, int32_t* __restrict__ blk_ndist,
    int32_t* __restrict__ seg, int32_t* __restrict__ list_cols, uint32_t* __restrict__ codes, int32_t* __restrict__ status,
    int stop /* timing only: leave after stage `stop` (0 = run everything) */) {
  __shared__ int16_t s_pmap[NPAN_MAX];          // panel -> compact index of the touched panels (ascending), -1
  __shared__ uint32_t s_bits[TP_MAX][BP / 32];  // one bit per column of every touched panel
  __shared__ int32_t s_gpre[GROUPS_MAX + 1];    // distinct columns before each group of GW bitmap words
  __shared__ int16_t s_cpan[TP_MAX];            // compact index -> panel
  __shared__ int32_t s_cnt[NW][SEGW];           // entries per (wave, tile), then running cursors
  __shared__ int s_scan[THREADS / 64 + 1];
  __shared__ int32_t s_rp[RMAX + 1];            // row pointers of the block, relative to its first entry
  __shared__ int s_ord[SEGW];                    // processing order of the tiles: j -> chunk of the column list
  __shared__ int s_tmp[3][SEGW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int b = blockIdx.x;
  const int row0 = blk_row[b];
  const int nrows = blk_row[b + 1] - row0;
  const int64_t e0 = rowptr[row0];
  const int64_t e1 = rowptr[row0 + nrows];
  const int npan = (int)((n_cols + BP - 1) / BP);

  for (int i = tid; i < NPAN_MAX; i += THREADS) s_pmap[i] = 0;
  for (int i = tid; i < TP_MAX * (BP / 32); i += THREADS) (&s_bits[0][0])[i] = 0u;
  for (int i = tid; i < NW * SEGW; i += THREADS) (&s_cnt[0][0])[i] = 0;
  for (int i = tid; i <= nrows; i += THREADS) s_rp[i] = (int32_t)(rowptr[row0 + i] - e0);
  __syncthreads();

  // pass A: which column panels does this block touch?  (8 loads in flight per thread: the passes over the
  // block's 0.6 MB of column indices are latency-bound otherwise)
  constexpr int MU = 8;
  for (int64_t eb = e0 + tid; eb < e1; eb += (int64_t)MU * THREADS) {
    int c[MU];
#pragma unroll
    for (int u = 0; u < MU; ++u) c[u] = col[min(eb + (int64_t)u * THREADS, e1 - 1)];
#pragma unroll
    for (int u = 0; u < MU; ++u) s_pmap[c[u] >> BP_BITS] = 1;  // (the clamped tail re-marks a valid entry)
  }
  __syncthreads();
  {  // compact index in ascending panel order (4 panels per thread)
    int f[4], c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pn = tid * 4 + k;
      f[k] = (pn < npan) ? s_pmap[pn] : 0;
      c += f[k];
    }
    int total;
    int pre = block_exscan(c, s_scan, &total);
    if (total > TP_MAX) {
      if (tid == 0) {
        atomicMax(status, 1);
        blk_ntile[b] = -1;
        blk_ndist[b] = 0;
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pn = tid * 4 + k;
      if (pn < NPAN_MAX) {
        if (f[k]) {
          s_pmap[pn] = (int16_t)pre;
          s_cpan[pre] = (int16_t)pn;
          ++pre;
        } else {
          s_pmap[pn] = -1;
        }
      }
    }
    if (tid == 0) s_scan[0] = total;
  }
  __syncthreads();
  const int tp = s_scan[0];
  __syncthreads();

  // pass B: one bit per distinct column
  for (int64_t eb = e0 + tid; eb < e1; eb += (int64_t)MU * THREADS) {
    int c[MU];
#pragma unroll
    for (int u = 0; u < MU; ++u) c[u] = col[min(eb + (int64_t)u * THREADS, e1 - 1)];
#pragma unroll
    for (int u = 0; u < MU; ++u) atomicOr(&s_bits[s_pmap[c[u] >> BP_BITS]][(c[u] & (BP - 1)) >> 5], 1u << (c[u] & 31));
  }
  __syncthreads();
  if (stop == 1) return;
  int ndist;
  {  // distinct columns before every group of GW words (<= GROUPS_MAX groups, GPT consecutive ones per thread)
    const int ngroups = tp * GPP;
    int cntk[GPT], c = 0;
#pragma unroll
    for (int k = 0; k < GPT; ++k) {
      const int g = tid * GPT + k;
      int sgrp = 0;
      if (g < ngroups) {
        const uint32_t* wp = &s_bits[g / GPP][(g % GPP) * GW];
#pragma unroll
        for (int q = 0; q < GW; ++q) sgrp += __popc(wp[q]);
      }
      cntk[k] = sgrp;
      c += sgrp;
    }
    int pre = block_exscan(c, s_scan, &ndist);
#pragma unroll
    for (int k = 0; k < GPT; ++k) {
      const int g = tid * GPT + k;
      if (g < ngroups) s_gpre[g] = pre;
      pre += cntk[k];
    }
    if (tid == 0) s_gpre[ngroups] = ndist;
  }
  __syncthreads();
  const int T = (ndist + CP - 1) / CP;
  if (T > TMAX) {
    if (tid == 0) {
      atomicMax(status, 2);
      blk_ntile[b] = -1;
      blk_ndist[b] = 0;
    }
    return;
  }
  // rank of a column among the block's distinct columns
  auto col_rank = [&](int c) -> int {
    const int ci = s_pmap[c >> BP_BITS];
    const int wd = (c & (BP - 1)) >> 5;
    int g = s_gpre[ci * GPP + wd / GW];
    const uint32_t* wp = &s_bits[ci][wd & ~(GW - 1)];
    // words of the group below wd, branch-free (GW - 1 reads, masked)
#pragma unroll
    for (int q = 0; q < GW - 1; ++q) g += (q < (wd & (GW - 1))) ? __popc(wp[q]) : 0;
    g += __popc(s_bits[ci][wd] & ((1u << (c & 31)) - 1u));
    return g;
  };
  // the sorted list of distinct columns (thread per bitmap word)
  for (int i = tid; i < tp * (BP / 32); i += THREADS) {
    const int ci = i / (BP / 32), wd = i % (BP / 32);
    uint32_t bits = s_bits[ci][wd];
    if (bits) {
      int g = s_gpre[ci * GPP + wd / GW];
      const uint32_t* wp = &s_bits[ci][wd & ~(GW - 1)];
      for (int q = 0; q < (wd & (GW - 1)); ++q) g += __popc(wp[q]);
      const int cbase = ((int)s_cpan[ci] << BP_BITS) + wd * 32;
      while (bits) {
        const int bit = __ffs(bits) - 1;
        bits &= bits - 1;
        list_cols[e0 + g] = cbase + bit;
        ++g;
      }
    }
  }
  if (stop == 2) return;
  // The walk: every owner wave goes through ITS rows in order, lanes = the row's entries (one coalesced load per
  // work item, RU items in flight), and counts its entries per tile.  A row's entries are sorted by column, so the
  // entries of one tile are a run of consecutive lanes; the per-tile counters of a wave live in ONE VGPR (lane t =
  // tile t; T <= 63) that the entry lanes read with ds_bpermute and the run heads update with ds_permute -- no loop
  // over the tiles of a row, no LDS memory in the dependent chain.  Every entry leaves a 32-bit code behind
  // (list chunk | position within its (wave, tile) segment in (row, column) order | tile-local column), from which
  // pt_fill_kernel places it without any ordering constraint.
  constexpr int RU = 8;  // work items in flight
  const int nmine = (w < NW && nrows > w) ? (nrows - w + NW - 1) / NW : 0;
  const int64_t elast = max(e1 - 1, e0);
  const uint64_t le = ((uint64_t)2 << lane) - 1;  // lanes <= mine
  int curs = 0;  // lane t: entries of list chunk t seen so far
  if (w < NW && e1 > e0) {
    // Work items = (row, 64-entry part of it), in order.  The fetch cursor runs ahead of the walk on its own (row
    // bounds come from LDS), so the walk itself is ONE loop without inner loops or loads behind branches -- hipcc's
    // wait-count pass puts s_waitcnt vmcnt(0) at every loop header / join it cannot see through, which made every
    // row wait for the row fetched last (1.5 us per row).
    int fk = 0, fpart = 0;  // fetch cursor: row index among this wave's rows, part of it
    int ik[RU], ioff[RU], iend[RU];  // item: row index, first entry (relative to e0), end of the row
    int cc[RU];
    auto fetch = [&](int u) __attribute__((always_inline)) {
      int rs = 0, rend = 0;
      if (fk < nmine) {
        const int rl = w + fk * NW;
        rs = s_rp[rl];
        rend = s_rp[rl + 1];
      }
      ik[u] = fk;
      ioff[u] = rs + fpart * 64;
      iend[u] = rend;
      cc[u] = col[min(e0 + ioff[u] + lane, elast)];
      // next item: the next part of a row longer than a wave, else the next row
      const bool more_parts = ioff[u] + 64 < rend;
      fpart = more_parts ? fpart + 1 : 0;
      fk = more_parts ? fk : fk + 1;
    };
#pragma unroll
    for (int u = 0; u < RU; ++u) fetch(u);
    bool busy = nmine > 0;
    bool too_long = false;
    while (busy) {
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int k = ik[u];
        const int off = ioff[u], rend = iend[u];
        const int c = cc[u];
        fetch(u);
        const bool act = (k < nmine) && (off + lane < rend);  // (the active lanes are 0 .. n - 1)
        const int g = act ? col_rank(c) : 0;
        const int ch = g >> CP_BITS;  // chunk of the column list: non-decreasing along the lanes
        const int chp = __shfl_up(ch, 1, 64);
        const bool head = act && (lane == 0 || ch != chp);
        const uint64_t hm = __ballot(head);
        const int nact = __popcll(__ballot(act));
        const int hl = 63 - __clzll((unsigned long long)(hm & le));                    // head lane of my run
        const uint64_t above = (lane == 63) ? 0 : (hm >> (lane + 1));
        const int run_end = above ? lane + __ffsll((unsigned long long)above) : nact;  // one past my run
        const int pos = __builtin_amdgcn_ds_bpermute(ch << 2, curs) + (lane - hl);     // position in the segment
        if (act) {
          too_long |= pos > 0xFFFF;
          codes[e0 + off + lane] = ((uint32_t)ch << 26) | ((uint32_t)(pos & 0xFFFF) << CP_BITS) | (uint32_t)(g & (CP - 1));
        }
        // every run head adds the length of its run to the counter of its tile (lane 63 is the dump of the other
        // lanes: tile indices are <= 62)
        curs += __builtin_amdgcn_ds_permute((head ? ch : 63) << 2, head ? run_end - lane : 0);
        if (u == RU - 1) busy = ik[0] < nmine;  // (uniform) items are in order: nothing left once slot 0 is past the end
      }
    }
    if (__any(too_long)) atomicMax(status, 4);
  }
  if (stop == 3) return;
  if (w < NW) s_cnt[w][lane] = (lane < T) ? curs : 0;
  __syncthreads();
  if (tid == 0) {
    // Processing order of the tiles.  A tile of the iterate takes the loaders about as long to stage whether ten
    // or ten thousand entries use it, so the order alternates heavy and light tiles: one of the heaviest left,
    // then two of the lightest left -- while the consumers work through a heavy tile the ring (NB = 4 buffers)
    // fills with the two light ones and the next heavy one.  j-th processed tile = chunk s_ord[j] of the list.
    int* tot = s_tmp[0];
    int* srt = s_tmp[1];
    for (int t = 0; t < T; ++t) {
      int c = 0;
      for (int ow = 0; ow < NW; ++ow) c += s_cnt[ow][t];
      tot[t] = c;
      int k = t;
      while (k > 0 && tot[srt[k - 1]] < c) {  // insertion sort, descending
        srt[k] = srt[k - 1];
        --k;
      }
      srt[k] = t;
    }
    int hi = 0, lo = T - 1, j = 0;
    while (hi <= lo) {
      s_ord[j++] = srt[hi++];
      for (int k = 0; k < 2 && hi <= lo; ++k) s_ord[j++] = srt[lo--];
    }
    for (int jj = T; jj < SEGW; ++jj) s_ord[jj] = 0;
  }
  __syncthreads();
  if (tid == 0) {  // segment offsets, (wave, processing order): a wave's stream is contiguous
    int run = 0;
    for (int ow = 0; ow < NW; ++ow) {
      int* tmp = s_tmp[2];
      for (int jj = 0; jj < SEGW; ++jj) tmp[jj] = (jj < T) ? s_cnt[ow][s_ord[jj]] : 0;
      for (int jj = 0; jj < SEGW; ++jj) {
        s_cnt[ow][jj] = run;
        run += tmp[jj];
      }
    }
    blk_ntile[b] = T;
    blk_ndist[b] = ndist;
  }
  __syncthreads();
  if (w < NW) {
    seg[((size_t)b * (NW + 1) + w) * SEGW + lane] = s_cnt[w][lane];
  } else if (w == NW) {
    seg[((size_t)b * (NW + 1) + NW) * SEGW + lane] = s_ord[lane];  // row NW: list chunk of every processed tile
  }
}

// Second step of the layout: one workgroup per (block, owner wave) places the wave's entries.  Every entry knows its
// segment and its position in it (the code written by pt_build_kernel), so this step has no ordering constraint: the
// entries are read in CSR order (coalesced), dropped at their final place in an LDS image of the wave's stream, and
// the image is copied out with full-line stores.  (Storing straight from the walk -- a few 8-byte pieces of ~27 x 64
// different cache lines per instruction -- cost 0.8 ms of the 1.5 ms build at 1M cells.)
// Final place within a segment of n = 64 q + r entries: TRANSPOSED over the 64 lanes of the consumer wave -- lane l
// owns a contiguous run of the segment's (row, column) order (q + 1 entries for l < r, else q) and chunk c holds each
// lane's c-th entry.  One instruction of the consumer then touches 64 entries that are n/64 apart in (row, column)
// order, i.e. different rows, and a lane meets the entries of a row in consecutive chunks and can sum the run in
// registers.
constexpr int FILL_CAP = 12800;  // entries of the LDS image (12 B each)
__global__ __launch_bounds__(THREADS) void pt_fill_kernel(const int64_t* __restrict__ rowptr, const double* __restrict__ val,
                                                          const uint32_t* __restrict__ codes, const int32_t* __restrict__ blk_row,
                                                          const int32_t* __restrict__ blk_ntile, const int32_t* __restrict__ seg,
                                                          double* __restrict__ pval, uint32_t* __restrict__ pidx) {
  __shared__ double s_val[FILL_CAP];
  __shared__ uint32_t s_idx[FILL_CAP];
  __shared__ int s_seg[SEGW + 1], s_inv[SEGW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.x / NW, w = blockIdx.x % NW;
  const int T = blk_ntile[b];
  if (T <= 0) return;
  const int row0 = blk_row[b];
  const int nrows = blk_row[b + 1] - row0;
  const int64_t e0 = rowptr[row0];
  if (tid < SEGW) {
    s_seg[tid] = seg[((size_t)b * (NW + 1) + w) * SEGW + tid];
    if (tid < T) s_inv[seg[((size_t)b * (NW + 1) + NW) * SEGW + tid]] = tid;  // list chunk -> processing position
  }
  __syncthreads();
  const int nmine = (nrows > w) ? (nrows - w + NW - 1) / NW : 0;  // rows of owner wave w
  // this workgroup wave's rows: k = wv, wv + 16, ...; their bounds are loaded once, one row per lane
  const int kmine = (nmine > wv) ? (nmine - wv + 15) / 16 : 0;  // <= 20 (RMAX / NW / 16)
  int64_t rs_l = 0, re_l = 0;
  if (lane < kmine) {
    const int rl = w + (wv + 16 * lane) * NW;
    rs_l = rowptr[row0 + rl];
    re_l = rowptr[row0 + rl + 1];
  }
  for (int ja = 0; ja < T;) {  // tile ranges [ja, jb) that fit the LDS image
    int jb = ja + 1;
    while (jb < T && s_seg[jb + 1] - s_seg[ja] <= FILL_CAP) ++jb;
    const int base = s_seg[ja], span = s_seg[jb] - s_seg[ja];
    if (span <= FILL_CAP) {
      auto place = [&](uint32_t code, double v, int k) __attribute__((always_inline)) {
        const int j = s_inv[code >> 26];
        if (j >= ja && j < jb) {
          const int pos = (code >> CP_BITS) & 0xFFFF;
          const int n = s_seg[j + 1] - s_seg[j];
          const int q = n >> 6, r = n & 63;
          int l, cch;
          if (pos < r * (q + 1)) {
            l = pos / (q + 1);
            cch = pos - l * (q + 1);
          } else {
            const int jj = pos - r * (q + 1);
            const int lq = jj / max(q, 1);
            l = r + lq;
            cch = jj - lq * q;
          }
          const int p = s_seg[j] - base + cch * 64 + l;
          s_val[p] = v;
          s_idx[p] = (code & (CP - 1)) | ((uint32_t)k << CP_BITS);
        }
      };
      constexpr int FG = 6;  // rows in flight per wave (their first 64 entries; what is longer follows in a plain loop)
      for (int i0 = 0; i0 < kmine; i0 += FG) {
        uint32_t code[FG];
        double v[FG];
        int64_t rs[FG], rend[FG];
#pragma unroll
        for (int g = 0; g < FG; ++g) {
          const int i = min(i0 + g, kmine - 1);
          rs[g] = __shfl(rs_l, i, 64);
          rend[g] = (i0 + g < kmine) ? __shfl(re_l, i, 64) : rs[g];
          const int64_t e = min(rs[g] + lane, max(rend[g] - 1, rs[g]));
          code[g] = codes[e];
          v[g] = val[e];
        }
#pragma unroll
        for (int g = 0; g < FG; ++g) {
          const int k = wv + 16 * (i0 + g);
          if (rs[g] + lane < rend[g]) place(code[g], v[g], k);
          for (int64_t e = rs[g] + 64 + lane; e < rend[g]; e += 64) place(codes[e], val[e], k);
        }
      }
      __syncthreads();
      for (int i = tid; i < span; i += THREADS) {
        pval[e0 + base + i] = s_val[i];
        pidx[e0 + base + i] = s_idx[i];
      }
      __syncthreads();
    } else {
      // one segment longer than the image (never on kNN graphs): its entries go out directly
      for (int i = 0; i < kmine; ++i) {
        const int k = wv + 16 * i;
        const int64_t rs = __shfl(rs_l, i, 64), rend = __shfl(re_l, i, 64);
        for (int64_t e = rs + lane; e < rend; e += 64) {
          const uint32_t code = codes[e];
          const int j = s_inv[code >> 26];
          if (j == ja) {
            const int pos = (code >> CP_BITS) & 0xFFFF;
            const int n = s_seg[j + 1] - s_seg[j];
            const int q = n >> 6, r = n & 63;
            int l, cch;
            if (pos < r * (q + 1)) {
              l = pos / (q + 1);
              cch = pos - l * (q + 1);
            } else {
              const int jj = pos - r * (q + 1);
              const int lq = jj / max(q, 1);
              l = r + lq;
              cch = jj - lq * q;
            }
            const int64_t p = e0 + s_seg[j] + cch * 64 + l;
            pval[p] = val[e];
            pidx[p] = (code & (CP - 1)) | ((uint32_t)k << CP_BITS);
          }
        }
      }
    }
    ja = jb;
  }
}

// pval32[e] = (float)pval[e] for the layout's nnz entries (nnz read from rowptr[n_rows] on the device)
__global__ __launch_bounds__(256) void pt_round_f32_kernel(const double* __restrict__ pval, float* __restrict__ pval32, int64_t,
                                                           const int64_t* __restrict__ rowptr, int64_t n_rows) {
  const int64_t nnz = rowptr[n_rows];
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < nnz; i += (int64_t)gridDim.x * blockDim.x * 2) {
    if (i + 1 < nnz) {
      const double2 v = *reinterpret_cast<const double2*>(pval + i);
      *reinterpret_cast<float2*>(pval32 + i) = make_float2((float)v.x, (float)v.y);
    } else {
      pval32[i] = (float)pval[i];
    }
  }
}

}  // namespace pt
}  // namespace meld

using namespace meld;

static int g_pt_ablate = 0;
// timing-only ablations for tools/spmm_compare.py (results are wrong while set): bits 0-2 pt_step_kernel
// (1 no accumulator updates, 2 no LDS gather, 4 no panel loads), bits 8-10 pt_build_kernel stops after stage n
extern "C" int meld_pt_debug_ablate(int mask) {
  g_pt_ablate = mask;
  return MELD_OK;
}

extern "C" int meld_pt_geometry(int* consumer_waves, int* rows_max, int* tile_cols, int* tiles_max) {
  if (consumer_waves) *consumer_waves = pt::NW;
  if (rows_max) *rows_max = pt::RMAX;
  if (tile_cols) *tile_cols = pt::CP;
  if (tiles_max) *tiles_max = pt::TMAX;
  return MELD_OK;
}

extern "C" int meld_pt_num_blocks(int64_t n_rows) {
  if (n_rows <= 0) return 0;
  // one block per CU and launch round: about 0.88 RMAX rows each leaves room for balancing the nonzeros
  const int64_t target = (int64_t)(0.88 * pt::RMAX);
  if (n_rows <= 64 * 256) return (int)std::max<int64_t>(1, ceil_div(n_rows, 256));
  const int64_t k = ceil_div(n_rows, 256 * target);
  int64_t nb = 256 * k;
  if (k == 1) nb = std::min<int64_t>(256, std::max<int64_t>(64, ceil_div(n_rows, 256)));
  while (nb * pt::RMAX < n_rows) nb += 256;
  return (int)nb;
}

extern "C" int64_t meld_pt_seg_len(int nb) { return (int64_t)nb * (pt::NW + 1) * pt::SEGW; }

extern "C" int meld_pt_build(const int64_t* rowptr, const int32_t* col, const double* val, int64_t n_rows, int64_t n_cols,
                             const meld_pt_layout_t* layout, uint32_t* codes, int32_t* status, meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && col && val && layout && layout->blk_row && layout->blk_ntile && layout->blk_ndist && layout->seg &&
                     layout->list_cols && layout->pval && layout->pidx && codes && status && n_rows > 0 && layout->nb > 0 && n_cols > 0,
                 "meld_pt_build: bad arguments");
  const int nb = layout->nb;
  MELD_CHECK_ARG((int64_t)nb * pt::RMAX >= n_rows, "meld_pt_build: %d blocks of %d rows cannot hold %lld rows", nb, pt::RMAX,
                 (long long)n_rows);
  hipStream_t st = S(stream);
  MELD_HIP_CALL(hipMemsetAsync(status, 0, sizeof(int32_t), st));
  if (ceil_div(n_cols, pt::BP) > pt::NPAN_MAX) {
    static const int32_t three = 3;
    MELD_HIP_CALL(hipMemcpyAsync(status, &three, sizeof(int32_t), hipMemcpyHostToDevice, st));
    return MELD_OK;
  }
  hipLaunchKernelGGL(pt::pt_plan_kernel, dim3(1), dim3(1024), 0, st, rowptr, n_rows, nb, const_cast<int32_t*>(layout->blk_row));
  hipLaunchKernelGGL(pt::pt_build_kernel, dim3(nb), dim3(pt::THREADS), 0, st, rowptr, col, val, n_cols, nb, layout->blk_row,
                     const_cast<int32_t*>(layout->blk_ntile), const_cast<int32_t*>(layout->blk_ndist),
                     const_cast<int32_t*>(layout->seg), const_cast<int32_t*>(layout->list_cols),
                     codes, status, (g_pt_ablate >> 8) & 7);
  if (((g_pt_ablate >> 8) & 7) == 0)
    hipLaunchKernelGGL(pt::pt_fill_kernel, dim3(nb * pt::NW), dim3(pt::THREADS), 0, st, rowptr, val, codes, layout->blk_row,
                       layout->blk_ntile, layout->seg, const_cast<double*>(layout->pval), const_cast<uint32_t*>(layout->pidx));
  if (layout->pval32 != nullptr)  // (its own streaming pass: a third scattered store in the walk costs 0.5 ms, this 0.08)
    hipLaunchKernelGGL(pt::pt_round_f32_kernel, dim3(2048), dim3(256), 0, st, layout->pval, const_cast<float*>(layout->pval32),
                       (int64_t)0, rowptr, n_rows);
  MELD_LAUNCH_CHECK("pt_build_kernel");
  return MELD_OK;
}

namespace {
template <int P, bool F32>
int pt_launch(const pt::StepArgs& a, hipStream_t st) {
  static bool configured = false;
  constexpr size_t lds = sizeof(double) * (size_t)(pt::RMAX + pt::NB * pt::CP) * P;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pt::pt_step_kernel<P, F32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_err("pt_step_kernel: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
      return MELD_ERR_HIP;
    }
    configured = true;
  }
  const unsigned grid = (unsigned)(ceil_div(a.nb, 8) * 8);
  hipLaunchKernelGGL((pt::pt_step_kernel<P, F32>), dim3(grid), dim3(pt::THREADS), lds, st, a);
  return MELD_OK;
}
}  // namespace

static int pt_step_cols(pt::StepArgs a, int p, hipStream_t st, bool f32) {
  int colofs = 0;
  a.ld = p;
  while (colofs + 2 <= p) {
    a.colofs = colofs;
    const int rc = pt_launch<2, false>(a, st);
    if (rc != MELD_OK) return rc;
    colofs += 2;
  }
  if (colofs < p) {
    a.colofs = colofs;
    const int rc = f32 ? pt_launch<1, true>(a, st) : pt_launch<1, false>(a, st);
    if (rc != MELD_OK) return rc;
  }
  return MELD_OK;
}

// internal entry (also used by the Lanczos drivers in spmm.hip)
int meld::pt_step(const meld_pt_layout_t* L, const int64_t* rowptr, const double* dw, int p, const double* x_full,
                  int64_t x_row_offset, const double* z, double* y, double* r, double alpha, double beta, double gamma,
                  double coef, double* dots, const double* coef_dev, hipStream_t st) {
  if (L->nb == 0) return MELD_OK;
  pt::StepArgs a;
  a.blk_row = L->blk_row; a.blk_ntile = L->blk_ntile; a.blk_ndist = L->blk_ndist; a.seg = L->seg;
  a.list_cols = L->list_cols; a.pval = L->pval; a.pidx = L->pidx; a.rowptr = rowptr; a.dw = dw; a.x_full = x_full;
  a.z = z; a.y = y; a.r = r; a.dots = dots; a.coef_dev = coef_dev; a.x_row_offset = x_row_offset; a.alpha = alpha;
  a.beta = beta; a.gamma = gamma; a.coef = coef; a.nb = L->nb; a.ld = p; a.colofs = 0;
  a.ablate = g_pt_ablate;
  a.pval32 = L->pval32;
  // the fp32 copy of the values serves the lmax estimate only (p = 1 with device-resident Lanczos scalars)
  return pt_step_cols(a, p, st, coef_dev != nullptr && p == 1 && L->pval32 != nullptr);
}

extern "C" int meld_pt_cheby_step(const meld_pt_layout_t* layout, const int64_t* rowptr, const double* dw, int64_t n_rows,
                                  int p, const double* x_full, int64_t x_row_offset, const double* z, double* y, double* r,
                                  double alpha, double beta, double gamma, double coef, double* dots, meld_stream_t stream) {
  MELD_CHECK_ARG(layout && layout->blk_row && layout->blk_ntile && layout->blk_ndist && layout->seg && layout->list_cols &&
                     layout->pval && layout->pidx && rowptr && dw && x_full && y && n_rows >= 0 && p >= 1,
                 "meld_pt_cheby_step: bad arguments");
  MELD_CHECK_ARG(gamma == 0.0 || z != nullptr, "meld_pt_cheby_step: z is required when gamma != 0");
  MELD_CHECK_ARG(dots == nullptr || p == 1, "meld_pt_cheby_step: dots are only produced for p == 1");
  hipStream_t st = S(stream);
  if (dots) MELD_HIP_CALL(hipMemsetAsync(dots, 0, sizeof(double) * 2 * pt::DOT_SLOTS, st));
  if (n_rows == 0) return MELD_OK;
  const int rc = pt_step(layout, rowptr, dw, p, x_full, x_row_offset, z, y, r, alpha, beta, gamma, coef, dots, nullptr, st);
  if (rc != MELD_OK) return rc;
  MELD_LAUNCH_CHECK("pt_step_kernel");
  return MELD_OK;
}
