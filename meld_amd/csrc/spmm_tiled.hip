// spmm_tiled.hip -- the graph-Laplacian recurrence step on a panel-tiled, symmetry-folded copy of W.
//
// Same operator as spmm.hip (one fused step of [UPSTREAM pygsp cheby_op], called from reference
// meld/filter.py:59, and the SpMV of the lmax estimate, meld/filter.py:39):
//     y = alpha * (dw .* x - W x) + beta * x + gamma * z ;   r += coef * y
// W is static over the ~65 steps of a fit_transform (35 Lanczos + 29 Chebyshev), so it is re-laid out once
// (round-3 layout):
//
//   * rows are cut into nb blocks of <= RMAX rows with balanced estimated TIME (entries + far-reaching entries + rows,
//     pt_row_weight_kernel; balanced nonzero counts when there is no scratch for the weights), ONE block per CU; the
//     block's accumulators (one per row) live in LDS for the whole step;
//   * IN part -- W is symmetric, and 40 % of a block's nonzeros (1M-cell benchmark graph, locality order) have
//     their column inside the block's own row range.  Those are stored ONCE per pair (i < j): the block's own
//     slice of the iterate is staged in LDS, a pair contributes v x_j to row i and v x_i to row j.  12 bytes
//     per PAIR instead of 24: the matrix stream shrinks by 20 %;
//   * OUT part -- the distinct columns OUTSIDE the block's row range are listed (sorted) and cut into tiles of
//     CP columns; a tile's slice of the iterate is staged into a ring of NB LDS buffers by four loader waves
//     while twelve consumer waves work; the nonzeros are stored tile by tile;
//   * every consumer wave streams ITS OWN contiguous slice of the block in whole chunks of 64 entries
//     (fp64 value + one 32-bit word: LDS byte offsets of the column's x and of the row's accumulator), eight
//     chunks in flight, no per-chunk bookkeeping: ~20 instructions per chunk (the round-2 kernel issued ~60
//     and was as much issue-bound as memory-bound);
//   * accumulation is ds_add_f64 into the LDS accumulators.  The j side of a pair is added by whichever wave
//     owns row i, so several waves add into one cell and the ORDER of the additions is not fixed: results
//     are reproducible to rounding (~1e-16 relative per step), not bit for bit.
//
// The symmetric fold is only valid for a bitwise symmetric W.  Graphs built by this library are (a + b and
// d_i d_j commute); the builder nevertheless verifies it for every block (a commutative hash of the upper
// and the lower in-block entries must cancel) and reports status 5 otherwise -- the caller rebuilds with the
// fold switched off (every in-block column then goes through the tiles like any other).
#include "common.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace meld {
namespace pt {

constexpr int NW = 12;                    // consumer waves = row owners
constexpr int NL = 4;                     // loader waves
constexpr int THREADS = 64 * (NW + NL);   // 1024: one workgroup per CU
constexpr int RMAX = 4080;                // rows per block at most (12 x 340; a row slot fits 12 bits)
constexpr int SLOTS = RMAX / NW;          // rows per consumer wave
constexpr int RPAD = 4096;                // accumulator cells in LDS
constexpr int CP = 1024;                  // columns per staged tile
constexpr int CP_BITS = 10;
constexpr int NB = 4;                     // ring of staged tiles in LDS (NB * CP columns of the iterate)
constexpr int TMAX = 61;                  // tiles per block at most (lanes 62, 63 of a segment row are header words)
constexpr int SEGW = 64;                  // words per segment row (one per lane of a wave)
constexpr int SEGROWS = 2 * (NW + 1);     // per block: NW + 1 rows of chunk bounds / header, NW + 1 rows of entry offsets
constexpr int QW = 8;                     // chunks per window of the IN part (a lane owns QW consecutive pairs)
constexpr int PADCAP = 64;                // padding entries a wave's stream may hold (the tail of its IN part)
constexpr int KMAX = 512;                 // chunk descriptors per wave (+ KSLACK zero descriptors the prefetch may read)
constexpr int KSLACK = 32;
constexpr int D_FIRST = 0x80, D_LAST = 0x100;  // descriptor = entries of the chunk (0..64) | first / last chunk of its tile
constexpr int BP = 2048;                  // columns per bitmap panel of the builder (independent of the tile size)
constexpr int BP_BITS = 11;
constexpr int NPAN_MAX = 4096;            // bitmap panels the builder can index (n_cols <= 8.4 M)
constexpr int TP_MAX = 384;               // distinct bitmap panels one block may touch (bitmap rows in LDS)
constexpr int DOT_SLOTS = 64;             // == spmm.hip
constexpr int STRIDE16 = 16;              // LDS bytes per row / column slot at p = 2 (8 at p = 1); index words hold slot * 16
constexpr unsigned XBASE = RPAD * STRIDE16; // LDS byte offset of the staged iterate (own slice, then the ring)
constexpr unsigned CTRL = XBASE + NB * CP * STRIDE16;  // LDS byte offset of the shared control words (512 B)
constexpr size_t LDS_BYTES = CTRL + 512;
constexpr int T_IN = 63, T_SKIP = 62;     // code markers of the builder: IN pair, dropped lower in-block entry

static_assert((NB & (NB - 1)) == 0, "NB");
static_assert((1 << CP_BITS) == CP && (1 << BP_BITS) == BP && BP / 32 == 64, "CP / BP");
static_assert(NB * CP * STRIDE16 == 65536 && RPAD * STRIDE16 == 65536, "LDS map: offsets are 16-bit fields of the index word");
static_assert(RMAX % NW == 0 && RMAX <= RPAD, "RMAX");

// lgkmcnt(0) only: LDS operations of this wave have completed; global loads stay in flight
#define PT_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xC07F)

struct StepArgs {
  const int32_t* blk_row;    // [nb + 1]
  const int32_t* blk_ntile;  // [nb]
  const int32_t* blk_ndist;  // [nb]
  const int32_t* seg;        // [nb][SEGROWS][SEGW]
  const int32_t* list_cols;  // block b's sorted distinct OUT columns start at rowptr[blk_row[b]]
  const double* pval;        // streams: block b starts at rowptr[blk_row[b]] + b * NW * PADCAP
  const float* pval32;       // the same rounded to fp32 (F32 instantiation: the lmax estimate's SpMV)
  const uint32_t* pidx;      // index words, see pt_fill_kernel
  const uint16_t* cdesc;     // [nb][NW][KMAX + KSLACK] chunk descriptors of every wave's stream
  const int64_t* rowptr;     // CSR row pointers (entry base of a block)
  const double* dw;
  const double* x_full;
  const double* z;
  double* y;
  double* r;
  double* dots;
  const double* coef_dev;
  PtLanczos lz;  // (lz.nrm2_prev != NULL: the scalars are derived in the kernel, see common.hpp)
  int64_t x_row_offset;
  double alpha, beta, gamma, coef;
  double coef_x;  // r += coef * y + coef_x * x (own rows): lets a recurrence touch r every other step (meld_pt_cheby_run)
  int nb;
  int ld, colofs;
  unsigned long long* stamps;  // development: [nb][16][8] wall-clock stamps of every wave (meld_pt_debug_stamps), or NULL
  int ablate;  // timing-only modes (results wrong): 4 no panel loads, 8 panel gathers from a 16 KB window of x
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
// LDS slot (16 bytes) at byte offset `off`
template <int P>
__device__ __forceinline__ V<P> lds_get(const char* lds, unsigned off) {
  V<P> o;
  if constexpr (P == 1) {
    o.v[0] = *reinterpret_cast<const double*>(lds + off);
  } else {
    const double2 t = *reinterpret_cast<const double2*>(lds + off);
    o.v[0] = t.x;
    o.v[1] = t.y;
  }
  return o;
}
template <int P>
__device__ __forceinline__ void lds_put(char* lds, unsigned off, const V<P>& a) {
  if constexpr (P == 1) {
    *reinterpret_cast<double*>(lds + off) = a.v[0];
  } else {
    *reinterpret_cast<double2*>(lds + off) = make_double2(a.v[0], a.v[1]);
  }
}
// LDS fp64 add without a return value (ds_add_f64)
__device__ __forceinline__ void lds_add(char* lds, unsigned off, double v) {
  __hip_atomic_fetch_add(reinterpret_cast<double*>(lds + off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

constexpr int U = 8;  // entry chunks (64 nonzeros each) in flight per consumer wave
static_assert(U == 8, "the consumer stream names its 8 slots (v96..v119) and waits with vmcnt(2 (U - 1))");

// Wave-uniform read-only words (chunk descriptors, schedule headers) are read through the constant address space: the
// compiler then issues SCALAR loads for them whatever stores the kernel makes elsewhere (a vector load in the consumer
// loop would break the hand-counted vmcnt of the entry stream).  The layout is never written by the step kernels.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4* const_u4_ptr;
typedef const __attribute__((address_space(4))) int32_t* const_i32_ptr;
__device__ __forceinline__ uint4 load_u4_const(const void* p) {
  const u32x4 t = *(const_u4_ptr)(uintptr_t)p;
  return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ int4 load_i4_const(const void* p) {
  const u32x4 t = *(const_u4_ptr)(uintptr_t)p;
  return make_int4((int)t.x, (int)t.y, (int)t.z, (int)t.w);
}
__device__ __forceinline__ int32_t load_i32_const(const void* p) { return *(const_i32_ptr)(uintptr_t)p; }

// LDS words shared by the waves of a workgroup, polled / bumped with plain LDS operations (the CU's LDS is
// coherent for its own waves; a wave's LDS operations complete in order)
__device__ __forceinline__ int lds_peek(const int* p) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void lds_bump(int* p) { __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// LDS slot of row rl of a block (rows are dealt to the consumer waves round-robin: row rl belongs to wave rl % NW):
// wave-major, so that the rows of ONE wave -- the rows the lanes of one instruction touch -- are consecutive slots
// and spread over all LDS banks (in the natural order they are 12 slots = 192 bytes apart: 4 bank positions, 16-way
// conflicts on every accumulator update)
__device__ __forceinline__ int row_slot(int rl) { return (rl % NW) * SLOTS + rl / NW; }

// block b's streams (pval / pidx) start here
__device__ __forceinline__ int64_t stream_base(const int64_t* __restrict__ rowptr, int row0, int b) {
  return rowptr[row0] + (int64_t)b * (NW * PADCAP);
}
// the 64-bit symmetry sum of block b lives in the first two words of the block's last (otherwise unused) segment row
__device__ __forceinline__ unsigned long long* symsum_of(int32_t* seg, int b) {
  return reinterpret_cast<unsigned long long*>(seg + ((size_t)b * SEGROWS + (SEGROWS - 1)) * SEGW);
}

// F32: the matrix VALUES are streamed from the fp32 copy (8 instead of 12 bytes per entry); vectors, products and
// sums stay fp64.  Only the Lanczos SpMV of the lmax estimate uses it: rounding W to fp32 moves the largest
// eigenvalue by < 1e-7 relative, far inside the tolerance that estimate is computed to (and the 1.01 factor on it).
template <int P, bool F32>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_num_vgpr(48))) void pt_step_kernel(StepArgs a) {
  // LDS map (bytes): [0, 64 Ki) accumulators, row rl of the block at row_slot(rl) * 16;  [64 Ki, 128 Ki) the iterate: first
  // the block's own rows (IN part), then the ring of NB staged tiles (OUT part), column slot c at 64 Ki + c * 16
  // -- no static LDS: the dynamic block starts at LDS address 0, so the fields of an index word ARE addresses --
  // and behind them the few words the waves share:
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int STRIDE = 8 * P;  // bytes per slot (p = 1: half of the index words' slot * 16)
  constexpr int SH = (P == 1) ? 1 : 0;
  double (*s_dot)[NW + NL] = reinterpret_cast<double (*)[NW + NL]>(lds + CTRL);  // [2][NW + NL]
  // ring state: s_prod[i] = tiles staged into buffer i so far (monotonic: buffer i holds tile t, t % NB == i,
  // once s_prod[i] == t / NB + 1; a tile is staged by ONE loader wave); s_cons[i] = consumer waves that have left a
  // tile of buffer i (tile t may be overwritten once s_cons[i] == NW * (t / NB + 1)); s_in_done = consumer waves
  // that have finished the IN part (the ring overlays the block's own slice of the iterate)
  int* s_prod = reinterpret_cast<int*>(lds + CTRL + 256);  // [NB]
  int* s_cons = s_prod + NB;                               // [NB]
  int* s_in_done_p = s_cons + NB;
#define s_in_done (*s_in_done_p)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-contiguous row blocks (workgroup b runs on XCD b % 8; placement used for L2 locality only)
  const int nbp = gridDim.x;  // multiple of 8
  const int per = nbp >> 3;
  const int b = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (b >= a.nb) return;  // padding workgroup (whole workgroup leaves: no barrier is left hanging)
  if (a.lz.stop != nullptr && *a.lz.stop != 0) return;  // (uniform) the host has what it needs: the rest of the batch is void

  double alpha = a.alpha, gamma = a.gamma;
  const double beta = a.beta, coef = a.coef, coef_x = a.coef_x;
  if (a.coef_dev != nullptr) {  // device-resident Lanczos: scalars written by the previous iteration
    alpha = a.coef_dev[3];
    gamma = a.coef_dev[4];
  }
  const int ab = a.ablate;
  auto stamp = [&](int i) __attribute__((always_inline)) {
    if (a.stamps != nullptr && lane == 0) a.stamps[((size_t)b * 16 + w) * 8 + i] = wall_clock64();
  };
  stamp(0);
  // the block's header words (one scalar load instead of a chain of dependent ones: first row, rows, tiles, distinct
  // OUT columns, CSR offset of the first row) sit behind the symmetry sum in the block's last segment row
  const int32_t* segb = a.seg + (size_t)b * SEGROWS * SEGW;
  const int4 hw0 = load_i4_const(segb + (SEGROWS - 1) * SEGW);
  const int4 hw1 = load_i4_const(segb + (SEGROWS - 1) * SEGW + 4);
  const int row0 = __builtin_amdgcn_readfirstlane(hw0.z);
  const int nrows = __builtin_amdgcn_readfirstlane(hw0.w);
  if (nrows == 0) return;  // (uniform) an empty block has nothing to stage, accumulate or write
  const int T = __builtin_amdgcn_readfirstlane(hw1.x);
  const int ndist = __builtin_amdgcn_readfirstlane(hw1.y);
  const int64_t e0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(hw1.w) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane(hw1.z));
  const int64_t ebase = e0 + (int64_t)b * (NW * PADCAP);

  // ---- consumer waves: header + the first U chunks requested before anything else -------------------------------
  // The entry stream keeps U chunks (64 entries each) in flight per wave with wait counts placed by hand: hipcc's
  // wait-count pass forgets the order of loads pending over a loop's back edge and drains the queue every round.
  // The loads are issued from inline asm into PHYSICAL registers the compiler does not own (the kernel is limited to
  // v0..v95 by amdgpu_num_vgpr(48) -- on gfx90a and later the attribute counts the unified file, i.e. its value is
  // doubled -- and the asm names v96..v119: values v[96 + 2u : 97 + 2u], index words v[112 + u]) -- a
  // register with a load in flight must never be copied and the allocator copies asm operands freely.  Loads return
  // in order and every slot is re-issued right after it has been read, so when chunk k is wanted exactly 2 (U - 1)
  // younger loads are in flight: s_waitcnt vmcnt(14).  Past the end of a wave's stream the loads read the next
  // wave's entries or the slack behind the arrays; nothing is done with them.
#ifndef PT_NT
#define PT_NT " nt"
#endif
#define PT_SLOT_LOAD(VLO, VHI, IX)                                                                          \
  asm volatile("global_load_dwordx2 v[" #VLO ":" #VHI "], %0, %2" PT_NT "\n\tglobal_load_dword v" #IX ", %1, %3" PT_NT \
               :                                                                                            \
               : "v"(off8), "v"(off4), "s"(pvk), "s"(pik)                                                   \
               : "memory", "v" #VLO, "v" #VHI, "v" #IX)
#define PT_SLOT_LOAD32(VLO, VHI, IX)                                                                   \
  asm volatile("global_load_dword v" #VLO ", %0, %1" PT_NT "\n\tglobal_load_dword v" #IX ", %0, %2" PT_NT        \
               :                                                                                         \
               : "v"(off4), "s"(pvk32), "s"(pik)                                                         \
               : "memory", "v" #VLO, "v" #VHI, "v" #IX)
#define PT_SLOT_TAKE(VLO, VHI, IX)                                                                                     \
  asm volatile("s_waitcnt vmcnt(14)\n\tv_mov_b32 %0, v" #VLO "\n\tv_mov_b32 %1, v" #VHI "\n\tv_mov_b32 %2, v" #IX \
               : "=v"(t_lo), "=v"(t_hi), "=v"(t_ix)                                                                    \
               :                                                                                                       \
               : "memory")
  const unsigned off8 = (unsigned)lane * 8u, off4 = (unsigned)lane * 4u;
  // issue the loads of slot u for the chunk that starts at pvk / pik (lanes beyond the chunk's entries read the
  // entries that follow; they are masked off when the chunk is processed)
  auto issue = [&](int u, const double* pvk, const float* pvk32, const uint32_t* pik) __attribute__((always_inline)) {
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
  // chunk descriptors of one round (U chunks, 16 bits each) as four dwords; the address is wave-uniform (scalar loads)
  auto desc_of = [](const uint4& d, int u) __attribute__((always_inline)) {
    const unsigned wd = (u < 2) ? d.x : (u < 4) ? d.y : (u < 6) ? d.z : d.w;
    return (int)((wd >> ((u & 1) * 16)) & 0xFFFFu);
  };

  const double* xs = a.x_full;
  const double* zs = a.z;
  double* ys = a.y;
  int n_in = 0, K = 0;
  const double* pv = nullptr;
  const float* pv32 = nullptr;
  const uint32_t* pi = nullptr;
  const uint16_t* cd = nullptr;
  uint4 dA = make_uint4(0, 0, 0, 0), dB = dA;  // descriptors of the round being processed / being requested
  int eo = 0;                                   // entry offset of the next chunk to request
  if (w < NW) {
    // header words of the wave's schedule: [62] IN chunks | all chunks << 16, [63] the wave's offset in the block's stream
    const int hdr = load_i32_const(segb + w * SEGW + 62);
    const int soff = load_i32_const(segb + w * SEGW + 63);
    n_in = __builtin_amdgcn_readfirstlane(hdr & 0xFFFF);
    K = __builtin_amdgcn_readfirstlane((int)((unsigned)hdr >> 16));
    pv = a.pval + ebase + soff;
    pv32 = a.pval32 + ebase + soff;
    pi = a.pidx + ebase + soff;
    cd = a.cdesc + ((size_t)b * NW + w) * (KMAX + KSLACK);
    dA = load_u4_const(cd);
    dB = load_u4_const(cd + U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      issue(u, pv + eo, pv32 + eo, pi + eo);
      eo += desc_of(dA, u) & 0x7F;
    }
  }

  // ---- all waves: accumulators to zero, the block's own rows of the iterate into LDS -----------------------------
  for (int i = tid; i < RPAD * STRIDE / 16; i += THREADS) *reinterpret_cast<double2*>(lds + i * 16) = make_double2(0.0, 0.0);
  for (int i = tid; i < nrows; i += THREADS)
    lds_put<P>(lds, XBASE + row_slot(i) * STRIDE, ldg<P>(xs, a.x_row_offset + row0 + i, a.ld, a.colofs));
  if (tid < NB) {
    s_prod[tid] = 0;
    s_cons[tid] = 0;
  }
  if (tid == 0) s_in_done = 0;
  stamp(1);
  __syncthreads();
  stamp(2);

  // No workgroup barrier from here to the end of the block's tiles: a consumer wave waits only for the loaders
  // (tile t staged), the loaders only for the slowest consumer NB tiles back (and once for the end of the IN part).
  if (w < NW) {
    // ------------------------------------------------------------------ consumer wave
    int cur = -1;                // the tile this wave holds (tiles are entered in order, every one exactly once)
    bool in_open = true;         // the IN part has not been signed off yet
    double s0 = 0.0, s1 = 0.0;   // running sums of the lane's current row run
    auto finish_in = [&]() __attribute__((always_inline)) {
      PT_WAIT_LDS();  // my reads of the block's own slice have completed
      if (lane == 0) lds_bump(&s_in_done);
      in_open = false;
      stamp(3);
    };
    // Two-stage pipeline over the chunks: the x values of chunk k are requested from LDS at the end of step k and
    // consumed at step k + 1 (after the entry wait of chunk k + 1), so the LDS round trip is covered by a step.
    bool h_any = false, h_pair = false, h_last = false;  // (uniform) the chunk waiting in stage 2
    bool h_act = false;
    double h_v = 0.0;
    unsigned h_clo = 0, h_chi = 0, h_flush = 0;
    bool new_row = true;  // the lane's next pair starts a row run (its x_i has to be fetched)
    V<P> h_x, h_xi;
#pragma unroll
    for (int c = 0; c < P; ++c) h_x.v[c] = h_xi.v[c] = 0.0;
    // A lane's consecutive chunks hold consecutive entries of the segment in (row, column) order (segments are stored
    // transposed over the wave), i.e. runs of the same row: the run is summed in registers and goes to the LDS
    // accumulator where the builder has flagged the lane's last entry of the row -- one ds_add per run instead of one
    // per entry, and never two lanes of one instruction on the same cell.
    auto stage2 = [&]() __attribute__((always_inline)) {
      if (h_any) {
        if (h_act) {
          s0 = fma(h_v, h_x.v[0], s0);
          if constexpr (P == 2) s1 = fma(h_v, h_x.v[1], s1);
          if (h_pair) {  // (uniform) a pair (i < j) of the block's own square also adds v x_i to row j
            lds_add(lds, h_clo, h_v * h_xi.v[0]);
            if constexpr (P == 2) lds_add(lds, h_clo + 8, h_v * h_xi.v[1]);
          }
          if (h_flush) {
            lds_add(lds, h_chi, s0);
            if constexpr (P == 2) lds_add(lds, h_chi + 8, s1);
            s0 = 0.0;
            s1 = 0.0;
          }
        }
        if (h_last && lane == 0) lds_bump(&s_cons[cur % NB]);  // leave the tile (its x values have been consumed)
      }
    };
    for (int k0 = 0; k0 < K; k0 += U) {
      const uint4 dC = load_u4_const(cd + k0 + 2 * U);  // (KSLACK zero descriptors follow the last one)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u;
        double v;
        uint32_t ix;
        take(u, v, ix);
        issue(u, pv + eo, pv32 + eo, pi + eo);  // the slot is free again: chunk k + U
        eo += desc_of(dB, u) & 0x7F;
        stage2();
        h_any = k < K;  // (uniform; the last round may be partial)
        if (h_any) {
          const int d = desc_of(dA, u);
          const int n = d & 0x7F;
          h_pair = k < n_in;
          h_last = (d & D_LAST) != 0;
          if (!h_pair) {
            if (in_open) finish_in();
            if (d & D_FIRST) {  // enter the next tile: wait until it is staged
              ++cur;
              const int need = cur / NB + 1;
              while (lds_peek(&s_prod[cur % NB]) < need) __builtin_amdgcn_s_sleep(1);
            }
          }
          h_act = lane < n;
          h_v = v;
          h_clo = (ix & 0xFFF0u) >> SH;  // column: row j of a pair / slot in the ring  (LDS byte offsets)
          h_chi = ix >> (16 + SH);       // row (i of a pair)
          h_flush = ix & 1u;
          h_x = lds_get<P>(lds, XBASE + (h_act ? h_clo : 0u));
          if (h_pair && new_row) h_xi = lds_get<P>(lds, XBASE + h_chi);
          new_row = h_flush != 0;
        }
      }
      dA = dB;
      dB = dC;
    }
    stage2();
    stamp(4);
    // the slots still have loads in flight (re-issues past the end of the stream)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (in_open) finish_in();
  } else {
    // ------------------------------------------------------------------ loader wave
    // Loader wave lw stages the tiles t = lw, lw + NL, ... on its own (64 lanes x PER columns), so NL tiles are in
    // flight besides what the ring holds; the gathered values of its next tile and the column list of the one after
    // are requested before it waits for the ring buffer of the current one.  Loads are unconditional from clamped
    // positions: slots beyond the block's last distinct column are never referenced by an entry.
    const int lw = w - NW;
    const int32_t* lst = a.list_cols + e0;
    const int srcv = segb[NW * SEGW + lane];  // lane j: list chunk of the j-th processed tile
    constexpr int PER = CP / 64;              // columns per lane and tile
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
      for (int k = 0; k < PER; ++k) xv[k] = ldg<P>(xs, (ab & 8) ? (col[k] & 1023) : col[k], a.ld, a.colofs);
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
      } else {        // the ring overlays the block's own slice: every consumer has finished the IN part
        while (lds_peek(&s_in_done) < NW) __builtin_amdgcn_s_sleep(1);
      }
      if (!(ab & 4)) {
#pragma unroll
        for (int k = 0; k < PER; ++k) lds_put<P>(lds, XBASE + (unsigned)(buf * CP + k * 64 + lane) * STRIDE, xv[k]);
        if (t + NL < T) gather();                   // my next tile (its list landed one tile ago)
        if (t + 2 * NL < T) load_list(t + 2 * NL);  // and the list of the one after
      }
      PT_WAIT_LDS();  // the panel has landed
      if (lane == 0) lds_bump(&s_prod[buf]);
    }
  }
#undef PT_SLOT_LOAD
#undef PT_SLOT_LOAD32
#undef PT_SLOT_TAKE
#undef s_in_done

  // ---------------------------------------------------------------------- epilogue (all 16 waves)
  // operands of the block's rows are requested before the closing barrier, consumed after it
  constexpr int NR = (RMAX + THREADS - 1) / THREADS;
  V<P> xl[NR], zl[NR], rl_[NR];
  double dwi[NR];
  const bool lz_on = a.lz.nrm2_prev != nullptr;  // (uniform)
  const double lz_part = lz_on ? a.lz.nrm2_prev[lane & (DOT_SLOTS - 1)] : 0.0;  // requested here, summed behind the barrier
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int rl = min(tid + q * THREADS, max(nrows - 1, 0));
    const int64_t row = row0 + rl;
    xl[q] = ldg<P>(xs, a.x_row_offset + row, a.ld, a.colofs);
    dwi[q] = a.dw[row];
#pragma unroll
    for (int c = 0; c < P; ++c) zl[q].v[c] = rl_[q].v[c] = 0.0;
    if (gamma != 0.0 || lz_on) zl[q] = ldg<P>(zs, row, a.ld, a.colofs);
    if (a.r != nullptr) rl_[q] = ldg<P>(a.r, row, a.ld, a.colofs);
  }
  stamp(5);
  PT_WAIT_LDS();
  __builtin_amdgcn_s_barrier();  // every accumulator is final
  stamp(6);
  if (lz_on) {  // every wave derives the iteration's scalars from the same 64 partial sums (same order, same value)
    const double beta_prev = sqrt(wave_sum(lz_part));
    const double s_prev = a.lz.state_prev[0];
    alpha = 1.0 / beta_prev;
    gamma = -beta_prev * s_prev;
    if (b == 0 && tid == 0) {
      a.lz.state_cur[0] = alpha;
      if (a.lz.it > 0) a.lz.betas[a.lz.it - 1] = beta_prev;
    }
    if (b == 0 && tid >= 64 && tid < 64 + DOT_SLOTS) a.lz.nrm2_zero[tid - 64] = 0.0;
  }
  double d_yx = 0.0, d_yy = 0.0;
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int rl = tid + q * THREADS;
    if (rl < nrows) {
      const int64_t row = row0 + rl;
      const V<P> ap = lds_get<P>(lds, (unsigned)row_slot(rl) * STRIDE);
      V<P> yv;
#pragma unroll
      for (int c = 0; c < P; ++c) {
        const double lx = dwi[q] * xl[q].v[c] - ap.v[c];  // (L x)_i
        yv.v[c] = alpha * lx + beta * xl[q].v[c] + gamma * zl[q].v[c];
        rl_[q].v[c] += coef * yv.v[c] + coef_x * xl[q].v[c];
        d_yx += yv.v[c] * xl[q].v[c];
        d_yy += yv.v[c] * yv.v[c];
      }
      stg<P>(ys, row, a.ld, a.colofs, yv);
      if (a.r != nullptr) stg<P>(a.r, row, a.ld, a.colofs, rl_[q]);
    }
  }
  stamp(7);
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

// Time-balanced plan.  A block's time is not its entry count: fitted on the 1M benchmark graph (per-wave stamps,
// profiles/r03_recurrence_step_timeline.txt) it is 0.361 us per 1000 entries + 1.077 us per 1000 DISTINCT out-of-block
// columns + 2.84 us per 1000 rows, and with equal entries the block of the sparsest region (rows at the cap, 38 k distinct
// columns against a mean of 24 k) ends 20 us after the median one.  The distinct count of a block is not known before the
// blocks are, but the entries that reach further than half a block from their own row are a per-row quantity that tracks
// it (tools/sim_plan.py: the slowest block of the model falls from 109 to 103 us with weights a deg + 0.7 b far + c).
// pt_row_weight_kernel writes those weights (integers, 1e-6 us), an exclusive scan turns them into the prefix the plan
// bisects instead of rowptr.
constexpr int W_ENTRY = 361, W_FAR = 754, W_ROW = 2840;
__global__ __launch_bounds__(256) void pt_row_weight_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            int64_t n_rows, int64_t col_base, int win, int32_t* __restrict__ w) {
  // 16 lanes per row: a row's entries (16-30 on a kNN graph) are one or two coalesced reads
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  int far = 0;
  int64_t e0 = 0, e1 = 0;
  if (i < n_rows) {
    e0 = rowptr[i];
    e1 = rowptr[i + 1];
    const int64_t self = col_base + i;
    for (int64_t e = e0 + sub; e < e1; e += 16) {
      const int64_t dlt = (int64_t)col[e] - self;
      far += (dlt > win || dlt < -win) ? 1 : 0;
    }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) far += __shfl_xor(far, off, 64);
  if (i < n_rows && sub == 0) {
    const int64_t v = (int64_t)W_ENTRY * (e1 - e0) + (int64_t)W_FAR * far + W_ROW;
    w[i] = (int32_t)min(v, (int64_t)0x3fffffff);
  }
}
__global__ __launch_bounds__(1024) void pt_plan_weighted_kernel(const int64_t* __restrict__ wpre, int64_t n_rows, int nb,
                                                                int32_t* __restrict__ blk_row) {
  const int64_t tot = wpre[n_rows];
  for (int b = threadIdx.x; b < nb; b += blockDim.x) {
    const int64_t target = (tot / nb) * (b + 1) + ((tot % nb) * (b + 1)) / nb;
    int64_t lo = 0, hi = n_rows;
    while (lo < hi) {  // first row index whose prefix reaches the target
      const int64_t mid = (lo + hi) >> 1;
      if (wpre[mid] < target) lo = mid + 1; else hi = mid;
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

__device__ __forceinline__ uint64_t lanes_below(int l) { return ((uint64_t)1 << l) - 1; }  // l in 0..63

// One workgroup per row block: list of the distinct OUT columns, and for every nonzero a 32-bit code that says where
// it goes (pt_fill_kernel places it):
//   OUT entry            list chunk << 26 | last entry of its row in this tile << 25 | position within its (wave, tile)
//                        segment in (row, column) order << 10 | tile-local column
//   IN pair (i < j)      T_IN << 26 | position among the owner wave's pairs in (row, column) order << 13 |
//                        last pair of its row (part) << 12 | j - row0
//   lower in-block entry T_SKIP << 26 | j - row0   (dropped: its twin (j, i) is stored by the wave that owns row j)
// plus the per-wave chunk schedule (seg).  `sym` = 0: no IN part, every column goes through the tiles.
// status[0] = max over blocks of an error code (1: unused since the panel groups, 2: more than TMAX
// tiles, 3: n_cols too large, 4: a segment / a wave's pairs / its padding beyond the code's range, 6: a diagonal entry).
__global__ __launch_bounds__(THREADS) void pt_build_kernel(
    const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t n_cols, int64_t col_base, int sym,
    int nb, const int32_t* __restrict__ blk_row, int32_t* __restrict__ blk_ntile, int32_t* __restrict__ blk_ndist,
    int32_t* __restrict__ seg, uint16_t* __restrict__ cdesc, int32_t* __restrict__ list_cols, uint32_t* __restrict__ codes,
    int32_t* __restrict__ status) {
  __shared__ int16_t s_pmap[NPAN_MAX];          // panel -> compact index of the touched panels (ascending), -1
  __shared__ uint32_t s_bits[TP_MAX][BP / 32];  // one bit per column of every touched panel
  __shared__ int32_t s_gpre[GROUPS_MAX + 1];    // distinct columns before each group of GW bitmap words
  __shared__ int16_t s_cpan[TP_MAX];            // compact index -> panel
  __shared__ int32_t s_cnt[NW][SEGW];           // OUT entries per (wave, list chunk)
  __shared__ int32_t s_nin[NW];                 // IN pairs per wave
  __shared__ int s_scan[THREADS / 64 + 1];
  __shared__ int32_t s_rp[RMAX + 1];            // row pointers of the block, relative to its first entry
  __shared__ int s_ord[SEGW];                   // processing order of the tiles: j -> chunk of the column list
  __shared__ int s_tmp[2][SEGW];
  __shared__ int32_t s_len[NW];                 // stream length of every wave (entries)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int b = blockIdx.x;
  const int row0 = blk_row[b];
  const int nrows = blk_row[b + 1] - row0;
  const int64_t e0 = rowptr[row0];
  const int64_t e1 = rowptr[row0 + nrows];
  const int npan = (int)((n_cols + BP - 1) / BP);
  // the block's own rows as columns: [cb0, cb0 + nrows)
  const int64_t cb0 = col_base + row0;
  auto in_block = [&](int c) { return sym && (uint64_t)((int64_t)c - cb0) < (uint64_t)nrows; };
  int32_t* segp = seg + (size_t)b * SEGROWS * SEGW;

  for (int i = tid; i < NPAN_MAX; i += THREADS) s_pmap[i] = 0;
  for (int i = tid; i < TP_MAX * (BP / 32); i += THREADS) (&s_bits[0][0])[i] = 0u;
  for (int i = tid; i < NW * SEGW; i += THREADS) (&s_cnt[0][0])[i] = 0;
  for (int i = tid; i <= nrows; i += THREADS) s_rp[i] = (int32_t)(rowptr[row0 + i] - e0);
  for (int i = tid; i < SEGROWS * SEGW; i += THREADS) segp[i] = 0;  // (an empty schedule, and the symmetry sum)
  __syncthreads();
  auto write_header = [&](int T_, int ndist_) {  // (one thread; behind the zeroing above)
    int32_t* h = segp + (SEGROWS - 1) * SEGW;
    h[2] = row0;
    h[3] = nrows;
    h[4] = T_;
    h[5] = ndist_;
    h[6] = (int32_t)(uint32_t)((uint64_t)e0 & 0xFFFFFFFFu);
    h[7] = (int32_t)(uint32_t)((uint64_t)e0 >> 32);
  };
  if (nrows == 0 || e1 == e0) {  // nothing to lay out (the step still writes y = (alpha dw + beta) x + gamma z for its rows)
    if (tid == 0) {
      blk_ntile[b] = 0;
      blk_ndist[b] = 0;
      write_header(0, 0);
    }
    return;
  }

  // pass A: which column panels does this block touch?  (8 loads in flight per thread: the passes over the
  // block's 0.6 MB of column indices are latency-bound otherwise)
  constexpr int MU = 8;
  for (int64_t eb = e0 + tid; eb < e1; eb += (int64_t)MU * THREADS) {
    int c[MU];
#pragma unroll
    for (int u = 0; u < MU; ++u) c[u] = col[min(eb + (int64_t)u * THREADS, e1 - 1)];
#pragma unroll
    for (int u = 0; u < MU; ++u)
      if (!in_block(c[u])) s_pmap[c[u] >> BP_BITS] = 1;  // (the clamped tail re-marks a valid entry)
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
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pn = tid * 4 + k;
      if (pn < NPAN_MAX) s_pmap[pn] = f[k] ? (int16_t)(pre++) : (int16_t)-1;
    }
    if (tid == 0) s_scan[0] = total;
  }
  __syncthreads();
  const int tp_all = s_scan[0];  // column panels the block touches
  __syncthreads();

  // The bitmap in LDS holds TP_MAX panels (786k columns' worth).  A block of a graph of up to ~1M cells touches
  // fewer; beyond that (2M cells in one chain order: 500+) the touched panels are handled TP_MAX at a time, in
  // ascending order -- passes B, the prefix sums, the column list and the walk below run once per group, ranks and
  // segment positions simply continue.  Within a (wave, tile) segment the entries are then ordered by (panel
  // group, row, column) instead of (row, column): a row's run may be cut in two at a group boundary (two flushes).
  constexpr int RU = 8;  // work items in flight
  const int nmine = (w < NW && nrows > w) ? (nrows - w + NW - 1) / NW : 0;
  const int64_t elast = max(e1 - 1, e0);
  int curs = 0;       // lane t: entries of list chunk t seen so far
  int in_cur = 0;     // (uniform) IN pairs of this wave so far
  int dist_base = 0;  // distinct OUT columns of the panel groups already done
  bool bad_range = false, bad_diag = false;
  const int n_groups_p = max(1, (tp_all + TP_MAX - 1) / TP_MAX);
  for (int pg = 0; pg < n_groups_p; ++pg) {
  const int pbase = pg * TP_MAX;
  const int tp = min(TP_MAX, tp_all - pbase);  // panels of this group (compact indices pbase .. pbase + tp - 1)
  auto panel_slot = [&](int c) -> int {        // row of the bitmap, or -1: not in this group
    const int ci = (int)s_pmap[c >> BP_BITS] - pbase;
    return ((unsigned)ci < (unsigned)TP_MAX) ? ci : -1;
  };
  if (pg > 0) {
    for (int i = tid; i < TP_MAX * (BP / 32); i += THREADS) (&s_bits[0][0])[i] = 0u;
  }
  for (int pn = tid; pn < npan; pn += THREADS) {
    const int ci = (int)s_pmap[pn] - pbase;
    if (s_pmap[pn] >= 0 && (unsigned)ci < (unsigned)TP_MAX) s_cpan[ci] = (int16_t)pn;
  }
  __syncthreads();

  // pass B: one bit per distinct OUT column
  for (int64_t eb = e0 + tid; eb < e1; eb += (int64_t)MU * THREADS) {
    int c[MU];
#pragma unroll
    for (int u = 0; u < MU; ++u) c[u] = col[min(eb + (int64_t)u * THREADS, e1 - 1)];
#pragma unroll
    for (int u = 0; u < MU; ++u) {
      if (in_block(c[u])) continue;
      const int ci = panel_slot(c[u]);
      if (ci >= 0) atomicOr(&s_bits[ci][(c[u] & (BP - 1)) >> 5], 1u << (c[u] & 31));
    }
  }
  __syncthreads();
  int ndist_g;
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
    int pre = block_exscan(c, s_scan, &ndist_g);
#pragma unroll
    for (int k = 0; k < GPT; ++k) {
      const int g = tid * GPT + k;
      if (g < ngroups) s_gpre[g] = pre;
      pre += cntk[k];
    }
    if (tid == 0) s_gpre[ngroups] = ndist_g;
  }
  __syncthreads();
  if ((dist_base + ndist_g + CP - 1) / CP > TMAX) {
    if (tid == 0) {
      atomicMax(status, 2);
      blk_ntile[b] = -1;
      blk_ndist[b] = 0;
    }
    return;
  }
  // rank of a column (of this panel group) among the block's distinct OUT columns
  auto col_rank = [&](int c, int ci) -> int {
    const int wd = (c & (BP - 1)) >> 5;
    int g = dist_base + s_gpre[ci * GPP + wd / GW];
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
      int g = dist_base + s_gpre[ci * GPP + wd / GW];
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
  // The walk: every owner wave goes through ITS rows in order, lanes = the row's entries (one coalesced load per
  // work item, RU items in flight), and counts its entries per tile.  A row's entries are sorted by column: its
  // in-block columns are one run of lanes in the middle, and among the other lanes the entries of one tile are
  // consecutive (skipping that run).  The per-tile counters of a wave live in ONE VGPR (lane t = list chunk t;
  // T <= 61) that the entry lanes read with ds_bpermute and the run heads update with ds_permute -- no loop over the
  // tiles of a row, no LDS memory in the dependent chain.
  if (w < NW) {
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
    const uint64_t below = lanes_below(lane);
    while (busy) {
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int k = ik[u];
        const int off = ioff[u], rend = iend[u];
        const int c = cc[u];
        fetch(u);
        const bool act = (k < nmine) && (off + lane < rend);  // (the active lanes are 0 .. n - 1)
        const int rl_self = w + k * NW;
        const int cg = (int)((int64_t)c - cb0);  // column relative to the block's own rows
        const bool inb = act && in_block(c);
        const bool first = pg == 0;             // the in-block lanes are dealt with in the first panel group
        const bool upper = first && inb && cg > rl_self;
        const int ci = (act && !inb) ? panel_slot(c) : -1;
        const bool outl = ci >= 0;
        bad_diag |= first && inb && cg == rl_self;
        // ---- OUT lanes: runs of equal list chunk among the OUT lanes (non-decreasing along them)
        const uint64_t om = __ballot(outl);
        const int g = outl ? col_rank(c, ci) : 0;
        const int ch = g >> CP_BITS;
        const uint64_t ob = om & below;
        const int pl = ob ? 63 - __clzll((unsigned long long)ob) : 0;  // previous OUT lane
        const int chp = __shfl(ch, pl, 64);
        const bool head = outl && (ob == 0 || chp != ch);
        const uint64_t hm = __ballot(head);
        const uint64_t hb = hm & (below | ((uint64_t)1 << lane));
        const int hl = hb ? 63 - __clzll((unsigned long long)hb) : 0;                  // head lane of my run
        const int before = __popcll(ob & ~lanes_below(hl));                              // OUT lanes in [hl, lane)
        const uint64_t habove = (lane == 63) ? 0 : (hm >> (lane + 1));
        const uint64_t upto = habove ? lanes_below(lane + (int)__ffsll((unsigned long long)habove)) : ~(uint64_t)0;
        const int runlen = __popcll(om & ~below & upto);                                  // OUT lanes in [lane, next head)
        const int pos = __builtin_amdgcn_ds_bpermute(ch << 2, curs) + before;           // position in the segment
        // ---- IN lanes
        const uint64_t um = __ballot(upper);
        const int ipos = in_cur + __popcll(um & below);
        const bool ilast = upper && (um >> lane) == 1;  // highest upper lane of this part
        if (outl) {
          bad_range |= pos > 0x7FFF;
          codes[e0 + off + lane] = ((uint32_t)ch << 26) | ((uint32_t)(runlen == 1) << 25) | ((uint32_t)(pos & 0x7FFF) << CP_BITS) |
                                   (uint32_t)(g & (CP - 1));
        } else if (upper) {
          bad_range |= ipos > 0x1FFF;
          codes[e0 + off + lane] = ((uint32_t)T_IN << 26) | ((uint32_t)(ipos & 0x1FFF) << 13) | ((uint32_t)ilast << 12) | (uint32_t)cg;
        } else if (first && inb) {
          codes[e0 + off + lane] = ((uint32_t)T_SKIP << 26) | (uint32_t)(cg & 0xFFF);
        }
        // every run head adds the length of its run to the counter of its tile (lane 63 is the dump of the other
        // lanes: list chunks are <= 60)
        curs += __builtin_amdgcn_ds_permute((head ? ch : 63) << 2, head ? runlen : 0);
        in_cur += __popcll(um);
        if (u == RU - 1) busy = ik[0] < nmine;  // (uniform) items are in order: nothing left once slot 0 is past the end
      }
    }
  }
  dist_base += ndist_g;
  __syncthreads();  // (the bitmap is reused by the next panel group)
  }  // panel groups
  const int ndist = dist_base;
  const int T = (ndist + CP - 1) / CP;
  if (w < NW) {
    if (__any(bad_range)) atomicMax(status, 4);
    if (__any(bad_diag)) atomicMax(status, 6);
    s_cnt[w][lane] = (lane < T) ? curs : 0;
    if (lane == 0) s_nin[w] = in_cur;
  }
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
    blk_ntile[b] = T;
    blk_ndist[b] = ndist;
    write_header(T, ndist);
  }
  __syncthreads();
  // The chunk schedule of every wave (one thread per wave).  Stream of a wave = its IN pairs padded to whole chunks,
  // then its OUT entries, dense, tile by tile in processing order.  One descriptor per chunk: the IN chunks (64
  // pairs each), then for every tile the chunks of the wave's segment -- the last one partial -- with the first / last
  // marks the ring protocol hangs on (a tile without entries of this wave gets one empty chunk).
  if (tid < NW) {
    const int ow = tid;
    const int nin = s_nin[ow];
    const int n_in_chunks = (nin + 63) >> 6;
    int32_t* bounds = segp + ow * SEGW;            // row ow: header words
    int32_t* eoff = segp + (NW + 1 + ow) * SEGW;   // row NW + 1 + ow: entry offset of every tile's segment
    uint16_t* cd = cdesc + ((size_t)b * NW + ow) * (KMAX + KSLACK);
    int k = 0, pos = 0;
    for (; k < n_in_chunks && k < KMAX; ++k) cd[k] = 64;
    bool over = n_in_chunks > KMAX;
    for (int j = 0; j < T; ++j) {
      const int cnt = s_cnt[ow][s_ord[j]];
      eoff[j] = pos;
      const int nch = max(1, (cnt + 63) >> 6);
      for (int c = 0; c < nch; ++c) {
        const int n = min(64, cnt - 64 * c);
        if (k < KMAX) cd[k] = (uint16_t)(max(n, 0) | (c == 0 ? D_FIRST : 0) | (c == nch - 1 ? D_LAST : 0));
        else over = true;
        ++k;
      }
      pos += cnt;
    }
    eoff[T] = pos;  // (T <= 61)
    for (int q = min(k, KMAX); q < KMAX + KSLACK; ++q) cd[q] = 0;
    eoff[62] = nin;
    eoff[63] = pos;
    bounds[62] = n_in_chunks | (min(k, KMAX) << 16);
    s_len[ow] = (n_in_chunks << 6) + pos;
    if (over) atomicMax(status, 4);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int ow = 0; ow < NW; ++ow) {
      segp[ow * SEGW + 63] = run;  // the wave's offset in the block's stream
      run += s_len[ow];
    }
  }
  if (w == NW) segp[NW * SEGW + lane] = s_ord[lane];  // row NW: list chunk of every processed tile
}

// Second step of the layout: one workgroup per (block, owner wave) places the wave's entries.  Every entry knows its
// place (the code written by pt_build_kernel + the schedule), so this step has no ordering constraint: the entries are
// read in CSR order (coalesced), dropped at their final place in an LDS image of the wave's stream, and the image is
// copied out with full-line stores.
//   index word of an OUT entry:  flush flag | ring slot of its column ((tile % NB) * CP + tile-local column) << 4 | row slot << 20
//   index word of an IN pair:    flush flag | slot of row j << 4 | slot of row i << 20      (row_slot())
// (the two 12-bit fields are LDS byte offsets once masked / shifted: x & 0xFFF0 and x >> 16).  IN pairs are laid out
// in windows of QW chunks, TRANSPOSED over the 64 lanes: lane l owns QW consecutive pairs of the (row, column) order
// and chunk c holds every lane's c-th pair, so a lane meets the pairs of a row in consecutive chunks and sums the
// i side of the run in registers; the flag marks a lane's last pair of a row (or of the window).
// Symmetry check: every upper in-block entry adds a hash of (i, j, value bits), every lower one subtracts the hash of
// (j, i, value bits); the 64-bit sums of a symmetric block cancel.
constexpr int FILL_CAP = 12288;  // entries of the LDS image (12 B each): 192 chunks
__device__ __forceinline__ uint64_t pair_hash(uint32_t i, uint32_t j, double v) {
  uint64_t h = ((uint64_t)i << 32 | j) * 0x9E3779B97F4A7C15ull;
  h ^= (uint64_t)__double_as_longlong(v) * 0xC2B2AE3D27D4EB4Full;
  h ^= h >> 29;
  h *= 0xBF58476D1CE4E5B9ull;
  h ^= h >> 32;
  return h;
}
__global__ __launch_bounds__(THREADS) void pt_fill_kernel(const int64_t* __restrict__ rowptr, const double* __restrict__ val,
                                                          const uint32_t* __restrict__ codes, const int32_t* __restrict__ blk_row,
                                                          const int32_t* __restrict__ blk_ntile, int32_t* __restrict__ seg,
                                                          double* __restrict__ pval, uint32_t* __restrict__ pidx) {
  __shared__ double s_val[FILL_CAP];
  __shared__ uint32_t s_idx[FILL_CAP];
  __shared__ int s_eoff[SEGW], s_inv[SEGW];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.x / NW, w = blockIdx.x % NW;
  const int T = blk_ntile[b];
  if (T < 0) return;
  const int row0 = blk_row[b];
  const int nrows = blk_row[b + 1] - row0;
  if (nrows == 0) return;
  const int32_t* segp = seg + (size_t)b * SEGROWS * SEGW;
  const int hdr = segp[w * SEGW + 62];
  const int n_in_chunks = hdr & 0xFFFF;
  const int soff = segp[w * SEGW + 63];
  const int64_t sbase = stream_base(rowptr, row0, b) + soff;
  if (tid < SEGW) {
    s_eoff[tid] = segp[(NW + 1 + w) * SEGW + tid];
    if (tid < T) s_inv[segp[NW * SEGW + tid]] = tid;  // list chunk -> processing position
  }
  __syncthreads();
  const int nmine = (nrows > w) ? (nrows - w + NW - 1) / NW : 0;  // rows of owner wave w
  const int in_base = n_in_chunks << 6;
  const int total = in_base + s_eoff[63];  // entries of the wave's stream
  uint64_t hsum = 0;
  for (int a0 = 0; a0 < total; a0 += FILL_CAP) {  // slot ranges that fit the LDS image (one, as a rule)
    const int span = min(FILL_CAP, total - a0);
    for (int i = tid; i < span; i += THREADS) {  // padding pattern: value 0, column slot 0, row 0, no flush
      s_val[i] = 0.0;
      s_idx[i] = 0u;
    }
    __syncthreads();
    auto place = [&](uint32_t code, double v, int rl) __attribute__((always_inline)) {
      const int tl = (int)(code >> 26);
      int slot;
      uint32_t word;
      if (tl == T_SKIP) {
        if (a0 == 0) hsum -= pair_hash(code & 0xFFF, (uint32_t)rl, v);
        return;
      } else if (tl == T_IN) {
        const int g = (code >> 13) & 0x1FFF;
        const int win = g / (64 * QW), u = g - win * (64 * QW);
        const int qw = min(QW, n_in_chunks - win * QW);  // chunks of this window (the last one may be short)
        const int l = u / qw, c = u - l * qw;
        slot = win * (64 * QW) + c * 64 + l;
        const uint32_t flush = ((code >> 12) & 1u) | (uint32_t)(c == qw - 1);
        const uint32_t rj = code & 0xFFF;
        word = flush | ((uint32_t)row_slot((int)rj) << 4) | ((uint32_t)row_slot(rl) << 20);
        if (a0 == 0) hsum += pair_hash((uint32_t)rl, rj, v);
      } else {
        // final place within its (wave, tile) segment of n = 64 q + r entries: TRANSPOSED over the 64 lanes -- lane l
        // owns a contiguous run of the segment's (row, column) order (q + 1 entries for l < r, else q) and chunk c
        // holds each lane's c-th entry
        const int j = s_inv[tl];
        const int pos = (int)((code >> CP_BITS) & 0x7FFF);
        const int n = s_eoff[j + 1] - s_eoff[j];
        const int q = n >> 6, r = n & 63;
        int l, c, mine;
        if (pos < r * (q + 1)) {
          l = pos / (q + 1);
          c = pos - l * (q + 1);
          mine = q + 1;
        } else {
          const int jj = pos - r * (q + 1);
          const int lq = jj / max(q, 1);
          l = r + lq;
          c = jj - lq * q;
          mine = q;
        }
        slot = in_base + s_eoff[j] + c * 64 + l;
        const uint32_t flush = ((code >> 25) & 1u) | (uint32_t)(c == mine - 1);
        const uint32_t cs = (uint32_t)((j & (NB - 1)) * CP) + (code & (CP - 1));
        word = flush | (cs << 4) | ((uint32_t)row_slot(rl) << 20);
      }
      slot -= a0;
      if (slot >= 0 && slot < span) {
        s_val[slot] = v;
        s_idx[slot] = word;
      }
    };
    // this workgroup's 16 waves share the owner wave's rows: k = wv, wv + 16, ...; FG rows in flight per wave
    constexpr int FG = 4;
    for (int k0 = wv; k0 < nmine; k0 += 16 * FG) {
      uint32_t code[FG];
      double v[FG];
      int64_t rs[FG], rend[FG];
#pragma unroll
      for (int g = 0; g < FG; ++g) {
        const int k = k0 + 16 * g;
        const int rl = w + min(k, nmine - 1) * NW;
        rs[g] = rowptr[row0 + rl];
        rend[g] = (k < nmine) ? rowptr[row0 + rl + 1] : rs[g];
        const int64_t e = min(rs[g] + lane, max(rend[g] - 1, rs[g]));
        code[g] = codes[e];
        v[g] = val[e];
      }
#pragma unroll
      for (int g = 0; g < FG; ++g) {
        const int rl = w + (k0 + 16 * g) * NW;
        if (rs[g] + lane < rend[g]) place(code[g], v[g], rl);
        for (int64_t e = rs[g] + 64 + lane; e < rend[g]; e += 64) place(codes[e], val[e], rl);
      }
    }
    __syncthreads();
    for (int i = tid; i < span; i += THREADS) {
      pval[sbase + a0 + i] = s_val[i];
      pidx[sbase + a0 + i] = s_idx[i];
    }
    __syncthreads();
  }
  // (wrapping 64-bit sums: the order of the additions does not matter)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) hsum += __shfl_xor((unsigned long long)hsum, off, 64);
  if (lane == 0 && hsum != 0) atomicAdd(symsum_of(seg, b), (unsigned long long)hsum);
}

// a block whose symmetry sum is not zero -> status 5 (W is not bitwise symmetric inside that block's own square)
__global__ __launch_bounds__(256) void pt_symcheck_kernel(int32_t* __restrict__ seg, int nb, int32_t* __restrict__ status) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb && *symsum_of(seg, b) != 0ull) atomicMax(status, 5);
}

// pval32[e] = (float)pval[e] over the streams (n entries)
__global__ __launch_bounds__(256) void pt_round_f32_kernel(const double* __restrict__ pval, float* __restrict__ pval32, int64_t n) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += (int64_t)gridDim.x * blockDim.x * 2) {
    if (i + 1 < n) {
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
// timing-only ablations for tools/spmm_compare.py (results are wrong while set): 4 no panel loads, 8 panel gathers
// from a 16 KB window of x
extern "C" int meld_pt_debug_ablate(int mask) {
  g_pt_ablate = mask;
  return MELD_OK;
}

static unsigned long long* g_pt_stamps = nullptr;
// development: every wave of the following step launches writes 8 wall-clock stamps (100 MHz) to buf[nb][16][8]
// (0 start, 1 before / 2 after the opening barrier, 3 IN part done, 4 stream done, 5 before / 6 after the closing
// barrier, 7 results written); NULL switches it off
extern "C" int meld_pt_debug_stamps(unsigned long long* buf) {
  g_pt_stamps = buf;
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
  // one block per CU and launch round: about 0.9575 RMAX rows each (3907 at 1M) leaves room for balancing the nonzeros
  const int64_t target = (int64_t)(0.9576 * pt::RMAX);
  if (n_rows <= 64 * 256) return (int)std::max<int64_t>(1, ceil_div(n_rows, 256));
  const int64_t k = ceil_div(n_rows, 256 * target);
  int64_t nb = 256 * k;
  if (k == 1) nb = std::min<int64_t>(256, std::max<int64_t>(64, ceil_div(n_rows, 256)));
  while (nb * pt::RMAX < n_rows) nb += 256;
  return (int)nb;
}

extern "C" int64_t meld_pt_seg_len(int nb) { return (int64_t)nb * pt::SEGROWS * pt::SEGW; }
// 16-bit chunk descriptors of the layout
extern "C" int64_t meld_pt_desc_len(int nb) { return (int64_t)nb * pt::NW * (pt::KMAX + pt::KSLACK); }

// entries the stream arrays (pval, pval32, pidx) must hold for a matrix of nnz entries in nb blocks: every block's
// stream starts at its CSR offset + b * NW * PADCAP; the consumer waves read up to U chunks past the last stream
extern "C" int64_t meld_pt_stream_len(int64_t nnz, int nb) {
  return nnz + (int64_t)nb * pt::NW * pt::PADCAP + (int64_t)(pt::U + 1) * 64;
}

extern "C" int meld_pt_build(const int64_t* rowptr, const int32_t* col, const double* val, int64_t n_rows, int64_t n_cols,
                             int64_t col_base, int symmetric, const meld_pt_layout_t* layout, uint32_t* codes,
                             int32_t* status, meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && col && val && layout && layout->blk_row && layout->blk_ntile && layout->blk_ndist && layout->seg &&
                     layout->list_cols && layout->pval && layout->pidx && layout->cdesc && codes && status && n_rows > 0 && layout->nb > 0 &&
                     n_cols > 0 && layout->stream_len > 0 && col_base >= 0,
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
  int32_t* seg = const_cast<int32_t*>(layout->seg);
  // plan: by estimated time when the scratch for the weights fits the (not yet written) value stream, by entries otherwise
  {
    static const bool by_entries = meld_dev_getenv("MELD_PT_PLAN") && !strcmp(meld_dev_getenv("MELD_PT_PLAN"), "entries");
    const size_t scan_bytes = meld_scan_temp_bytes(n_rows);
    const size_t off_w = sizeof(int64_t) * (size_t)(n_rows + 1), off_t = (off_w + sizeof(int32_t) * (size_t)n_rows + 255) / 256 * 256;
    if (!by_entries && nb > 1 && off_t + scan_bytes <= sizeof(double) * (size_t)layout->stream_len) {
      char* scratch = reinterpret_cast<char*>(const_cast<double*>(layout->pval));
      int64_t* wpre = reinterpret_cast<int64_t*>(scratch);
      int32_t* w = reinterpret_cast<int32_t*>(scratch + off_w);
      const int win = (int)std::max<int64_t>(1, n_rows / nb / 2);
      hipLaunchKernelGGL(pt::pt_row_weight_kernel, dim3((unsigned)ceil_div(n_rows * 16, 256)), dim3(256), 0, st, rowptr, col, n_rows,
                         col_base, win, w);
      const int rc = meld_exclusive_scan_i32_i64(w, wpre, n_rows, scratch + off_t, scan_bytes, stream);
      if (rc != MELD_OK) return rc;
      hipLaunchKernelGGL(pt::pt_plan_weighted_kernel, dim3(1), dim3(1024), 0, st, wpre, n_rows, nb, const_cast<int32_t*>(layout->blk_row));
    } else {
      hipLaunchKernelGGL(pt::pt_plan_kernel, dim3(1), dim3(1024), 0, st, rowptr, n_rows, nb, const_cast<int32_t*>(layout->blk_row));
    }
  }
  hipLaunchKernelGGL(pt::pt_build_kernel, dim3(nb), dim3(pt::THREADS), 0, st, rowptr, col, n_cols, col_base, symmetric ? 1 : 0, nb,
                     layout->blk_row, const_cast<int32_t*>(layout->blk_ntile), const_cast<int32_t*>(layout->blk_ndist), seg,
                     const_cast<uint16_t*>(layout->cdesc), const_cast<int32_t*>(layout->list_cols), codes, status);
  hipLaunchKernelGGL(pt::pt_fill_kernel, dim3(nb * pt::NW), dim3(pt::THREADS), 0, st, rowptr, val, codes, layout->blk_row,
                     layout->blk_ntile, seg, const_cast<double*>(layout->pval), const_cast<uint32_t*>(layout->pidx));
  if (symmetric)
    hipLaunchKernelGGL(pt::pt_symcheck_kernel, dim3((unsigned)ceil_div(nb, 256)), dim3(256), 0, st, seg, nb, status);
  if (layout->pval32 != nullptr)  // (its own streaming pass: a third scattered store in the walk costs 0.5 ms, this 0.08)
    hipLaunchKernelGGL(pt::pt_round_f32_kernel, dim3(2048), dim3(256), 0, st, layout->pval, const_cast<float*>(layout->pval32),
                       layout->stream_len);
  MELD_LAUNCH_CHECK("pt_build_kernel");
  return MELD_OK;
}

namespace {
template <int P, bool F32>
int pt_launch(const pt::StepArgs& a, hipStream_t st) {
  constexpr size_t lds = pt::LDS_BYTES;  // 128 KiB + control words
  // the attribute is per device, and a process may drive several
  static bool configured[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!configured[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pt::pt_step_kernel<P, F32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_err("pt_step_kernel: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
      return MELD_ERR_HIP;
    }
    configured[dev] = true;
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
                  double coef, double* dots, const double* coef_dev, hipStream_t st, double coef_x, const PtLanczos* lz) {
  if (L->nb == 0) return MELD_OK;
  pt::StepArgs a;
  a.blk_row = L->blk_row; a.blk_ntile = L->blk_ntile; a.blk_ndist = L->blk_ndist; a.seg = L->seg;
  a.list_cols = L->list_cols; a.pval = L->pval; a.pidx = L->pidx; a.cdesc = L->cdesc; a.rowptr = rowptr; a.dw = dw; a.x_full = x_full;
  a.z = z; a.y = y; a.r = r; a.dots = dots; a.coef_dev = coef_dev; a.x_row_offset = x_row_offset; a.alpha = alpha;
  a.beta = beta; a.gamma = gamma; a.coef = coef; a.coef_x = coef_x; a.nb = L->nb; a.ld = p; a.colofs = 0;
  a.ablate = g_pt_ablate;
  a.stamps = g_pt_stamps;
  a.pval32 = L->pval32;
  if (lz != nullptr) a.lz = *lz; else a.lz = PtLanczos{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr};
  // the fp32 copy of the values serves the lmax estimate only (p = 1 with device-resident Lanczos scalars)
  return pt_step_cols(a, p, st, (coef_dev != nullptr || lz != nullptr) && p == 1 && L->pval32 != nullptr);
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

// Steps k = 2 .. n_coef - 1 of the Chebyshev recurrence in one call (single GPU: x_row_offset = 0, no collective between
// the steps):  T_k = alpha2 L T_{k-1} + beta2 T_{k-1} - T_{k-2},  r += c_k T_k  [UPSTREAM pygsp cheby_op, reference
// meld/filter.py:59].  t_prev2 / t_prev1 hold T_0 / T_1 on entry and are used as the two ping-pong buffers (T_k
// overwrites T_{k-2}); r already holds c_0 / 2 T_0 + c_1 T_1.  The accumulator is touched every OTHER step only: a step
// that holds T_k in its result and T_{k-1} in its own rows of the iterate adds c_k T_k + c_{k-1} T_{k-1} at once, the
// step before it neither reads nor writes r -- 32 of the 80 bytes of vector traffic per row and step pair at p = 2.
// coeffs: n_coef doubles on the HOST.  *last = 0 / 1: the buffer (t_prev2 / t_prev1) that holds T of the last order.
extern "C" int meld_pt_cheby_run(const meld_pt_layout_t* layout, const int64_t* rowptr, const double* dw, int64_t n_rows, int p,
                                 double* t_prev2, double* t_prev1, double* r, const double* coeffs, int n_coef, double alpha2,
                                 double beta2, int* last, meld_stream_t stream) {
  MELD_CHECK_ARG(layout && layout->blk_row && layout->blk_ntile && layout->blk_ndist && layout->seg && layout->list_cols &&
                     layout->pval && layout->pidx && rowptr && dw && t_prev2 && t_prev1 && r && coeffs && n_rows >= 0 && p >= 1 &&
                     n_coef >= 2,
                 "meld_pt_cheby_run: bad arguments");
  hipStream_t st = S(stream);
  double* t_old = t_prev2;
  double* t_cur = t_prev1;
  int which = 1;  // t_cur is t_prev1
  if (n_rows > 0) {
    int k = 2;
    if ((n_coef - 2) % 2 == 1) {  // an odd number of steps: the first one alone
      const int rc = pt_step(layout, rowptr, dw, p, t_cur, 0, t_old, t_old, r, alpha2, beta2, -1.0, coeffs[k], nullptr, nullptr, st, 0.0);
      if (rc != MELD_OK) return rc;
      std::swap(t_old, t_cur);
      which ^= 1;
      ++k;
    }
    for (; k + 1 < n_coef; k += 2) {
      // T_k: no accumulator traffic
      int rc = pt_step(layout, rowptr, dw, p, t_cur, 0, t_old, t_old, nullptr, alpha2, beta2, -1.0, 0.0, nullptr, nullptr, st, 0.0);
      if (rc != MELD_OK) return rc;
      std::swap(t_old, t_cur);
      // T_{k+1}, and r += c_{k+1} T_{k+1} + c_k T_k (T_k = this step's own rows of the iterate)
      rc = pt_step(layout, rowptr, dw, p, t_cur, 0, t_old, t_old, r, alpha2, beta2, -1.0, coeffs[k + 1], nullptr, nullptr, st, coeffs[k]);
      if (rc != MELD_OK) return rc;
      std::swap(t_old, t_cur);
    }
  }
  if (last) *last = which;
  MELD_LAUNCH_CHECK("pt_step_kernel");
  return MELD_OK;
}
