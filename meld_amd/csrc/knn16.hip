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
// B fragments, 64 VGPRs); the waves share the reference tile stream (64 refs, 16 KiB: hi and lo
// planes) double-buffered in LDS.  One A fragment (ds_read_b128 x2) feeds two MFMA chains
// (the two query groups), so the matrix pipe never waits on a dependent accumulator.  Three
// workgroups are resident per CU (157 VGPRs, 47 KiB LDS): a selection slow path in one
// workgroup stalls its four waves at the tile barrier, and the other workgroups keep the matrix
// pipe fed (measured: 8-wave workgroups, one per CU, left the pipe 65 % idle).  L2 -> LDS: each
// tile is read once per workgroup: 16 KiB per 1536 matrix-pipe cycles x 3 ~ 32 B/clk/CU.
//
// Selection: as in knn.hip (threshold per query, append to the row buffer, compact when full),
// but compaction finds the new threshold by a 32-step radix select on the ordered float bits and
// squeezes the survivors with ballots (~5x cheaper than ranking); rows are ranked once, at the end.
#include "common.hpp"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>

namespace meld {

constexpr int K16_TS = 64;         // references per LDS tile
constexpr int K16_BQ = 256;        // queries per workgroup
constexpr int K16_THREADS = 256;   // 4 waves (64 queries each); 2-3 workgroups resident per CU
constexpr int K16_NWAVE = K16_THREADS / 64;
constexpr int K16_SLACK = 128;     // CAP = ksel + slack
constexpr int K16_CAPMAX = 256;
constexpr int K16_SLOTS = K16_CAPMAX / 64;  // row entries per lane in the compaction routines
constexpr float K16_BIG = 1.0e30f;  // squared norm of the padding references ("infinitely far")

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// v_min3_f32 without the canonicalising v_max that hipcc puts in front of fminf on MFMA outputs
// (the accumulators never hold signalling NaNs): 8 instructions for the minimum of 16 values
// instead of ~36.  Measured: 366 -> 347 ms at 1M cells.
__device__ __forceinline__ float min3f(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float min16(const float (&v)[16]) {
  const float a = min3f(v[0], v[1], v[2]);
  const float b = min3f(v[3], v[4], v[5]);
  const float c = min3f(v[6], v[7], v[8]);
  const float d = min3f(v[9], v[10], v[11]);
  const float e = min3f(v[12], v[13], v[14]);
  const float f = min3f(a, b, v[15]);
  const float g = min3f(c, d, e);
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(f), "v"(g));
  return r;
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

// Keep (unsorted) the entries of a candidate row whose d2 is <= the ksel-th smallest d2.
// Returns the new count through *n_out and the threshold as return value.  Wave-uniform args.
__device__ float knn16_squeeze_row(int n, int ksel, int cap, float* __restrict__ d2row, int* __restrict__ idxrow,
                                   int lane, int* n_out) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's append stores have reached L2
  float d[K16_SLOTS];
  int ix[K16_SLOTS];
  unsigned key[K16_SLOTS];
#pragma unroll
  for (int e = 0; e < K16_SLOTS; ++e) {
    const int p = lane + 64 * e;
    if (p < n) {
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
  // survivors: key <= T (ties at T all stay; if that overflows the row the caller re-ranks)
  int base = 0;
  float thr = INFINITY;
#pragma unroll
  for (int e = 0; e < K16_SLOTS; ++e) {
    const bool keep = key[e] <= T && (lane + 64 * e) < n;
    const unsigned long long b = __ballot(keep);
    if (keep) {
      const int pos = base + __popcll(b & ((1ull << lane) - 1ull));
      if (pos < cap) {
        d2row[pos] = d[e];
        idxrow[pos] = ix[e];
      }
    }
    base += __popcll(b);
    const unsigned long long bt = __ballot(keep && key[e] == T);
    if (bt) thr = __shfl(d[e], __ffsll((long long)bt) - 1, 64);
  }
  *n_out = min(base, cap);
  return thr;
}

// Final ordering of a row: rank by (d2, idx), keep the ksel smallest sorted, scale back.
__device__ void knn16_rank_row(int n, int ksel, float out_scale, float* __restrict__ d2row, int* __restrict__ idxrow,
                               float* sd, int* si, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float d[K16_SLOTS];
  int ix[K16_SLOTS];
#pragma unroll
  for (int e = 0; e < K16_SLOTS; ++e) {
    const int p = lane + 64 * e;
    if (p < n) {
      d[e] = ld_sc1_f(d2row + p);
      ix[e] = ld_sc1_i(idxrow + p);
    } else {
      d[e] = INFINITY;
      ix[e] = 0x7fffffff;
    }
    sd[p] = d[e];
    si[p] = ix[e];
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
    const int p = lane + 64 * q;
    if (p < n && rk[q] < ksel) {
      d2row[rk[q]] = d[q] * out_scale;
      idxrow[rk[q]] = ix[q];
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int KB, int ABL, int NPROD>  // KP16 = 16 * KB >= d + 2; ABL: 0 = product, 1 / 3 = profiling ablations;
                                      // NPROD: split products on the coordinate K blocks (1 = hi.hi only, 3 = hi.hi + hi.lo + lo.hi)
__global__ __launch_bounds__(K16_THREADS, (KB <= 4 ? 3 : 2)) void knn16_topk_kernel(
    const _Float16* __restrict__ Q16, const float* __restrict__ Qn, const _Float16* __restrict__ Rt16,
    const float* __restrict__ scale_info, int n_ref, int n_tiles, int ksel, int cap, const float* __restrict__ lb2, const float* __restrict__ norm2_max,
    float err_coef, int tile_origin, int* __restrict__ cand_idx, float* __restrict__ cand_d2,
    int* __restrict__ cand_cnt) {
  // reference tile = KB coordinate blocks [kb][k-half][plane][ref][8 halves] followed by the 64
  // squared norms (fp32): the norms are added in the epilogue instead of riding through the MFMAs
  constexpr int TILE_H = KB * 2 * 2 * K16_TS * 8 + 2 * K16_TS;  // halves per reference tile (a norm = 2 halves)
  constexpr int TILE_V4 = TILE_H / 8;                           // 16-byte vectors per tile = KB * 256 + 16
  constexpr int NV = TILE_V4 / K16_THREADS;                     // full rounds of the 256 threads = KB
  constexpr bool HAS_TAIL = (TILE_V4 % K16_THREADS) != 0;       // the 16 norm vectors
  static_assert(NV <= 8, "tile too large for the staging registers");
  static_assert(K16_THREADS == 256 && K16_TS == 64 && NV == KB && HAS_TAIL,
                "the plane-skipping staging assumes vector u*256+tid = K block u, followed by 16 norm vectors");

  __shared__ __attribute__((aligned(16))) _Float16 lds_tile[2][TILE_H];
  __shared__ int lds_cnt[K16_NWAVE][64];
  __shared__ float lds_sd[K16_NWAVE][K16_CAPMAX];
  __shared__ int lds_si[K16_NWAVE][K16_CAPMAX];
  __shared__ float lds_wthr[2][K16_NWAVE];  // per-wave max threshold, double-buffered by step parity

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int jq = lane & 31;
  const int h = lane >> 5;
  const int q_base = blockIdx.x * K16_BQ + wave * 64;  // first query of this wave
  // reference slices (gridDim.y > 1): slice y scans tiles [tile_lo, tile_lo + n_scan) and writes its
  // own candidate rows; meld_knn16_merge_slices combines them.  Used to spread a small query set
  // (the second-stage re-search) over the whole chip.
  const int tile_lo = (int)((long long)n_tiles * blockIdx.y / gridDim.y);
  const int n_scan = (int)((long long)n_tiles * (blockIdx.y + 1) / gridDim.y) - tile_lo;
  const int row_base = (int)(blockIdx.y * (gridDim.x * K16_BQ)) + q_base;  // first candidate row of this wave

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
  float nq[2];
  nq[0] = Qn[q_base + jq];
  nq[1] = Qn[q_base + 32 + jq];

  lds_cnt[wave][lane] = 0;
  if (lane == 0) lds_wthr[0][wave] = lds_wthr[1][wave] = INFINITY;
  float thr[2] = {INFINITY, INFINITY};
  float thrp[2] = {INFINITY, INFINITY};  // thr - |q|^2
  float wmax = INFINITY;  // max threshold over this wave's 64 queries (wave-uniform)

  // Tiles are visited starting at the workgroup's own position (its spatial neighbourhood when
  // the cells are in locality order) and wrapping around: step s -> tile (t0 + s) mod n_tiles.
  // lb2 (optional) holds, per (workgroup, tile), a lower bound on the squared distance between
  // any query of the workgroup and any reference of the tile; a tile whose bound exceeds every
  // threshold of the workgroup cannot contribute a candidate and is skipped without being loaded.
  // search-error allowance in the scaled space: a skipped tile must fail `d2_approx < thr` for sure
  const float prune_margin = lb2 ? err_coef * norm2_max[0] * scale_info[0] * scale_info[0] : 0.0f;
  const int t0 = (int)(((long long)tile_origin + (long long)blockIdx.x * (K16_BQ / K16_TS)) % n_scan);
  const float* my_lb = lb2 ? lb2 + (size_t)blockIdx.x * n_tiles : nullptr;
  auto tile_of = [&](int s) {
    const int t = t0 + s;
    return tile_lo + (t >= n_scan ? t - n_scan : t);
  };
  // first step >= s whose tile may hold a candidate given the block-wide threshold bound
  auto next_live = [&](int s, float bound) {
    if (my_lb == nullptr) return s;
    for (int base = s; base < n_scan; base += 64) {
      const int ss = base + lane;
      const bool live = ss < n_scan && my_lb[tile_of(ss)] <= bound;
      const unsigned long long b = __ballot(live);
      if (b) return base + (int)__ffsll((long long)b) - 1;
    }
    return n_scan;
  };

  const float4* R4 = reinterpret_cast<const float4*>(Rt16);
  const bool tail_ok = HAS_TAIL && (NV * K16_THREADS + tid < TILE_V4);
  float4 p0, p1, p2, p3, p4, p5, p6, p7, pt;
  p0 = p1 = p2 = p3 = p4 = p5 = p6 = p7 = pt = make_float4(0.f, 0.f, 0.f, 0.f);
  // Vector v = tid + u * 256 of a tile is (kb = u, k-half, plane, ref) = (u, tid >> 7, (tid >> 6) & 1, tid & 63):
  // a wave copies one plane.  With NPROD == 1 the lo planes of the coordinate blocks are never read,
  // so the two waves that own them skip those copies (3/8 of the staging traffic).
  const bool lo_wave = ((tid >> 6) & 1) != 0;
#define K16_NEED(U) (NPROD == 3 || !lo_wave)
#define K16_LOAD(SRC)                                                            \
  do {                                                                           \
    if constexpr (NV > 0) if (K16_NEED(0)) p0 = (SRC)[tid + 0 * K16_THREADS];    \
    if constexpr (NV > 1) if (K16_NEED(1)) p1 = (SRC)[tid + 1 * K16_THREADS];    \
    if constexpr (NV > 2) if (K16_NEED(2)) p2 = (SRC)[tid + 2 * K16_THREADS];    \
    if constexpr (NV > 3) if (K16_NEED(3)) p3 = (SRC)[tid + 3 * K16_THREADS];    \
    if constexpr (NV > 4) if (K16_NEED(4)) p4 = (SRC)[tid + 4 * K16_THREADS];    \
    if constexpr (NV > 5) if (K16_NEED(5)) p5 = (SRC)[tid + 5 * K16_THREADS];    \
    if constexpr (NV > 6) if (K16_NEED(6)) p6 = (SRC)[tid + 6 * K16_THREADS];    \
    if constexpr (NV > 7) if (K16_NEED(7)) p7 = (SRC)[tid + 7 * K16_THREADS];    \
    if (tail_ok) pt = (SRC)[tid + NV * K16_THREADS];                             \
  } while (0)
#define K16_STORE(DST)                                                           \
  do {                                                                           \
    if constexpr (NV > 0) if (K16_NEED(0)) (DST)[tid + 0 * K16_THREADS] = p0;    \
    if constexpr (NV > 1) if (K16_NEED(1)) (DST)[tid + 1 * K16_THREADS] = p1;    \
    if constexpr (NV > 2) if (K16_NEED(2)) (DST)[tid + 2 * K16_THREADS] = p2;    \
    if constexpr (NV > 3) if (K16_NEED(3)) (DST)[tid + 3 * K16_THREADS] = p3;    \
    if constexpr (NV > 4) if (K16_NEED(4)) (DST)[tid + 4 * K16_THREADS] = p4;    \
    if constexpr (NV > 5) if (K16_NEED(5)) (DST)[tid + 5 * K16_THREADS] = p5;    \
    if constexpr (NV > 6) if (K16_NEED(6)) (DST)[tid + 6 * K16_THREADS] = p6;    \
    if constexpr (NV > 7) if (K16_NEED(7)) (DST)[tid + 7 * K16_THREADS] = p7;    \
    if (tail_ok) (DST)[tid + NV * K16_THREADS] = pt;                             \
  } while (0)
  {
    const float4* src = R4 + (size_t)tile_of(0) * TILE_V4;
    K16_LOAD(src);
  }
  K16_STORE(reinterpret_cast<float4*>(lds_tile[0]));
  __syncthreads();

  int s_cur = 0;
  int cur = 0;
  int par = 0;
  while (s_cur < n_scan) {
    // block-uniform bound: every wave reads the values published before the last barrier
    float bound = fmaxf(fmaxf(lds_wthr[par][0], lds_wthr[par][1]), fmaxf(lds_wthr[par][2], lds_wthr[par][3]));
    bound += prune_margin;
    const int s_next = next_live(s_cur + 1, bound);
    if (s_next < n_scan) {
      const float4* src = R4 + (size_t)tile_of(s_next) * TILE_V4;
      K16_LOAD(src);
    }
    const int t = tile_of(s_cur);

#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      // The accumulators start from |r|^2 of the lane's 16 references (rows 8m + 4h + (0..3)), the
      // MFMAs add -2 q.r, and |q|^2 is folded into the threshold (thrp = thr - |q|^2): the epilogue
      // needs no arithmetic before the vote.
      const float4* nr4 = reinterpret_cast<const float4*>(lds_tile[cur] + KB * 2 * 2 * K16_TS * 8) + sub * 8 + h;
      f32x16 acc0, acc1;
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) {
        const float4 v4 = nr4[2 * m4];
        acc0[4 * m4 + 0] = acc1[4 * m4 + 0] = v4.x;
        acc0[4 * m4 + 1] = acc1[4 * m4 + 1] = v4.y;
        acc0[4 * m4 + 2] = acc1[4 * m4 + 2] = v4.z;
        acc0[4 * m4 + 3] = acc1[4 * m4 + 3] = v4.w;
      }
      // tile layout [kb][h][plane][i][8 halves]: lane reads 16 B at ((kb*2+h)*2+plane)*TS + i
      const f16x8* a8 = reinterpret_cast<const f16x8*>(lds_tile[cur]) + sub * 32 + jq;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        // -2 x.y on the matrix pipe; the coordinate blocks run on the hi parts alone (NPROD == 1,
        // error <= 2^-9 |x~||y~|, see meld_knn16_error_coef) or with the full hi/lo split
        const bool full = (NPROD == 3);
        const f16x8 ahi = a8[((kb * 2 + h) * 2 + 0) * K16_TS];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[0][kb], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi[1][kb], acc1, 0, 0, 0);
        if (full) {
          const f16x8 alo = a8[((kb * 2 + h) * 2 + 1) * K16_TS];
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo[0][kb], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo[1][kb], acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi[0][kb], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi[1][kb], acc1, 0, 0, 0);
        }
      }

      const int ref_base = t * K16_TS + sub * 32 + 4 * h;
      // The vote below reads the accumulators from inline asm (v_min3_f32), which the hazard
      // recognizer does not look into: the MFMA -> VALU-read wait states (up to 19 for a 16-pass
      // XDL op) are inserted here by hand, tied to the accumulators so that no MFMA can be
      // scheduled after them.  (Without this the first registers of a chain were read stale.)
      asm volatile("s_nop 15\n\ts_nop 4" : "+v"(acc0), "+v"(acc1));
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x16 acc = g ? acc1 : acc0;
        if (ABL == 3) {  // profiling ablation: MFMAs only, accumulators kept live
          asm volatile("" ::"v"(acc[0]), "v"(acc[5]), "v"(acc[10]), "v"(acc[15]));
          continue;
        }
        float av[16];  // |r|^2 - 2 q.r  (= d2 - |q|^2)
#pragma unroll
        for (int r = 0; r < 16; ++r) av[r] = acc[r];
        const float m = min16(av);
        if (ABL == 1) {
          asm volatile("" ::"v"(m));  // profiling ablation: distances + minimum, selection removed
        } else if (__any(m < thrp[g])) {
          int* cntp = &lds_cnt[wave][g * 32 + jq];
          const size_t rowoff = (size_t)(row_base + g * 32 + jq) * cap;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = av[r];
            const int ref = ref_base + (r & 3) + 8 * (r >> 2);
            if (v < thrp[g] && ref < n_ref) {
              const int pos = atomicAdd(cntp, 1);
              if (pos < cap) {
                cand_d2[rowoff + pos] = v + nq[g];
                cand_idx[rowoff + pos] = ref;
              }
            }
          }
          const int c = __hip_atomic_load(cntp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          unsigned long long need = __ballot(h == 0 && c > cap - 32);
          if (need) {
            while (need) {
              const int j = __ffsll((long long)need) - 1;
              need &= need - 1;
              int* cj = &lds_cnt[wave][g * 32 + j];
              const int n = min(__hip_atomic_load(cj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), cap);
              const size_t ro = (size_t)(row_base + g * 32 + j) * cap;
              int n_new;
              float nt = knn16_squeeze_row(n, ksel, cap, cand_d2 + ro, cand_idx + ro, lane, &n_new);
              if (n_new > cap - 32) {
                // pathological ties at the threshold: rank the row down to exactly ksel entries
                knn16_rank_row(n_new, ksel, 1.0f, cand_d2 + ro, cand_idx + ro, lds_sd[wave], lds_si[wave], lane);
                n_new = min(n_new, ksel);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                nt = ld_sc1_f(cand_d2 + ro + n_new - 1);
              }
              if (lane == 0) *cj = n_new;
              if (jq == j) {
                thr[g] = (n >= ksel) ? nt : INFINITY;
                thrp[g] = thr[g] - nq[g];
              }
            }
            float w = fmaxf(thr[0], thr[1]);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) w = fmaxf(w, __shfl_xor(w, off, 64));
            wmax = w;
          }
        }
      }
    }

    if (s_next < n_scan) K16_STORE(reinterpret_cast<float4*>(lds_tile[cur ^ 1]));
    if (lane == 0) lds_wthr[par ^ 1][wave] = wmax;
    __syncthreads();
    s_cur = s_next;
    cur ^= 1;
    par ^= 1;
  }

  // final: sort every row, convert back to input units, publish its length
  const float out_scale = scale_info[1];  // 1 / s^2
  for (int j = 0; j < 64; ++j) {
    const int n = min(__hip_atomic_load(&lds_cnt[wave][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), cap);
    const int qr = row_base + j;
    const size_t ro = (size_t)qr * cap;
    knn16_rank_row(n, ksel, out_scale, cand_d2 + ro, cand_idx + ro, lds_sd[wave], lds_si[wave], lane);
    if (lane == 0) cand_cnt[qr] = min(n, ksel);
  }
#undef K16_LOAD
#undef K16_STORE
#undef K16_NEED
}

// Merge the per-slice candidate rows of one query (n_slices * q_pad rows of stride cap) into its
// final row: the ksel smallest of the union, sorted by (d2, idx).  One wave per query; at most
// K16_CAPMAX entries in the union.
constexpr int K16_MERGE_MAX = 1024;                 // entries in the union of the slice lists of one query
constexpr int K16_MERGE_SLOTS = K16_MERGE_MAX / 64;  // per lane

__global__ __launch_bounds__(64) void knn16_merge_slices_kernel(const int* __restrict__ s_idx,
                                                                const float* __restrict__ s_d2,
                                                                const int* __restrict__ s_cnt, int q_count, int q_pad,
                                                                int ksel, int cap, int n_slices,
                                                                int* __restrict__ out_idx, float* __restrict__ out_d2,
                                                                int* __restrict__ out_cnt) {
  __shared__ float sd[K16_MERGE_MAX];
  __shared__ int si[K16_MERGE_MAX];
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
// pruning bounds: bounding spheres of reference tiles / query workgroups in the scaled space
// ---------------------------------------------------------------------------------------------
// One wave per group of `gsize` (<= 256, multiple of 64 or the 64-ref tile) consecutive points
// [first + g*gsize, ...) clipped to [0, n_pts): centre = mean, radius = max distance to it.
// centres are written transposed ([k][group]) so that the table kernel reads them coalesced.
__global__ __launch_bounds__(64) void group_spheres_kernel(const double* __restrict__ X, int64_t n_pts, int d,
                                                           const double* __restrict__ mean,
                                                           const float* __restrict__ scale_info, int64_t first,
                                                           int gsize, int n_groups, float* __restrict__ centre_t,
                                                           float* __restrict__ radius) {
  const int g = blockIdx.x;
  const int lane = threadIdx.x;
  const float s = scale_info[0];
  const int64_t b = first + (int64_t)g * gsize;
  const int64_t e = min(b + gsize, n_pts);
  const int cnt = (int)max((int64_t)0, e - b);
  // pass 1: centre
  for (int k = 0; k < d; ++k) {
    float acc = 0.0f;
    for (int64_t i = b + lane; i < e; i += 64) acc += s * (float)(X[i * d + k] - mean[k]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) centre_t[(size_t)k * n_groups + g] = cnt ? acc / (float)cnt : 0.0f;
  }
  __syncthreads();
  // pass 2: radius
  float r2 = 0.0f;
  for (int64_t i = b + lane; i < e; i += 64) {
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) {
      const float t = s * (float)(X[i * d + k] - mean[k]) - centre_t[(size_t)k * n_groups + g];
      acc = fmaf(t, t, acc);
    }
    r2 = fmaxf(r2, acc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, off, 64));
  if (lane == 0) radius[g] = cnt ? sqrtf(r2) * 1.0001f + 1e-30f : -1.0f;  // -1: empty group
}

// lb2[qb][t] = max(0, |cq - ct| (1 - eps) - rq - rt)^2 ; empty tiles get +inf (always pruned)
__global__ __launch_bounds__(256) void bounds_table_kernel(const float* __restrict__ cq_t, const float* __restrict__ rq,
                                                           int n_qb, const float* __restrict__ ct_t,
                                                           const float* __restrict__ rt, int n_tiles, int d,
                                                           float* __restrict__ lb2) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int qb = blockIdx.y;
  if (t >= n_tiles) return;
  float acc = 0.0f;
  for (int k = 0; k < d; ++k) {
    const float df = cq_t[(size_t)k * n_qb + qb] - ct_t[(size_t)k * n_tiles + t];
    acc = fmaf(df, df, acc);
  }
  float out;
  if (rt[t] < 0.0f) {
    out = INFINITY;
  } else if (rq[qb] < 0.0f) {
    out = 0.0f;
  } else {
    const float lb = sqrtf(acc) * 0.9999f - rq[qb] - rt[t];
    out = lb > 0.0f ? lb * lb : 0.0f;
  }
  lb2[(size_t)qb * n_tiles + t] = out;
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

// scale_info: [0] = s (multiply centred data by s), [1] = 1/s^2, [2] = absmax (input of this kernel)
__global__ void finish_scale_kernel(float* scale_info) {
  const float a = scale_info[2];
  const float s = (a > 0.0f) ? 1.0f / a : 1.0f;
  scale_info[0] = s;
  scale_info[1] = 1.0f / (s * s);
}

__device__ __forceinline__ void split_store(_Float16* dst_hi, _Float16* dst_lo, float v) {
  const _Float16 hi = (_Float16)v;
  *dst_hi = hi;
  *dst_lo = (_Float16)(v - (float)hi);
}

__device__ __forceinline__ float scaled_norm2(const double* xrow, const double* mean, float s, int d) {
  float n = 0.0f;
  for (int k = 0; k < d; ++k) {
    const float v = s * (float)(xrow[k] - mean[k]);
    n = fmaf(v, v, n);
  }
  return n;
}

__global__ __launch_bounds__(256) void prepare_refs16_kernel(const double* __restrict__ X, int64_t N, int d,
                                                             const double* __restrict__ mean,
                                                             const float* __restrict__ scale_info, int KB,
                                                             int64_t n_pad, _Float16* __restrict__ Rt16,
                                                             float* __restrict__ norm2, float* __restrict__ norm2_max) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float s = scale_info[0];
  float n_orig = 0.0f;
  if (i < n_pad) {
    const int64_t t = i / K16_TS;
    const int ii = (int)(i % K16_TS);
    const size_t tile_h = (size_t)KB * 2 * 2 * K16_TS * 8 + 2 * K16_TS;
    _Float16* tile = Rt16 + (size_t)t * tile_h;
    const bool real = i < N;
    const double* xrow = X + (real ? i : 0) * d;
    const float n = real ? scaled_norm2(xrow, mean, s, d) : K16_BIG;  // padding rows are infinitely far
    if (real) {
      n_orig = n * scale_info[1];
      norm2[i] = n_orig;
    }
    reinterpret_cast<float*>(tile + (size_t)KB * 2 * 2 * K16_TS * 8)[ii] = n;
    for (int c = 0; c < KB * 16; ++c) {
      const float v = (real && c < d) ? -2.0f * (s * (float)(xrow[c] - mean[c])) : 0.0f;
      const int kb = c >> 4, hh = (c >> 3) & 1, e = c & 7;
      _Float16* base = tile + ((size_t)((kb * 2 + hh) * 2) * K16_TS + ii) * 8 + e;
      split_store(base, base + (size_t)K16_TS * 8, v);
    }
  }
  float m = n_orig;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(norm2_max), __float_as_int(m));
}

__global__ __launch_bounds__(256) void prepare_queries16_kernel(const double* __restrict__ X, int d,
                                                                const double* __restrict__ mean,
                                                                const float* __restrict__ scale_info, int KB,
                                                                int64_t q_begin, int64_t q_count, int64_t q_pad,
                                                                const int* __restrict__ rows,
                                                                _Float16* __restrict__ Q16, float* __restrict__ Qn) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= q_pad) return;
  const float s = scale_info[0];
  const int64_t qq = q < q_count ? q : q_count - 1;
  const int64_t src = q_begin + (rows ? (int64_t)rows[qq] : qq);
  const double* xrow = X + src * d;
  Qn[q] = scaled_norm2(xrow, mean, s, d);
  _Float16* row = Q16 + (size_t)q * (KB * 32);
  for (int c = 0; c < KB * 16; ++c) {
    const float v = c < d ? s * (float)(xrow[c] - mean[c]) : 0.0f;
    const int kb = c >> 4, hh = (c >> 3) & 1, e = c & 7;
    _Float16* base = row + ((kb * 2 + hh) * 2) * 8 + e;
    split_store(base, base + 8, v);
  }
}

}  // namespace meld

using namespace meld;

extern "C" int meld_knn16_kblocks(int d) {
  if (d < 1) return MELD_ERR_INVALID;
  const int kb = (d + 15) / 16;
  if (kb > 8) {
    set_err("meld_knn16_kblocks: d=%d exceeds the largest instantiated distance kernel (d <= 128)", d);
    return MELD_ERR_UNSUPPORTED;
  }
  return kb;
}
// bytes of one reference tile / one query row of the fp16 operand arrays
extern "C" size_t meld_knn16_tile_bytes(int d) {
  const int kb = meld_knn16_kblocks(d);
  return kb < 0 ? 0 : (size_t)kb * 2 * 2 * K16_TS * 16 + sizeof(float) * K16_TS;
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
// Bound on |d2_approx - d2_exact| / max_i |x~_i|^2 (n_max) that meld_knn_refine budgets for.
// d2 = sum_c q_c r_c with sum |q_c r_c| <= |x~_q|^2 + |x~_r|^2 + 2 |x~_q||x~_r| <= 4 n_max.
//   fp32 accumulation of <= 64 products per chain: gamma_64 * 4 n_max = 2^-16 n_max;
//   nprod = 3: every operand is hi + lo (|v - hi - lo| <= 2^-22 |v|), dropped lo.lo terms
//              <= 3 * 2^-22 * 4 n_max: total < 2^-15 n_max            (measured: 7.6e-7 n_max)
//   nprod = 1: coordinate blocks on the fp16 hi parts only:
//              |q.r - qhi.rhi| <= |qlo.r| + |qhi.rlo| <= 2 * 2^-11 |x~_q| |2 x~_r| <= 2^-9 n_max
//              (Cauchy-Schwarz); the norms are added in fp32 in the epilogue   (measured: 5.4e-4 n_max)
extern "C" double meld_knn16_error_coef(int nprod) {
  const double full = 3.0517578125e-05;  // 2^-15
  return nprod == 1 ? (0.001953125 + full) : full;
}
// The same bound split for a per-row allowance  E_i = c_const max|x~|^2 + c_lin |x~_i| max|x~|
// (the nprod = 1 term is |qlo.r| + |qhi.rlo| <= 2^-9 |x~_q| |x~_r|): rows near the centre of
// the data get a tighter allowance than the global worst case.
extern "C" double meld_knn16_error_coef_const(int nprod) {
  (void)nprod;
  return 3.0517578125e-05;
}
extern "C" double meld_knn16_error_coef_lin(int nprod) { return nprod == 1 ? 0.001953125 : 0.0; }

extern "C" int meld_knn16_prepare(const double* X, int64_t N, int d, const double* mean, int64_t q_begin,
                                  int64_t q_count, void* Rt16, void* Q16, float* Qn, float* norm2, float* norm2_max,
                                  float* scale_info, meld_stream_t stream) {
  MELD_CHECK_ARG(X && mean && Rt16 && Q16 && Qn && norm2 && norm2_max && scale_info && N > 0,
                 "meld_knn16_prepare: null/empty argument");
  MELD_CHECK_ARG(q_count > 0 && q_begin >= 0 && q_begin + q_count <= N, "meld_knn16_prepare: bad query range");
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  hipStream_t st = S(stream);
  MELD_HIP_CALL(hipMemsetAsync(scale_info, 0, 4 * sizeof(float), st));
  MELD_HIP_CALL(hipMemsetAsync(norm2_max, 0, sizeof(float), st));
  hipLaunchKernelGGL(absmax_centered_kernel, dim3(2048), dim3(256), 0, st, X, N * (int64_t)d, d, mean, scale_info);
  hipLaunchKernelGGL(finish_scale_kernel, dim3(1), dim3(1), 0, st, scale_info);
  const int64_t n_pad = ceil_div(N, K16_TS) * K16_TS;
  hipLaunchKernelGGL(prepare_refs16_kernel, dim3((unsigned)ceil_div(n_pad, 256)), dim3(256), 0, st, X, N, d, mean,
                     scale_info, KB, n_pad, reinterpret_cast<_Float16*>(Rt16), norm2, norm2_max);
  const int64_t q_pad = ceil_div(q_count, K16_BQ) * K16_BQ;
  hipLaunchKernelGGL(prepare_queries16_kernel, dim3((unsigned)ceil_div(q_pad, 256)), dim3(256), 0, st, X, d, mean,
                     scale_info, KB, q_begin, q_count, q_pad, (const int*)nullptr, reinterpret_cast<_Float16*>(Q16), Qn);
  MELD_LAUNCH_CHECK("meld_knn16_prepare");
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
  hipLaunchKernelGGL(prepare_queries16_kernel, dim3((unsigned)ceil_div(q_pad, 256)), dim3(256), 0, S(stream), X, d, mean,
                     scale_info, KB, q_begin, n_rows, q_pad, rows, reinterpret_cast<_Float16*>(Q16), Qn);
  MELD_LAUNCH_CHECK("meld_knn16_prepare_rows");
  return MELD_OK;
}

extern "C" size_t meld_knn16_bounds_bytes(int64_t n_ref, int64_t q_count) {
  return (size_t)ceil_div(q_count, K16_BQ) * (size_t)ceil_div(n_ref, K16_TS) * sizeof(float);
}
extern "C" size_t meld_knn16_bounds_temp_bytes(int64_t n_ref, int d, int64_t q_count) {
  const size_t n_t = (size_t)ceil_div(n_ref, K16_TS), n_q = (size_t)ceil_div(q_count, K16_BQ);
  return sizeof(float) * ((n_t + n_q) * (size_t)(d + 1)) + 256;
}

extern "C" int meld_knn16_bounds(const double* X, int64_t N, int d, const double* mean, const float* scale_info,
                                 int64_t q_begin, int64_t q_count, void* temp, float* lb2, meld_stream_t stream) {
  MELD_CHECK_ARG(X && mean && scale_info && temp && lb2 && N > 0 && q_count > 0 && q_begin >= 0 && q_begin + q_count <= N,
                 "meld_knn16_bounds: bad arguments");
  const int n_t = (int)ceil_div(N, K16_TS), n_q = (int)ceil_div(q_count, K16_BQ);
  float* ct = reinterpret_cast<float*>(temp);
  float* rt = ct + (size_t)n_t * d;
  float* cq = rt + n_t;
  float* rq = cq + (size_t)n_q * d;
  hipStream_t st = S(stream);
  hipLaunchKernelGGL(group_spheres_kernel, dim3(n_t), dim3(64), 0, st, X, N, d, mean, scale_info, (int64_t)0, K16_TS, n_t,
                     ct, rt);
  hipLaunchKernelGGL(group_spheres_kernel, dim3(n_q), dim3(64), 0, st, X, q_begin + q_count, d, mean, scale_info, q_begin,
                     K16_BQ, n_q, cq, rq);
  hipLaunchKernelGGL(bounds_table_kernel, dim3((unsigned)ceil_div(n_t, 256), n_q), dim3(256), 0, st, cq, rq, n_q, ct, rt,
                     n_t, d, lb2);
  MELD_LAUNCH_CHECK("meld_knn16_bounds");
  return MELD_OK;
}

extern "C" int meld_knn16_topk(const void* Q16, const float* Qn, const void* Rt16, const float* scale_info,
                               int64_t n_ref, int d,
                               int64_t q_count, int ksel, int nprod, int n_slices, const float* lb2,
                               const float* norm2_max, int64_t q_begin, int32_t* cand_idx, float* cand_d2,
                               int32_t* cand_cnt, meld_stream_t stream) {
  MELD_CHECK_ARG(nprod == 1 || nprod == 3, "meld_knn16_topk: nprod must be 1 or 3");
  MELD_CHECK_ARG(n_slices >= 1 && n_slices * ksel <= K16_MERGE_MAX && (n_slices == 1 || lb2 == nullptr),
                 "meld_knn16_topk: n_slices=%d must satisfy n_slices * ksel <= %d and excludes pruning", n_slices,
                 K16_MERGE_MAX);
  MELD_CHECK_ARG(Q16 && Qn && Rt16 && scale_info && cand_idx && cand_d2 && cand_cnt, "meld_knn16_topk: null pointer");
  MELD_CHECK_ARG(lb2 == nullptr || norm2_max != nullptr, "meld_knn16_topk: pruning needs norm2_max");
  MELD_CHECK_ARG(n_ref > 0 && n_ref < (int64_t)1 << 31 && q_count > 0, "meld_knn16_topk: bad sizes");
  const int cap = meld_knn16_row_capacity(ksel);
  if (cap < 0) return cap;
  const int KB = meld_knn16_kblocks(d);
  if (KB < 0) return KB;
  const int n_tiles = (int)ceil_div(n_ref, K16_TS);
  const dim3 grid((unsigned)ceil_div(q_count, K16_BQ), (unsigned)n_slices);
  MELD_CHECK_ARG(n_slices <= ceil_div(n_ref, K16_TS), "meld_knn16_topk: more slices than reference tiles");
  const int tile_origin = (int)((q_begin / K16_TS) % n_tiles);  // the scan starts at the queries' own position
  // profiling hooks (never set in production): MELD_KNN16_ABLATION=1 distances without selection,
  // =3 MFMAs only; MELD_KNN16_PADLDS=<bytes> extra dynamic LDS to lower the workgroups per CU
  const char* abl_env = getenv("MELD_KNN16_ABLATION");
  const int abl = abl_env ? atoi(abl_env) : 0;
  const char* pad_env = getenv("MELD_KNN16_PADLDS");
  const size_t pad_lds = pad_env ? (size_t)atoi(pad_env) : 0;
  const _Float16* q = reinterpret_cast<const _Float16*>(Q16);
  const _Float16* r = reinterpret_cast<const _Float16*>(Rt16);
#define K16_LAUNCH2(KBV, ABLV, NP)                                                                             \
  hipLaunchKernelGGL((knn16_topk_kernel<KBV, ABLV, NP>), grid, dim3(K16_THREADS), pad_lds, S(stream), q, Qn, r,  \
                     scale_info, (int)n_ref, n_tiles, ksel, cap, lb2, norm2_max, (float)meld_knn16_error_coef(nprod), \
                     tile_origin, cand_idx, cand_d2, cand_cnt)
#define K16_LAUNCH(KBV, ABLV)        \
  do {                               \
    if (nprod == 1)                  \
      K16_LAUNCH2(KBV, ABLV, 1);     \
    else                             \
      K16_LAUNCH2(KBV, ABLV, 3);     \
  } while (0)
#define K16_CASE(KBV)                  \
  case KBV:                            \
    if (abl == 1)                      \
      K16_LAUNCH(KBV, 1);              \
    else if (abl == 3)                 \
      K16_LAUNCH(KBV, 3);              \
    else                               \
      K16_LAUNCH(KBV, 0);              \
    break;
  switch (KB) {
    K16_CASE(1)
    K16_CASE(2)
    K16_CASE(3)
    K16_CASE(4)
    K16_CASE(5)
    K16_CASE(6)
    K16_CASE(7)
    K16_CASE(8)
    default:
      set_err("meld_knn16_topk: KB=%d is not an instantiated size", KB);
      return MELD_ERR_UNSUPPORTED;
  }
#undef K16_CASE
#undef K16_LAUNCH
#undef K16_LAUNCH2
  MELD_LAUNCH_CHECK("knn16_topk_kernel");
  return MELD_OK;
}

extern "C" int meld_knn16_merge_slices(const int32_t* s_idx, const float* s_d2, const int32_t* s_cnt, int64_t q_count,
                                       int ksel, int n_slices, int32_t* out_idx, float* out_d2, int32_t* out_cnt,
                                       meld_stream_t stream) {
  MELD_CHECK_ARG(s_idx && s_d2 && s_cnt && out_idx && out_d2 && out_cnt && q_count > 0, "meld_knn16_merge_slices: null");
  const int cap = meld_knn16_row_capacity(ksel);
  if (cap < 0) return cap;
  MELD_CHECK_ARG(n_slices >= 1 && n_slices * ksel <= K16_MERGE_MAX, "meld_knn16_merge_slices: too many slices");
  const int q_pad = (int)(ceil_div(q_count, K16_BQ) * K16_BQ);
  hipLaunchKernelGGL(knn16_merge_slices_kernel, dim3((unsigned)q_count), dim3(64), 0, S(stream), s_idx, s_d2,
                     s_cnt, (int)q_count, q_pad, ksel, cap, n_slices, out_idx, out_d2, out_cnt);
  MELD_LAUNCH_CHECK("knn16_merge_slices_kernel");
  return MELD_OK;
}
