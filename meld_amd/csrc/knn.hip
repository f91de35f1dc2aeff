// knn.hip -- brute-force nearest-neighbour candidate search on the gfx950 matrix cores.
//
// Replaces the "Calculating KNN search" stage of graphtools
// ([UPSTREAM kNNGraph.build_kernel_to_data -> sklearn NearestNeighbors.kneighbors], reached from
// reference meld/meld.py:273).  It only *selects candidates*: every candidate is re-evaluated in
// exact fp64 by refine.hip before it is used, and refine.hip proves per row that no reference was
// missed (or sends the row to the exact fp64 fallback).
//
// Formulation.  With centred points x~ and n_i = |x~_i|^2 the squared distance is one dot
// product of two augmented vectors
//     q_j = [ x~_j , n_j , 1 , 0.. ]      r_i = [ -2 x~_i , 1 , n_i , 0.. ]      d2_ij = <q_j, r_i>
// so the whole N x N distance matrix is a GEMM with inner dimension KP = d + 2 (padded), and the
// norms ride in what would otherwise be K-padding.  It runs on v_mfma_f32_32x32x2_f32 (exact fp32
// FMA chain, 157 TF peak).
//
// Work decomposition (64-wide waves):
//   workgroup = 4 waves = 128 queries;  grid = ceil(Nq / 128)  (>> 256 CUs)
//   wave      = 32 queries, held as MFMA B-fragments in registers for the whole kernel
//   all 4 waves share the stream of reference tiles (64 refs = 2 MFMA sub-tiles), double-buffered
//   in LDS in a tile-major layout [KP/2][64][2] that makes the A-fragment reads conflict-free
//   ds_read_b64 and the global->LDS copy a flat, fully coalesced 16-byte copy.
//   MFMA output layout: lane l holds query (l & 31) and 16 different references, so the
//   per-query selection threshold is ONE register per lane.
//
// Selection: a candidate survives if d2 < thr(query), thr = current ksel-th smallest.  Survivors
// are appended to the query's row of the output buffer (capacity CAP = ksel + 64, counter in
// LDS); when a row has fewer than 32 free slots the wave compacts it (rank by (d2, idx), keep the
// ksel smallest, tighten thr).  Expected survivors per query ~ ksel * ln(N / ksel), so the
// steady-state cost per candidate is one v_min + the shared v_cmp/ballot.
#include "common.hpp"

#include <algorithm>

namespace meld {

constexpr int KNN_TS = 64;        // references per LDS tile
constexpr int KNN_BQ = 128;       // queries per workgroup
constexpr int KNN_THREADS = 256;  // 4 waves
constexpr int KNN_SLACK = 64;     // CAP = ksel + KNN_SLACK
constexpr int KNN_CAPMAX = 192;   // ksel <= 128
constexpr float KNN_BIG = 1.0e30f;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float ld_l2_f(const float* p) {
  // L1-bypassing load (sc1): the row was last written by this same wave through L2
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_l2_i(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave-cooperative compaction of one candidate row: keep the `ksel` smallest of its n entries,
// written back sorted by (d2, idx).  Returns the new threshold (ksel-th smallest d2, or +inf while
// the row holds fewer than ksel entries).  All 64 lanes must call it with wave-uniform arguments.
__device__ float knn_compact_row(int n, int ksel, float* __restrict__ d2row, int* __restrict__ idxrow,
                                 float* sd, int* si, int lane) {
  // every append store of this wave must have reached L2 before we read the row back
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float d[3];
  int ix[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int p = lane + 64 * e;
    if (p < n) {
      d[e] = ld_l2_f(d2row + p);
      ix[e] = ld_l2_i(idxrow + p);
    } else {
      d[e] = INFINITY;
      ix[e] = 0x7fffffff;
    }
    sd[p] = d[e];
    si[p] = ix[e];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int rk[3] = {0, 0, 0};
  for (int e = 0; e < n; ++e) {
    const float de = sd[e];
    const int ie = si[e];
#pragma unroll
    for (int q = 0; q < 3; ++q) rk[q] += (de < d[q] || (de == d[q] && ie < ix[q])) ? 1 : 0;
  }
  float newthr = INFINITY;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int p = lane + 64 * q;
    const bool valid = p < n;
    if (valid && rk[q] < ksel) {
      d2row[rk[q]] = d[q];
      idxrow[rk[q]] = ix[q];
    }
    const unsigned long long b = __ballot(valid && rk[q] == ksel - 1);
    if (b) newthr = __shfl(d[q], __ffsll((long long)b) - 1, 64);
  }
  // the LDS scratch is reused by the next compaction of this wave
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  return newthr;
}

template <int KP>
__global__ __launch_bounds__(KNN_THREADS, (KP <= 64 ? 2 : 1)) void knn_topk_kernel(
    const float* __restrict__ Q, const float* __restrict__ Rt, int n_ref, int n_tiles, int ksel, int cap,
    int* __restrict__ cand_idx, float* __restrict__ cand_d2, int* __restrict__ cand_cnt) {
  constexpr int NP = KP / 4;          // float2 pairs per lane half
  constexpr int TILE_F = KP * KNN_TS; // floats per reference tile
  constexpr int TILE_V4 = TILE_F / 4;

  __shared__ __attribute__((aligned(16))) float lds_tile[2][TILE_F];
  __shared__ int lds_cnt[4][32];
  __shared__ float lds_sd[4][KNN_CAPMAX];
  __shared__ int lds_si[4][KNN_CAPMAX];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int jq = lane & 31;  // query column of this lane == reference row it feeds to the MFMA
  const int h = lane >> 5;   // which half of the K range this lane feeds
  const int q_row = blockIdx.x * KNN_BQ + wave * 32 + jq;
  const size_t rowoff = (size_t)q_row * cap;

  // query fragments: pairs [h*NP, (h+1)*NP) of the augmented query row
  float2 bq[NP];
  {
    const float2* qrow = reinterpret_cast<const float2*>(Q + (size_t)q_row * KP) + h * NP;
#pragma unroll
    for (int u = 0; u < NP; ++u) bq[u] = qrow[u];
  }

  if (lane < 32) lds_cnt[wave][lane] = 0;
  float thr = INFINITY;

  const float4* Rt4 = reinterpret_cast<const float4*>(Rt);
  // tile staging registers: up to 8 unconditional float4 per thread + one guarded tail, kept as
  // named scalars (an indexed array here ends up in scratch memory)
  constexpr int NFULL = TILE_V4 / KNN_THREADS;
  static_assert(NFULL <= 8, "tile too large for the staging registers");
  constexpr bool HAS_TAIL = (TILE_V4 % KNN_THREADS) != 0;
  const bool tail_ok = HAS_TAIL && (NFULL * KNN_THREADS + tid < TILE_V4);
  float4 p0, p1, p2, p3, p4, p5, p6, p7, pt;
  p0 = p1 = p2 = p3 = p4 = p5 = p6 = p7 = pt = make_float4(0.f, 0.f, 0.f, 0.f);
#define MELD_TILE_LOAD(SRC)                                        \
  do {                                                             \
    if constexpr (NFULL > 0) p0 = (SRC)[tid + 0 * KNN_THREADS];    \
    if constexpr (NFULL > 1) p1 = (SRC)[tid + 1 * KNN_THREADS];    \
    if constexpr (NFULL > 2) p2 = (SRC)[tid + 2 * KNN_THREADS];    \
    if constexpr (NFULL > 3) p3 = (SRC)[tid + 3 * KNN_THREADS];    \
    if constexpr (NFULL > 4) p4 = (SRC)[tid + 4 * KNN_THREADS];    \
    if constexpr (NFULL > 5) p5 = (SRC)[tid + 5 * KNN_THREADS];    \
    if constexpr (NFULL > 6) p6 = (SRC)[tid + 6 * KNN_THREADS];    \
    if constexpr (NFULL > 7) p7 = (SRC)[tid + 7 * KNN_THREADS];    \
    if (tail_ok) pt = (SRC)[tid + NFULL * KNN_THREADS];            \
  } while (0)
#define MELD_TILE_STORE(DST)                                       \
  do {                                                             \
    if constexpr (NFULL > 0) (DST)[tid + 0 * KNN_THREADS] = p0;    \
    if constexpr (NFULL > 1) (DST)[tid + 1 * KNN_THREADS] = p1;    \
    if constexpr (NFULL > 2) (DST)[tid + 2 * KNN_THREADS] = p2;    \
    if constexpr (NFULL > 3) (DST)[tid + 3 * KNN_THREADS] = p3;    \
    if constexpr (NFULL > 4) (DST)[tid + 4 * KNN_THREADS] = p4;    \
    if constexpr (NFULL > 5) (DST)[tid + 5 * KNN_THREADS] = p5;    \
    if constexpr (NFULL > 6) (DST)[tid + 6 * KNN_THREADS] = p6;    \
    if constexpr (NFULL > 7) (DST)[tid + 7 * KNN_THREADS] = p7;    \
    if (tail_ok) (DST)[tid + NFULL * KNN_THREADS] = pt;            \
  } while (0)
  MELD_TILE_LOAD(Rt4);
  MELD_TILE_STORE(reinterpret_cast<float4*>(lds_tile[0]));
  __syncthreads();

  int* cntp = &lds_cnt[wave][jq];

  for (int t = 0; t < n_tiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < n_tiles) {
      const float4* src = Rt4 + (size_t)(t + 1) * TILE_V4;
      MELD_TILE_LOAD(src);
    }

#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      const float2* a2 = reinterpret_cast<const float2*>(lds_tile[cur]) + (h * NP) * KNN_TS + sub * 32 + jq;
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const float2 a = a2[u * KNN_TS];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[u].y, acc, 0, 0, 0);
      }

      // fast path: nothing in this 32x32 block beats any query's threshold
      float m01 = fminf(fminf(acc[0], acc[1]), fminf(acc[2], acc[3]));
      float m23 = fminf(fminf(acc[4], acc[5]), fminf(acc[6], acc[7]));
      float m45 = fminf(fminf(acc[8], acc[9]), fminf(acc[10], acc[11]));
      float m67 = fminf(fminf(acc[12], acc[13]), fminf(acc[14], acc[15]));
      const float m = fminf(fminf(m01, m23), fminf(m45, m67));
      if (__any(m < thr)) {
        const int ref_base = t * KNN_TS + sub * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[r];
          const int ref = ref_base + (r & 3) + 8 * (r >> 2);
          if (v < thr && ref < n_ref) {
            const int pos = atomicAdd(cntp, 1);
            if (pos < cap) {
              cand_d2[rowoff + pos] = v;
              cand_idx[rowoff + pos] = ref;
            }
          }
        }
        const int c = __hip_atomic_load(cntp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned long long need = __ballot(h == 0 && c > cap - 32);
        while (need) {
          const int j = __ffsll((long long)need) - 1;
          need &= need - 1;
          const int n = min(__hip_atomic_load(&lds_cnt[wave][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), cap);
          const size_t ro = (size_t)(blockIdx.x * KNN_BQ + wave * 32 + j) * cap;
          const float nt = knn_compact_row(n, ksel, cand_d2 + ro, cand_idx + ro, lds_sd[wave], lds_si[wave], lane);
          if (lane == 0) lds_cnt[wave][j] = min(n, ksel);
          if (jq == j) thr = nt;
        }
      }
    }

    if (t + 1 < n_tiles) {
      MELD_TILE_STORE(reinterpret_cast<float4*>(lds_tile[cur ^ 1]));
    }
    __syncthreads();
  }

  // final: sort every row and publish its length
  for (int j = 0; j < 32; ++j) {
    const int n = min(__hip_atomic_load(&lds_cnt[wave][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), cap);
    const int qr = blockIdx.x * KNN_BQ + wave * 32 + j;
    const size_t ro = (size_t)qr * cap;
    (void)knn_compact_row(n, ksel, cand_d2 + ro, cand_idx + ro, lds_sd[wave], lds_si[wave], lane);
    if (lane == 0) cand_cnt[qr] = min(n, ksel);
  }
}

// ---------------------------------------------------------------------------------------------
// operand preparation
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void col_sums_kernel(const double* __restrict__ X, int64_t N, int d,
                                                       double* __restrict__ sums) {
  // thread t owns column t % d for rows (t / d) + k * rows_per_pass
  __shared__ double part[256];
  const int tid = threadIdx.x;
  const int rpp = 256 / d;  // rows per pass (d <= 256)
  const int c = tid % d;
  const int rl = tid / d;
  double s = 0.0;
  if (rl < rpp) {
    for (int64_t row = (int64_t)blockIdx.x * rpp + rl; row < N; row += (int64_t)gridDim.x * rpp)
      s += X[row * d + c];
  }
  part[tid] = (rl < rpp) ? s : 0.0;
  __syncthreads();
  if (tid < d) {
    double tot = 0.0;
    for (int k = 0; k < rpp; ++k) tot += part[k * d + tid];
    atomicAdd(&sums[tid], tot);
  }
}

__device__ __forceinline__ float centred_norm2(const double* __restrict__ xrow, const double* __restrict__ mean, int d) {
  float n = 0.0f;
  for (int k = 0; k < d; ++k) {
    const float v = (float)(xrow[k] - mean[k]);
    n = fmaf(v, v, n);
  }
  return n;
}

__global__ __launch_bounds__(256) void prepare_refs_kernel(const double* __restrict__ X, int64_t N, int d,
                                                           const double* __restrict__ mean, int KP,
                                                           int64_t n_pad, float* __restrict__ Rt,
                                                           float* __restrict__ norm2,
                                                           float* __restrict__ norm2_max) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float n = 0.0f;
  if (i < n_pad) {
    const int64_t t = i / KNN_TS;
    const int ii = (int)(i % KNN_TS);
    float2* dst = reinterpret_cast<float2*>(Rt) + (size_t)t * (KP / 2) * KNN_TS + ii;
    if (i < N) {
      const double* xrow = X + i * d;
      n = centred_norm2(xrow, mean, d);
      norm2[i] = n;
      for (int kp = 0; kp < KP / 2; ++kp) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = 2 * kp + e;
          v[e] = (c < d) ? -2.0f * (float)(xrow[c] - mean[c]) : (c == d ? 1.0f : (c == d + 1 ? n : 0.0f));
        }
        dst[(size_t)kp * KNN_TS] = make_float2(v[0], v[1]);
      }
    } else {
      for (int kp = 0; kp < KP / 2; ++kp) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = 2 * kp + e;
          v[e] = (c == d + 1) ? KNN_BIG : 0.0f;
        }
        dst[(size_t)kp * KNN_TS] = make_float2(v[0], v[1]);
      }
    }
  }
  // one atomicMax per wave (norms are >= 0, so the int ordering of the bits is the float ordering)
  float m = n;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(norm2_max), __float_as_int(m));
}

__global__ __launch_bounds__(256) void prepare_queries_kernel(const double* __restrict__ X, int d,
                                                              const double* __restrict__ mean, int KP,
                                                              int64_t q_begin, int64_t q_count, int64_t q_pad,
                                                              float* __restrict__ Q) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= q_pad) return;
  const int64_t src = q_begin + (q < q_count ? q : q_count - 1);
  const double* xrow = X + src * d;
  const float n = centred_norm2(xrow, mean, d);
  float* dst = Q + (size_t)q * KP;
  for (int c = 0; c < KP; ++c)
    dst[c] = (c < d) ? (float)(xrow[c] - mean[c]) : (c == d ? n : (c == d + 1 ? 1.0f : 0.0f));
}

static const int kSupportedKP[] = {8, 12, 16, 24, 32, 40, 52, 64, 80, 104, 128};

}  // namespace meld

using namespace meld;

extern "C" int meld_knn_padded_dim(int d) {
  if (d < 1) return MELD_ERR_INVALID;
  for (int kp : kSupportedKP)
    if (d + 2 <= kp) return kp;
  set_err("meld_knn_padded_dim: d=%d exceeds the largest instantiated distance kernel (d <= 126)", d);
  return MELD_ERR_UNSUPPORTED;
}
extern "C" double meld_knn_error_coef(int d) {
  const int kp = meld_knn_padded_dim(d);
  return kp < 0 ? -1.0 : (double)kp * 4.76837158203125e-07; /* KP * 2^-21 */
}
extern "C" int meld_knn_tile_refs(void) { return KNN_TS; }
extern "C" int meld_knn_block_queries(void) { return KNN_BQ; }
extern "C" int meld_knn_row_capacity(int ksel) {
  if (ksel < 1 || ksel > KNN_CAPMAX - KNN_SLACK) {
    set_err("meld_knn_row_capacity: ksel=%d outside [1, %d]", ksel, KNN_CAPMAX - KNN_SLACK);
    return MELD_ERR_UNSUPPORTED;
  }
  return ksel + KNN_SLACK;
}

// column sums, minima and maxima in ONE pass over X (the centring mean, the NaN / infinity check of the front end -- a
// non-finite value makes its column sum non-finite -- and the scale of the search operands: max_i |x_ic - mean_c| is attained
// at the column's minimum or maximum).  Per-workgroup partials in `part` [3][grid][d], reduced in a fixed order: the same
// bits on every run (col_sums_kernel's atomics add in arrival order).
namespace meld {
constexpr int COL_STATS_GRID = 1024;
__global__ __launch_bounds__(256) void col_stats_kernel(const double* __restrict__ X, int64_t N, int d, double* __restrict__ part) {
  __shared__ double ps[256], pmin[256], pmax[256];
  const int tid = threadIdx.x;
  const int rpp = 256 / d;  // rows per pass (d <= 256)
  const int c = tid % d;
  const int rl = tid / d;
  double s = 0.0, mn = INFINITY, mx = -INFINITY;
  if (rl < rpp) {
    for (int64_t row = (int64_t)blockIdx.x * rpp + rl; row < N; row += (int64_t)gridDim.x * rpp) {
      const double v = X[row * d + c];
      s += v;
      mn = fmin(mn, v);
      mx = fmax(mx, v);
    }
  }
  ps[tid] = s;
  pmin[tid] = mn;
  pmax[tid] = mx;
  __syncthreads();
  if (tid < d) {
    double ts = 0.0, tmn = INFINITY, tmx = -INFINITY;
    for (int k = 0; k < rpp; ++k) {
      ts += ps[k * d + tid];
      tmn = fmin(tmn, pmin[k * d + tid]);
      tmx = fmax(tmx, pmax[k * d + tid]);
    }
    const size_t g = gridDim.x;
    part[(0 * g + blockIdx.x) * d + tid] = ts;
    part[(1 * g + blockIdx.x) * d + tid] = tmn;
    part[(2 * g + blockIdx.x) * d + tid] = tmx;
  }
}
// one workgroup per column: 256 threads fold the per-workgroup partials (thread t: partials t, t + 256, ...), then a fixed tree
// (a single workgroup walking all 3 x 1024 x d partials took 0.3 ms: more than the pass over X it finishes)
__global__ __launch_bounds__(256) void col_stats_finish_kernel(const double* __restrict__ part, int grid, int d, double* __restrict__ sums,
                                                               double* __restrict__ mins, double* __restrict__ maxs) {
  __shared__ double ps[256], pmin[256], pmax[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double ts = 0.0, tmn = INFINITY, tmx = -INFINITY;
  for (int b = tid; b < grid; b += 256) {
    ts += part[((size_t)0 * grid + b) * d + c];
    tmn = fmin(tmn, part[((size_t)1 * grid + b) * d + c]);
    tmx = fmax(tmx, part[((size_t)2 * grid + b) * d + c]);
  }
  ps[tid] = ts;
  pmin[tid] = tmn;
  pmax[tid] = tmx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) {
      ps[tid] += ps[tid + w];
      pmin[tid] = fmin(pmin[tid], pmin[tid + w]);
      pmax[tid] = fmax(pmax[tid], pmax[tid + w]);
    }
    __syncthreads();
  }
  if (tid == 0) {
    sums[c] = ps[0];
    mins[c] = pmin[0];
    maxs[c] = pmax[0];
  }
}
}  // namespace meld
extern "C" size_t meld_col_stats_temp_bytes(int d) { return sizeof(double) * 3 * (size_t)meld::COL_STATS_GRID * (size_t)(d > 0 ? d : 1); }
extern "C" int meld_col_stats_f64(const double* X, int64_t N, int d, double* sums, double* mins, double* maxs, void* temp, size_t temp_bytes,
                                  meld_stream_t stream) {
  using namespace meld;
  MELD_CHECK_ARG(X && sums && mins && maxs && temp && N > 0 && d > 0 && d <= 256, "meld_col_stats_f64: bad arguments (d must be <= 256)");
  MELD_CHECK_ARG(temp_bytes >= meld_col_stats_temp_bytes(d), "meld_col_stats_f64: temp too small");
  const int rpp = 256 / d;
  const int grid = (int)std::min<int64_t>(COL_STATS_GRID, ceil_div(N, rpp));
  hipLaunchKernelGGL(col_stats_kernel, dim3(grid), dim3(256), 0, S(stream), X, N, d, reinterpret_cast<double*>(temp));
  hipLaunchKernelGGL(col_stats_finish_kernel, dim3((unsigned)d), dim3(256), 0, S(stream), reinterpret_cast<const double*>(temp), grid, d,
                     sums, mins, maxs);
  MELD_LAUNCH_CHECK("col_stats_kernel");
  return MELD_OK;
}

extern "C" int meld_col_sums_f64(const double* X, int64_t N, int d, double* sums, meld_stream_t stream) {
  MELD_CHECK_ARG(X && sums && N > 0 && d > 0 && d <= 256, "meld_col_sums_f64: bad arguments (d must be <= 256)");
  MELD_HIP_CALL(hipMemsetAsync(sums, 0, sizeof(double) * d, S(stream)));
  const int rpp = 256 / d;
  int grid = (int)std::min<int64_t>(1024, ceil_div(N, rpp));
  hipLaunchKernelGGL(col_sums_kernel, dim3(grid), dim3(256), 0, S(stream), X, N, d, sums);
  MELD_LAUNCH_CHECK("col_sums_kernel");
  return MELD_OK;
}

extern "C" int meld_knn_prepare_refs(const double* X, int64_t N, int d, const double* mean, int KP, float* Rt,
                                     float* norm2, float* norm2_max, meld_stream_t stream) {
  MELD_CHECK_ARG(X && mean && Rt && norm2 && norm2_max && N > 0, "meld_knn_prepare_refs: null/empty argument");
  MELD_CHECK_ARG(KP == meld_knn_padded_dim(d), "meld_knn_prepare_refs: KP=%d does not match d=%d", KP, d);
  const int64_t n_pad = ceil_div(N, KNN_TS) * KNN_TS;
  hipLaunchKernelGGL(prepare_refs_kernel, dim3((unsigned)ceil_div(n_pad, 256)), dim3(256), 0, S(stream), X, N, d,
                     mean, KP, n_pad, Rt, norm2, norm2_max);
  MELD_LAUNCH_CHECK("prepare_refs_kernel");
  return MELD_OK;
}

extern "C" int meld_knn_prepare_queries(const double* X, int64_t N, int d, const double* mean, int KP,
                                        int64_t q_begin, int64_t q_count, float* Q, meld_stream_t stream) {
  MELD_CHECK_ARG(X && mean && Q && q_count > 0 && q_begin >= 0 && q_begin + q_count <= N,
                 "meld_knn_prepare_queries: bad row range");
  MELD_CHECK_ARG(KP == meld_knn_padded_dim(d), "meld_knn_prepare_queries: KP=%d does not match d=%d", KP, d);
  const int64_t q_pad = ceil_div(q_count, KNN_BQ) * KNN_BQ;
  hipLaunchKernelGGL(prepare_queries_kernel, dim3((unsigned)ceil_div(q_pad, 256)), dim3(256), 0, S(stream), X, d,
                     mean, KP, q_begin, q_count, q_pad, Q);
  MELD_LAUNCH_CHECK("prepare_queries_kernel");
  return MELD_OK;
}

extern "C" int meld_knn_topk(const float* Q, const float* Rt, int64_t n_ref, int KP, int64_t q_count, int ksel,
                             int32_t* cand_idx, float* cand_d2, int32_t* cand_cnt, meld_stream_t stream) {
  MELD_CHECK_ARG(Q && Rt && cand_idx && cand_d2 && cand_cnt, "meld_knn_topk: null pointer");
  MELD_CHECK_ARG(n_ref > 0 && n_ref < (int64_t)1 << 31 && q_count > 0, "meld_knn_topk: bad sizes");
  const int cap = meld_knn_row_capacity(ksel);
  if (cap < 0) return cap;
  const int n_tiles = (int)ceil_div(n_ref, KNN_TS);
  const unsigned grid = (unsigned)ceil_div(q_count, KNN_BQ);
#define MELD_KNN_CASE(KPV)                                                                                   \
  case KPV:                                                                                                  \
    hipLaunchKernelGGL(knn_topk_kernel<KPV>, dim3(grid), dim3(KNN_THREADS), 0, S(stream), Q, Rt, (int)n_ref, \
                       n_tiles, ksel, cap, cand_idx, cand_d2, cand_cnt);                                     \
    break;
  switch (KP) {
    MELD_KNN_CASE(8)
    MELD_KNN_CASE(12)
    MELD_KNN_CASE(16)
    MELD_KNN_CASE(24)
    MELD_KNN_CASE(32)
    MELD_KNN_CASE(40)
    MELD_KNN_CASE(52)
    MELD_KNN_CASE(64)
    MELD_KNN_CASE(80)
    MELD_KNN_CASE(104)
    MELD_KNN_CASE(128)
    default:
      set_err("meld_knn_topk: KP=%d is not an instantiated size", KP);
      return MELD_ERR_UNSUPPORTED;
  }
#undef MELD_KNN_CASE
  MELD_LAUNCH_CHECK("knn_topk_kernel");
  return MELD_OK;
}
