// reorder.hip -- helper for the cache-locality permutation of the cells.
//
// No reference counterpart: graphtools / pygsp keep the cells in input order.  The Chebyshev
// recurrence (spmm.hip) gathers x[col] for every nonzero of W; with cells in arbitrary order each
// gather touches its own 128-byte line of a vector that exceeds an XCD's 4 MiB L2.  Ordering the
// cells so that graph neighbours are close in index turns most gathers into L1/L2 hits.  The
// ordering is a two-level nearest-centroid assignment (host code in meld_amd/reorder.py); this
// kernel is its only device-side arithmetic: nearest of a small set of centroids for every cell.
#include "common.hpp"

#include <algorithm>

#include <rocprim/rocprim.hpp>

#include <cstdlib>
#include <cstring>

namespace meld {

constexpr int AS_CT = 32;    // centroids per LDS tile
constexpr int AS_DMAX = 128;  // largest supported dimension

// out[i] = argmin_c |X[i] - C[c]|^2 over one shared set of n_cent centroids.
// Workgroup = 256 points; centroids stream through LDS in tiles of 32, laid out [k][c] so that one
// broadcast ds_read_b128 serves four centroids; 32 running sums per thread stay in registers.
__global__ __launch_bounds__(256) void assign_nearest_shared_kernel(const double* __restrict__ X, int64_t N, int d,
                                                                    const double* __restrict__ cents, int n_cent,
                                                                    int* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float cs[AS_DMAX][AS_CT];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const double* xi = X + (i < N ? i : N - 1) * d;
  float best = INFINITY;
  int arg = 0;
  for (int c0 = 0; c0 < n_cent; c0 += AS_CT) {
    __syncthreads();
    for (int u = threadIdx.x; u < AS_CT * d; u += blockDim.x) {
      const int c = u / d, k = u % d;
      cs[k][c] = (c0 + c < n_cent) ? (float)cents[(int64_t)(c0 + c) * d + k] : 3.0e18f;
    }
    __syncthreads();
    float acc[AS_CT];
#pragma unroll
    for (int c = 0; c < AS_CT; ++c) acc[c] = 0.0f;
    for (int k = 0; k < d; ++k) {
      const float xk = (float)xi[k];
      const float4* row = reinterpret_cast<const float4*>(cs[k]);
#pragma unroll
      for (int c4 = 0; c4 < AS_CT / 4; ++c4) {
        const float4 cv = row[c4];
        float t;
        t = xk - cv.x; acc[4 * c4 + 0] = fmaf(t, t, acc[4 * c4 + 0]);
        t = xk - cv.y; acc[4 * c4 + 1] = fmaf(t, t, acc[4 * c4 + 1]);
        t = xk - cv.z; acc[4 * c4 + 2] = fmaf(t, t, acc[4 * c4 + 2]);
        t = xk - cv.w; acc[4 * c4 + 3] = fmaf(t, t, acc[4 * c4 + 3]);
      }
    }
#pragma unroll
    for (int c = 0; c < AS_CT; ++c)
      if (acc[c] < best) {
        best = acc[c];
        arg = c0 + c;
      }
  }
  if (i < N) out[i] = arg;
}

// out[i] = argmin over the n_per_group centroids of point i's own group.  LP lanes share a point
// (LP = power of two >= min(n_per_group, 64)): lane l of the team scores centroids l, l + LP, ...,
// so the point's row is one broadcast read per coordinate instead of a 400-byte-strided gather
// across the wave (the one-thread-per-point version spent 5.5 ms at 1M x 50 on those gathers), and
// the team's centroids -- one group's, the points are walked in `order`, sorted by group -- come from L1.
template <int LP>
__global__ __launch_bounds__(256) void assign_nearest_grouped_kernel(const double* __restrict__ X, int64_t N, int d,
                                                                     const double* __restrict__ cents, int n_per_group,
                                                                     const int* __restrict__ group,
                                                                     const int64_t* __restrict__ order,
                                                                     int* __restrict__ out) {
  constexpr int PPB = 256 / LP;  // points per block
  const int team = threadIdx.x / LP, l = threadIdx.x % LP;
  const int64_t t = (int64_t)blockIdx.x * PPB + team;
  if (t >= N) return;
  const int64_t i = order ? order[t] : t;
  const double* xi = X + i * d;
  const double* cb = cents + (int64_t)group[i] * n_per_group * d;
  float best = INFINITY;
  int arg = 0x7fffffff;
  for (int c = l; c < n_per_group; c += LP) {
    const double* cc = cb + (int64_t)c * d;
    float s = 0.0f;
    for (int k = 0; k < d; ++k) {
      const float t2 = (float)(xi[k] - cc[k]);
      s = fmaf(t2, t2, s);
    }
    if (s < best) {  // ascending c: the first minimum wins, as in a serial scan
      best = s;
      arg = c;
    }
  }
  // team argmin (ties -> smallest centroid index)
#pragma unroll
  for (int off = LP / 2; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off, 64);
    const int oa = __shfl_xor(arg, off, 64);
    if (ob < best || (ob == best && oa < arg)) {
      best = ob;
      arg = oa;
    }
  }
  if (l == 0) out[i] = arg;
}

// Tiled form of both assignments (the one the launcher uses): a workgroup takes 64 points in processing
// order (`order`, sorted by group, or the identity), reads their rows with coalesced loads into LDS as fp32,
// then stages the centroids of one group at a time ([k][c] in LDS) and lets thread (point p, part q) score
// n_per_group / 4 of them from LDS -- both operands come from LDS, the only global traffic is one coalesced
// pass over X (the team-per-point kernel above re-reads every point row and 16 strided centroid rows through
// L1: 1.14 ms per level at 1M x 50, this one ~0.3 ms).  group == nullptr: one shared centroid set.
constexpr int AT_PTS = 64;   // points per workgroup
constexpr int AT_CMAX = 64;  // centroids per group (fan-out) at most
template <int PER>  // centroids per quarter: n_per_group <= 4 PER
__global__ __launch_bounds__(256) void assign_nearest_tiled_kernel(const double* __restrict__ X, int64_t N, int d,
                                                                   const double* __restrict__ cents, int n_per_group,
                                                                   const int* __restrict__ group,
                                                                   const int64_t* __restrict__ order,
                                                                   int* __restrict__ out) {
  __shared__ float xs[AT_PTS][AS_DMAX + 1];
  __shared__ __attribute__((aligned(16))) float cs[AS_DMAX][AT_CMAX];
  __shared__ int64_t s_idx[AT_PTS];
  __shared__ int s_grp[AT_PTS];
  const int tid = threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.x * AT_PTS;
  const int n_here = (int)min((int64_t)AT_PTS, N - t0);
  if (tid < AT_PTS) {
    const int64_t i = tid < n_here ? (order ? order[t0 + tid] : t0 + tid) : -1;
    s_idx[tid] = i;
    s_grp[tid] = (i >= 0 && group) ? group[i] : 0;
  }
  __syncthreads();
  for (int u = tid; u < AT_PTS * d; u += 256) {
    const int r = u / d, k = u - r * d;
    const int64_t i = s_idx[r];
    xs[r][k] = i >= 0 ? (float)X[i * d + k] : 0.0f;
  }
  const int p = tid >> 2, q = tid & 3;           // point, quarter of the centroid set
  constexpr int per = PER;
  const int g_mine = s_grp[p];
  // groups present in this workgroup: the points are sorted by group, so they form a range
  int g_lo = s_grp[0], g_hi = s_grp[0];
  for (int r = 1; r < n_here; ++r) {
    g_lo = min(g_lo, s_grp[r]);
    g_hi = max(g_hi, s_grp[r]);
  }
  float best = INFINITY;
  int arg = 0x7fffffff;
  for (int g = g_lo; g <= g_hi; ++g) {
    __syncthreads();  // (also orders the xs stores before their first use)
    const double* cb = cents + (int64_t)g * n_per_group * d;
    for (int u = tid; u < n_per_group * d; u += 256) {
      const int c = u / d, k = u - c * d;
      cs[k][c] = (float)cb[u];
    }
    __syncthreads();
    if (g_mine != g || s_idx[p] < 0) continue;
    float acc[PER];
#pragma unroll
    for (int c = 0; c < PER; ++c) acc[c] = 0.0f;
    const int c_base = q * per;
    for (int k = 0; k < d; ++k) {
      const float xk = xs[p][k];
#pragma unroll
      for (int c = 0; c < PER; ++c) {
        const float t = xk - cs[k][c_base + c];
        acc[c] = fmaf(t, t, acc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < PER; ++c) {
      if (c_base + c < n_per_group && acc[c] < best) {  // ascending c: the first minimum wins
        best = acc[c];
        arg = c_base + c;
      }
    }
  }
  // the four quarters of a point are adjacent lanes: argmin (ties -> smallest centroid index)
#pragma unroll
  for (int off = 1; off < 4; off <<= 1) {
    const float ob = __shfl_xor(best, off, 64);
    const int oa = __shfl_xor(arg, off, 64);
    if (ob < best || (ob == best && oa < arg)) {
      best = ob;
      arg = oa;
    }
  }
  if (q == 0 && s_idx[p] >= 0) out[s_idx[p]] = arg;
}

// Greedy nearest-neighbour chain over the m (<= 64) rows of every group P[g] ([n_groups][m][d]): one wave
// per group, lane i = row i.  Starts at the row with the smallest first coordinate; rank[g][i] = position
// of row i along the chain.  (Was a host loop: the device -> host copy, the NumPy walk and the copy back
// cost 4.6 ms at 1M cells with three levels.)
__global__ __launch_bounds__(64) void chain_order_kernel(const double* __restrict__ P, int m, int d,
                                                         int* __restrict__ rank, int use_lds) {
  // the group's rows are staged in LDS once (row stride d + 1 doubles): every step of the walk reads the current row
  // (broadcast) and each lane its own -- from global memory that was 2 d dependent-latency loads per lane and step, 5 us per
  // step and 0.3-0.4 ms per level of the ordering, all of it on the critical path of the level
  extern __shared__ __attribute__((aligned(16))) double chain_lds[];
  const int lane = threadIdx.x;
  const double* Pg0 = P + (size_t)blockIdx.x * m * d;
  const int ld = use_lds ? d + 1 : d;  // (rows too wide for the LDS a launch may ask for stay in global memory)
  const double* Pr = use_lds ? chain_lds : Pg0;
  if (use_lds) {
    for (int u = lane; u < m * d; u += 64) {
      const int r = u / d, k = u - r * d;
      chain_lds[r * ld + k] = Pg0[u];
    }
    __syncthreads();
  }
  int* rg = rank + (size_t)blockIdx.x * m;
  // lowest index among the minima: a DPP minimum (in-row butterflies, then lane 15 / 31 handed to the rows above) and a ballot.
  // (The (value, index) butterfly over ds_bpermute was 18 dependent LDS round trips per step of a walk that is one wave and up
  // to 64 steps long, on the critical path of every level of the ordering.)
  auto argmin = [&](double v) {
    double mn = v;
#define CHAIN_DPP_MIN(CTRL, ROWS)                                                                                         \
    {                                                                                                                      \
      const int lo = __builtin_amdgcn_update_dpp(__double2loint(mn), __double2loint(mn), CTRL, ROWS, 0xF, false);          \
      const int hi = __builtin_amdgcn_update_dpp(__double2hiint(mn), __double2hiint(mn), CTRL, ROWS, 0xF, false);          \
      mn = fmin(mn, __hiloint2double(hi, lo));                                                                             \
    }
    CHAIN_DPP_MIN(0xB1, 0xF)   // quad_perm [1, 0, 3, 2]
    CHAIN_DPP_MIN(0x4E, 0xF)   // quad_perm [2, 3, 0, 1]
    CHAIN_DPP_MIN(0x141, 0xF)  // row_half_mirror
    CHAIN_DPP_MIN(0x140, 0xF)  // row_mirror: every lane holds its row's minimum
    CHAIN_DPP_MIN(0x142, 0xA)  // row_bcast15 -> rows 1 and 3
    CHAIN_DPP_MIN(0x143, 0xC)  // row_bcast31 -> rows 2 and 3: lane 63 holds the wave's
#undef CHAIN_DPP_MIN
    mn = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mn), 63), __builtin_amdgcn_readlane(__double2loint(mn), 63));
    return __ffsll((long long)__ballot(v == mn)) - 1;
  };
  bool used = lane >= m;
  int cur = argmin(lane < m ? Pr[lane * ld] : INFINITY);
  if (lane == cur) {
    used = true;
    rg[lane] = 0;
  }
  for (int step = 1; step < m; ++step) {
    double dist = INFINITY;
    if (!used) {
      dist = 0.0;
      int k = 0;
      for (; k + 8 <= d; k += 8) {  // (eight pairs of reads in flight; same summation order)
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = Pr[lane * ld + k + u] - Pr[cur * ld + k + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) dist = fma(t[u], t[u], dist);
      }
      for (; k < d; ++k) {
        const double t = Pr[lane * ld + k] - Pr[cur * ld + k];
        dist = fma(t, t, dist);
      }
    }
    cur = argmin(dist);
    if (lane == cur) {
      used = true;
      rg[lane] = step;
    }
  }
}

// The same assignment on the matrix pipe: d2(x, c) - |x|^2 = |c|^2 - 2 x.c, the products on v_mfma_f32_32x32x2_f32 (fp32
// operands and accumulation; 157 TF/s against the ~13 TF/s the LDS-fed FMA loops above reach), |c|^2 as the start value of
// the accumulators.  One wave = 32 points (rows) x up to 64 centroids (two column blocks) per pass, K two coordinates
// at a time: lane (j, h) feeds point j's coordinate 2s + h on the A side and centroid j's on the B side.  The minimum
// of a row is a butterfly over the 32 lanes of its half on keys (ordered value bits, low 6 bits replaced by the centroid
// index): ties and near-ties (2^-18 relative) go to the lower index -- any assignment gives a valid ordering, it only has
// to be the same on every run and every rank, which fixed-order fp32 arithmetic is.  Points are taken in `order`
// (sorted by group) so that a workgroup meets one or two groups; it loads each group's centroids into LDS once and
// its waves skip the chunks that hold no point of it.
constexpr int AM_CHUNKS = 4;                   // chunks of 32 points per wave
constexpr int AM_WG_PTS = 4 * AM_CHUNKS * 32;  // points per workgroup
typedef float am_f32x16 __attribute__((ext_vector_type(16)));
template <int NB>  // centroid blocks of 32: n_per_group <= 32 NB
__global__ __launch_bounds__(256) void assign_nearest_mfma_kernel(const double* __restrict__ X, int64_t N, int d,
                                                                  const double* __restrict__ cents, int n_per_group,
                                                                  const int* __restrict__ group,
                                                                  const int64_t* __restrict__ order, int* __restrict__ out) {
  // LDS (dynamic, sized for d): s_pi [4][32] int64 | cs [KP][32 NB] centroids, [k][n], zero beyond d / n_per_group |
  // cn [32 NB] |c|^2 (+inf for the padding columns) | xs [4][32][KP + 1] per wave: the points of the chunk in flight, times -2
  extern __shared__ __attribute__((aligned(16))) unsigned char am_lds[];
  const int KP = (d + 1) & ~1, XS = KP + 1;
  int64_t(*s_pi)[32] = reinterpret_cast<int64_t(*)[32]>(am_lds);
  float(*cs)[32 * NB] = reinterpret_cast<float(*)[32 * NB]>(am_lds + 4 * 32 * 8);
  float* cn = reinterpret_cast<float*>(am_lds + 4 * 32 * 8) + KP * 32 * NB;
  float* xs = cn + 32 * NB;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int64_t p0 = (int64_t)blockIdx.x * AM_WG_PTS;
  const int n_here = (int)min((int64_t)AM_WG_PTS, N - p0);
  int g_lo = 0, g_hi = 0;
  if (group) {  // positions are sorted by group: the workgroup's groups are a range
    g_lo = group[order ? order[p0] : p0];
    g_hi = group[order ? order[p0 + n_here - 1] : p0 + n_here - 1];
  }
  // this lane's point in each of the wave's chunks
  int64_t pi[AM_CHUNKS];
  int pg[AM_CHUNKS];
#pragma unroll
  for (int c = 0; c < AM_CHUNKS; ++c) {
    const int rel = (w * AM_CHUNKS + c) * 32 + j;
    pi[c] = rel < n_here ? (order ? order[p0 + rel] : p0 + rel) : -1;
    pg[c] = (pi[c] >= 0 && group) ? group[pi[c]] : (pi[c] >= 0 ? 0 : -1);
  }
  for (int g = g_lo; g <= g_hi; ++g) {
    __syncthreads();
    const double* cb = cents + (int64_t)g * n_per_group * d;
    for (int u = tid; u < KP * 32 * NB; u += 256) {
      const int n = u / KP, k = u - n * KP;  // (consecutive threads read consecutive coordinates of a centroid)
      cs[k][n] = (n < n_per_group && k < d) ? (float)cb[(int64_t)n * d + k] : 0.0f;
    }
    __syncthreads();
    if (tid < 32 * NB) {
      float sq = 0.0f;
      for (int k = 0; k < d; ++k) sq = fmaf(cs[k][tid], cs[k][tid], sq);
      cn[tid] = tid < n_per_group ? sq : INFINITY;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < AM_CHUNKS; ++c) {
      if (!__any(pg[c] == g)) continue;
      am_f32x16 acc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float c2 = cn[nb * 32 + j];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nb][r] = c2;
      }
      // the chunk's 32 rows go through LDS: consecutive lanes read consecutive doubles (512 contiguous bytes per
      // instruction, at most two rows), where one lane per row and coordinate touched 32 lines per instruction
      // and the loads, not the products, set the pace (1M x 64 centroids: 0.33 ms, 0.14 without the loads; 0.24 this way)
      float* xw = xs + (size_t)w * 32 * XS;
      if (h == 0) s_pi[w][j] = pi[c];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      {
        int row = lane / d, k = lane - row * d;  // element lane + 64 it of the 32 x d block
        for (int it = 0; it < (32 * d + 63) / 64; ++it) {
          if (row < 32) {
            const int64_t pr = s_pi[w][row];
            xw[row * XS + k] = pr >= 0 ? -2.0f * (float)X[pr * d + k] : 0.0f;
          }
          k += 64;
          while (k >= d) {
            k -= d;
            ++row;
          }
        }
        if ((d & 1) && lane < 32) xw[lane * XS + d] = 0.0f;
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      for (int s2 = 0; s2 < KP / 2; ++s2) {
        const int k = 2 * s2 + h;
        const float a = xw[j * XS + k];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, cs[k][nb * 32 + j], acc[nb], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the next chunk overwrites the rows)
      // row m = 8 (r / 4) + 4 h + r % 4 of the block is point m of the chunk; its minimum over the columns
      // (a butterfly that halves what a lane carries at every step -- 16 exchanges instead of 16 x 5, lane l ending with row
      // (l >> 1) & 15 -- keeps the 16 keys live across the exchanges: the ordering took 2.1 instead of 1.8 ms at 1M cells; not kept)
      unsigned mine = 0xffffffffu;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        unsigned key = 0xffffffffu;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          unsigned u = __float_as_uint(acc[nb][r] + 0.0f);
          u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
          key = min(key, (u & ~63u) | (unsigned)(nb * 32 + j));
        }
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) key = min(key, (unsigned)__shfl_xor((int)key, off, 64));
        const int m = 8 * (r >> 2) + 4 * h + (r & 3);
        if (m == j) mine = key;
      }
      // (point j's row sits in half (j >> 2) & 1)
      if (((j >> 2) & 1) == h && pg[c] == g) out[pi[c]] = (int)(mine & 63u);
    }
  }
}

// Glue of the ordering levels (meld_amd/reorder.py), one launch each instead of a dozen tensor operations: the stage was
// bound by the host issuing ~140 small launches (3.4 ms of CPU for 2.7 ms of GPU work at 1M cells).
// starts[g] = first position of key g in the sorted keys, g = 0 .. n_groups (lower bound)
__global__ __launch_bounds__(256) void order_starts_kernel(const uint32_t* __restrict__ keys_sorted, int64_t n, int n_groups,
                                                           int64_t* __restrict__ starts) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > n_groups) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (keys_sorted[mid] < (uint32_t)g) lo = mid + 1; else hi = mid;
  }
  starts[g] = lo;
}
// sub-centroids of every group: f evenly spaced members (in sorted order) of the group's cells
__global__ __launch_bounds__(256) void order_pick_centroids_kernel(const double* __restrict__ X, int64_t N, int d,
                                                                   const int64_t* __restrict__ order,
                                                                   const int64_t* __restrict__ starts, int n_groups, int f,
                                                                   double* __restrict__ cents) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= (int64_t)n_groups * f * d) return;
  const int64_t gc = u / d;
  const int k = (int)(u - gc * d);
  const int g = (int)(gc / f), c = (int)(gc - (int64_t)g * f);
  const int64_t s0 = starts[g], cnt = starts[g + 1] - s0;
  const double frac = ((double)c + 0.5) / (double)f;
  int64_t pick = s0 + (int64_t)(frac * (double)cnt);
  pick = min(pick, s0 + max(cnt - 1, (int64_t)0));
  pick = min(max(pick, (int64_t)0), N - 1);
  cents[u] = X[order[pick] * d + k];
}
// key of the next level: position of the cell's child along its group's chain
__global__ __launch_bounds__(256) void order_update_keys_kernel(uint32_t* __restrict__ key, const int32_t* __restrict__ child,
                                                                const int32_t* __restrict__ rank, int64_t n, int f) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t base = key[i] * (uint32_t)f;
  key[i] = base + (uint32_t)rank[base + (uint32_t)child[i]];
}

}  // namespace meld

using namespace meld;

// rocPRIM sorts fewer than 2^20 pairs of 32-bit keys by a block sort + ~10 merge passes (its radix_sort dispatch: 1M cells fall
// just below the limit -- 23 launches, 165 us per argsort, three per ordering); for keys of at most 16 bits it takes the
// onesweep radix path from 100k elements on (histogram, scan, one pass per 8 bits).  The ordering's keys are group numbers
// below 2^16, so they are narrowed to 16 bits for the sort and widened again behind it: ~60 us per argsort at 1M cells.
namespace meld {
__global__ __launch_bounds__(256) void narrow_keys_kernel(const uint32_t* __restrict__ in, uint16_t* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint16_t)in[i];
}
__global__ __launch_bounds__(256) void widen_keys_kernel(const uint16_t* __restrict__ in, uint32_t* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
constexpr int64_t ARGSORT_NARROW_MIN = 100000;  // rocPRIM's own bound for the 16-bit path
static size_t argsort_round(size_t b) { return (b + 255) & ~(size_t)255; }
}  // namespace meld

extern "C" size_t meld_argsort_u32_temp_bytes(int64_t n) {
  size_t bytes = 0, bytes16 = 0;
  uint32_t* k = nullptr;
  uint16_t* k16 = nullptr;
  int64_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, rocprim::make_counting_iterator<int64_t>(0), v, (size_t)n, 0u, 32u);
  (void)rocprim::radix_sort_pairs(nullptr, bytes16, k16, k16, rocprim::make_counting_iterator<int64_t>(0), v, (size_t)n, 0u, 16u);
  // (the narrow path: two 16-bit key arrays in front of rocPRIM's own scratch)
  return std::max(bytes, argsort_round(bytes16) + 2 * argsort_round((size_t)n * 2)) + 512;
}
extern "C" int meld_argsort_u32(const uint32_t* keys, int64_t n, int end_bit, int64_t* order, uint32_t* keys_sorted, void* temp,
                                size_t temp_bytes, meld_stream_t stream) {
  MELD_CHECK_ARG(keys && order && keys_sorted && temp && n > 0 && end_bit > 0 && end_bit <= 32, "meld_argsort_u32: bad arguments");
  MELD_CHECK_ARG(temp_bytes >= meld_argsort_u32_temp_bytes(n), "meld_argsort_u32: temp too small");
  if (end_bit <= 16 && n >= ARGSORT_NARROW_MIN) {
    char* p = reinterpret_cast<char*>(temp);
    p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 255) & ~(uintptr_t)255);
    uint16_t* k_in = reinterpret_cast<uint16_t*>(p);
    uint16_t* k_out = reinterpret_cast<uint16_t*>(p + argsort_round((size_t)n * 2));
    void* scratch = p + 2 * argsort_round((size_t)n * 2);
    size_t bytes = temp_bytes - (size_t)(reinterpret_cast<char*>(scratch) - reinterpret_cast<char*>(temp));
    const unsigned grid = (unsigned)ceil_div(n, 256);
    hipLaunchKernelGGL(narrow_keys_kernel, dim3(grid), dim3(256), 0, S(stream), keys, k_in, n);
    MELD_HIP_CALL(rocprim::radix_sort_pairs(scratch, bytes, k_in, k_out, rocprim::make_counting_iterator<int64_t>(0), order, (size_t)n,
                                            0u, (unsigned)end_bit, S(stream)));
    hipLaunchKernelGGL(widen_keys_kernel, dim3(grid), dim3(256), 0, S(stream), k_out, keys_sorted, n);
    MELD_LAUNCH_CHECK("meld_argsort_u32");
    return MELD_OK;
  }
  size_t bytes = temp_bytes;
  MELD_HIP_CALL(rocprim::radix_sort_pairs(temp, bytes, keys, keys_sorted, rocprim::make_counting_iterator<int64_t>(0), order, (size_t)n,
                                          0u, (unsigned)end_bit, S(stream)));
  return MELD_OK;
}
extern "C" int meld_order_starts(const uint32_t* keys_sorted, int64_t n, int n_groups, int64_t* starts, meld_stream_t stream) {
  MELD_CHECK_ARG(keys_sorted && starts && n > 0 && n_groups > 0, "meld_order_starts: bad arguments");
  hipLaunchKernelGGL(order_starts_kernel, dim3((unsigned)ceil_div((int64_t)n_groups + 1, 256)), dim3(256), 0, S(stream), keys_sorted, n,
                     n_groups, starts);
  MELD_LAUNCH_CHECK("order_starts_kernel");
  return MELD_OK;
}
extern "C" int meld_order_pick_centroids(const double* X, int64_t N, int d, const int64_t* order, const int64_t* starts, int n_groups,
                                         int f, double* cents, meld_stream_t stream) {
  MELD_CHECK_ARG(X && order && starts && cents && N > 0 && d > 0 && n_groups > 0 && f > 0, "meld_order_pick_centroids: bad arguments");
  hipLaunchKernelGGL(order_pick_centroids_kernel, dim3((unsigned)ceil_div((int64_t)n_groups * f * d, 256)), dim3(256), 0, S(stream), X, N,
                     d, order, starts, n_groups, f, cents);
  MELD_LAUNCH_CHECK("order_pick_centroids_kernel");
  return MELD_OK;
}
extern "C" int meld_order_update_keys(uint32_t* key, const int32_t* child, const int32_t* rank, int64_t n, int f, meld_stream_t stream) {
  MELD_CHECK_ARG(key && child && rank && n > 0 && f > 0, "meld_order_update_keys: bad arguments");
  hipLaunchKernelGGL(order_update_keys_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, S(stream), key, child, rank, n, f);
  MELD_LAUNCH_CHECK("order_update_keys_kernel");
  return MELD_OK;
}

extern "C" int meld_assign_nearest(const double* X, int64_t N, int d, const double* cents, int n_per_group,
                                   const int32_t* group, const int64_t* order, int32_t* out, meld_stream_t stream) {
  MELD_CHECK_ARG(X && cents && out && N > 0 && d > 0 && n_per_group > 0, "meld_assign_nearest: bad arguments");
  static const bool valu_only = meld_dev_getenv("MELD_ASSIGN") && !strcmp(meld_dev_getenv("MELD_ASSIGN"), "valu");  // (A/B: the FMA kernels)
  if (!valu_only && d <= AS_DMAX && n_per_group <= 64 && (group == nullptr || order != nullptr)) {
    const unsigned grid = (unsigned)ceil_div(N, (int64_t)AM_WG_PTS);
    const int KP = (d + 1) & ~1, nbl = n_per_group <= 32 ? 1 : 2;
    const size_t lds = 4 * 32 * 8 + sizeof(float) * ((size_t)KP * 32 * nbl + 32 * nbl + (size_t)4 * 32 * (KP + 1));
    if (nbl == 1)
      hipLaunchKernelGGL((assign_nearest_mfma_kernel<1>), dim3(grid), dim3(256), lds, S(stream), X, N, d, cents, n_per_group, group, order, out);
    else
      hipLaunchKernelGGL((assign_nearest_mfma_kernel<2>), dim3(grid), dim3(256), lds, S(stream), X, N, d, cents, n_per_group, group, order, out);
    MELD_LAUNCH_CHECK("assign_nearest_mfma_kernel");
    return MELD_OK;
  }
  if (d <= AS_DMAX && n_per_group <= AT_CMAX && (group == nullptr || order != nullptr)) {
#define MELD_ASSIGN_TILED(PERV)                                                                                         \
  hipLaunchKernelGGL((assign_nearest_tiled_kernel<PERV>), dim3((unsigned)ceil_div(N, AT_PTS)), dim3(256), 0, S(stream), X, \
                     N, d, cents, n_per_group, group, order, out)
    if (n_per_group <= 16) {
      MELD_ASSIGN_TILED(4);
    } else if (n_per_group <= 32) {
      MELD_ASSIGN_TILED(8);
    } else {
      MELD_ASSIGN_TILED(16);
    }
#undef MELD_ASSIGN_TILED
    MELD_LAUNCH_CHECK("assign_nearest_tiled_kernel");
    return MELD_OK;
  }
  if (group == nullptr) {
    MELD_CHECK_ARG(d <= AS_DMAX, "meld_assign_nearest: d=%d exceeds %d", d, AS_DMAX);
    hipLaunchKernelGGL(assign_nearest_shared_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, S(stream), X, N,
                       d, cents, n_per_group, out);
  } else {
#define MELD_ASSIGN_GROUPED(LPV)                                                                                        \
  hipLaunchKernelGGL((assign_nearest_grouped_kernel<LPV>), dim3((unsigned)ceil_div(N, (int64_t)(256 / LPV))), dim3(256), 0, \
                     S(stream), X, N, d, cents, n_per_group, group, order, out)
    if (n_per_group <= 4) {
      MELD_ASSIGN_GROUPED(4);
    } else if (n_per_group <= 8) {
      MELD_ASSIGN_GROUPED(8);
    } else if (n_per_group <= 16) {
      MELD_ASSIGN_GROUPED(16);
    } else if (n_per_group <= 32) {
      MELD_ASSIGN_GROUPED(32);
    } else {
      MELD_ASSIGN_GROUPED(64);
    }
#undef MELD_ASSIGN_GROUPED
  }
  MELD_LAUNCH_CHECK("assign_nearest_kernel");
  return MELD_OK;
}

// out[i] = X[perm[i]] for rows of d doubles: the cells in the device order.  32 lanes per row, 16-byte pieces when the rows are
// 16-byte aligned (d even); the library gather moved the 2 x 400 MB of 1M x 50 cells at 1.5 TB/s.
__global__ __launch_bounds__(256) void gather_rows_kernel(const double* __restrict__ X, const int64_t* __restrict__ perm, int64_t N,
                                                          int d, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int l = threadIdx.x & 31;
  if (i >= N) return;
  const int64_t src = perm[i];
  if ((d & 1) == 0) {
    const double2* a = reinterpret_cast<const double2*>(X + src * d);
    double2* b = reinterpret_cast<double2*>(out + i * d);
    for (int c = l; c < d / 2; c += 32) b[c] = a[c];
  } else {
    for (int c = l; c < d; c += 32) out[i * d + c] = X[src * d + c];
  }
}

extern "C" int meld_gather_rows_f64(const double* X, const int64_t* perm, int64_t N, int d, double* out, meld_stream_t stream) {
  MELD_CHECK_ARG(X && perm && out && N >= 0 && d > 0, "meld_gather_rows_f64: bad arguments");
  if (N == 0) return MELD_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)ceil_div(N, 8)), dim3(256), 0, S(stream), X, perm, N, d, out);
  MELD_LAUNCH_CHECK("gather_rows_kernel");
  return MELD_OK;
}

extern "C" int meld_chain_order(const double* P, int64_t n_groups, int m, int d, int32_t* rank, meld_stream_t stream) {
  MELD_CHECK_ARG(P && rank && n_groups > 0 && m >= 1 && m <= 64 && d > 0, "meld_chain_order: bad arguments (1 <= m <= 64)");
  const size_t chain_bytes = sizeof(double) * (size_t)m * (d + 1);
  const int use_lds = chain_bytes <= 60 * 1024 ? 1 : 0;
  hipLaunchKernelGGL(chain_order_kernel, dim3((unsigned)n_groups), dim3(64), use_lds ? chain_bytes : 0, S(stream), P, m, d, rank, use_lds);
  MELD_LAUNCH_CHECK("chain_order_kernel");
  return MELD_OK;
}
