// reorder.hip -- helper for the cache-locality permutation of the cells.
//
// No reference counterpart: graphtools / pygsp keep the cells in input order.  The Chebyshev
// recurrence (spmm.hip) gathers x[col] for every nonzero of W; with cells in arbitrary order each
// gather touches its own 128-byte line of a vector that exceeds an XCD's 4 MiB L2.  Ordering the
// cells so that graph neighbours are close in index turns most gathers into L1/L2 hits.  The
// ordering is a two-level nearest-centroid assignment (host code in meld_amd/reorder.py); this
// kernel is its only device-side arithmetic: nearest of a small set of centroids for every cell.
#include "common.hpp"

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
                                                         int* __restrict__ rank) {
  const int lane = threadIdx.x;
  const double* Pg = P + (size_t)blockIdx.x * m * d;
  int* rg = rank + (size_t)blockIdx.x * m;
  auto argmin = [&](double v) {  // lowest index among the minima
    int i = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(v, off, 64);
      const int oi = __shfl_xor(i, off, 64);
      if (ov < v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
      }
    }
    return i;
  };
  bool used = lane >= m;
  int cur = argmin(lane < m ? Pg[(size_t)lane * d] : INFINITY);
  if (lane == cur) {
    used = true;
    rg[lane] = 0;
  }
  for (int step = 1; step < m; ++step) {
    double dist = INFINITY;
    if (!used) {
      dist = 0.0;
      for (int k = 0; k < d; ++k) {
        const double t = Pg[(size_t)lane * d + k] - Pg[(size_t)cur * d + k];
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

}  // namespace meld

using namespace meld;

extern "C" int meld_assign_nearest(const double* X, int64_t N, int d, const double* cents, int n_per_group,
                                   const int32_t* group, const int64_t* order, int32_t* out, meld_stream_t stream) {
  MELD_CHECK_ARG(X && cents && out && N > 0 && d > 0 && n_per_group > 0, "meld_assign_nearest: bad arguments");
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

extern "C" int meld_chain_order(const double* P, int64_t n_groups, int m, int d, int32_t* rank, meld_stream_t stream) {
  MELD_CHECK_ARG(P && rank && n_groups > 0 && m >= 1 && m <= 64 && d > 0, "meld_chain_order: bad arguments (1 <= m <= 64)");
  hipLaunchKernelGGL(chain_order_kernel, dim3((unsigned)n_groups), dim3(64), 0, S(stream), P, m, d, rank);
  MELD_LAUNCH_CHECK("chain_order_kernel");
  return MELD_OK;
}
