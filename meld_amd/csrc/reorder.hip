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

// out[i] = argmin over the n_per_group centroids of point i's own group.  Threads walk the points
// in `order` (points sorted by group), so a wave reads one group's centroids -- a broadcast the L1
// serves -- and only the point rows themselves are gathered.
__global__ __launch_bounds__(256) void assign_nearest_grouped_kernel(const double* __restrict__ X, int64_t N, int d,
                                                                     const double* __restrict__ cents, int n_per_group,
                                                                     const int* __restrict__ group,
                                                                     const int64_t* __restrict__ order,
                                                                     int* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  const int64_t i = order ? order[t] : t;
  const double* xi = X + i * d;
  const double* cb = cents + (int64_t)group[i] * n_per_group * d;
  float best = INFINITY;
  int arg = 0;
  for (int c = 0; c < n_per_group; ++c) {
    const double* cc = cb + (int64_t)c * d;
    float s = 0.0f;
    for (int k = 0; k < d; ++k) {
      const float t2 = (float)(xi[k] - cc[k]);
      s = fmaf(t2, t2, s);
    }
    if (s < best) {
      best = s;
      arg = c;
    }
  }
  out[i] = arg;
}

}  // namespace meld

using namespace meld;

extern "C" int meld_assign_nearest(const double* X, int64_t N, int d, const double* cents, int n_per_group,
                                   const int32_t* group, const int64_t* order, int32_t* out, meld_stream_t stream) {
  MELD_CHECK_ARG(X && cents && out && N > 0 && d > 0 && n_per_group > 0, "meld_assign_nearest: bad arguments");
  if (group == nullptr) {
    MELD_CHECK_ARG(d <= AS_DMAX, "meld_assign_nearest: d=%d exceeds %d", d, AS_DMAX);
    hipLaunchKernelGGL(assign_nearest_shared_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, S(stream), X, N,
                       d, cents, n_per_group, out);
  } else {
    hipLaunchKernelGGL(assign_nearest_grouped_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, S(stream), X, N,
                       d, cents, n_per_group, group, order, out);
  }
  MELD_LAUNCH_CHECK("assign_nearest_kernel");
  return MELD_OK;
}
