// kmeans.hip -- Lloyd iteration (assignment + partial centroid sums) for VertexFrequencyCluster.predict.
//
// Replaces the KMeans step of reference meld/cluster.py:315-345 ([UPSTREAM sklearn.cluster.KMeans], Lloyd): one
// pass over the points assigns each to its nearest centroid and accumulates, per workgroup, the sums and counts of
// every cluster; the per-workgroup partials are written out and reduced in a fixed order by the caller, so the
// result does not depend on scheduling (no floating-point atomics on global memory).
//
// Shape: points [N, d] fp64 row-major with d <= 32, k <= 64 centroids staged in LDS; a thread owns a point, a
// workgroup 256 consecutive points per sweep (grid-stride), HBM-bound (one read of the points per iteration).
#include "common.hpp"

namespace meld {

constexpr int KM_DMAX = 32;
constexpr int KM_KMAX = 64;
constexpr int KM_THREADS = 256;

__global__ __launch_bounds__(KM_THREADS) void kmeans_assign_kernel(const double* __restrict__ X, int64_t n, int d,
                                                                   const double* __restrict__ cent, int k,
                                                                   int32_t* __restrict__ labels, double* __restrict__ part_sum,
                                                                   double* __restrict__ part_cnt, double* __restrict__ part_inertia) {
  __shared__ double s_c[KM_KMAX * KM_DMAX];
  __shared__ double s_sum[KM_KMAX * KM_DMAX];
  __shared__ double s_cnt[KM_KMAX];
  __shared__ double s_in[KM_THREADS / 64];
  const int tid = threadIdx.x;
  for (int i = tid; i < k * d; i += KM_THREADS) {
    s_c[i] = cent[i];
    s_sum[i] = 0.0;
  }
  for (int i = tid; i < k; i += KM_THREADS) s_cnt[i] = 0.0;
  __syncthreads();
  double inertia = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * KM_THREADS + tid; i < n; i += (int64_t)gridDim.x * KM_THREADS) {
    double x[KM_DMAX];
#pragma unroll 4
    for (int c = 0; c < d; ++c) x[c] = X[i * d + c];
    double best = 1e300;
    int bi = 0;
    for (int j = 0; j < k; ++j) {  // ties go to the lowest index (argmin convention)
      double s = 0.0;
      for (int c = 0; c < d; ++c) {
        const double t = x[c] - s_c[j * d + c];
        s += t * t;
      }
      if (s < best) {
        best = s;
        bi = j;
      }
    }
    labels[i] = bi;
    inertia += best;
    // LDS accumulation: the order of the additions into a cell varies, so the workgroup's partial sums are
    // reproducible only to rounding; the cross-workgroup reduction is done in a fixed order by the caller
    for (int c = 0; c < d; ++c) atomicAdd(&s_sum[bi * d + c], x[c]);
    atomicAdd(&s_cnt[bi], 1.0);
  }
  inertia = wave_sum(inertia);
  if ((tid & 63) == 0) s_in[tid >> 6] = inertia;
  __syncthreads();
  for (int i = tid; i < k * d; i += KM_THREADS) part_sum[(size_t)blockIdx.x * k * d + i] = s_sum[i];
  for (int i = tid; i < k; i += KM_THREADS) part_cnt[(size_t)blockIdx.x * k + i] = s_cnt[i];
  if (tid == 0) part_inertia[blockIdx.x] = s_in[0] + s_in[1] + s_in[2] + s_in[3];
}

}  // namespace meld

using namespace meld;

extern "C" int meld_kmeans_max_blocks(void) { return 1024; }

extern "C" int meld_kmeans_assign(const double* X, int64_t n, int d, const double* centroids, int k, int32_t* labels,
                                  double* part_sum, double* part_cnt, double* part_inertia, int n_blocks,
                                  meld_stream_t stream) {
  MELD_CHECK_ARG(X && centroids && labels && part_sum && part_cnt && part_inertia && n > 0 && d >= 1 && d <= KM_DMAX &&
                     k >= 1 && k <= KM_KMAX && n_blocks >= 1 && n_blocks <= 1024,
                 "meld_kmeans_assign: bad arguments (d <= %d, k <= %d)", KM_DMAX, KM_KMAX);
  hipLaunchKernelGGL(kmeans_assign_kernel, dim3(n_blocks), dim3(KM_THREADS), 0, S(stream), X, n, d, centroids, k, labels,
                     part_sum, part_cnt, part_inertia);
  MELD_LAUNCH_CHECK("kmeans_assign_kernel");
  return MELD_OK;
}
