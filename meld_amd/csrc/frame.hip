// frame.hip -- the cells' principal frame for the candidate search (no reference counterpart).
//
// The list-driven first pass of the search (knn16.hip, EE) drops a block of 32 references behind its first K block when no
// partial distance is within reach of its row; that pays when the leading coordinates carry the distances.  The host
// (meld_amd/graph.py, HipOps.principal_frame) therefore hands the search the cells as (X - mean) V, V = the eigenvectors of
// the covariance in descending order of variance: distances are unchanged (V orthonormal), the candidates the search
// nominates are evaluated in fp64 on X itself.  Two kernels: the covariance of evenly spaced rows (a 65536 x 50 product
// through the library took 3.6 ms, a fifth of the search it is meant to shorten) and the rotation of all rows (0.7 ms as a
// library GEMM of this shape).  The 50 x 50 eigenproblem is solved on the host.
#include "common.hpp"

namespace meld {

constexpr int FR_ROWS = 64;   // rows per tile
constexpr int FR_DMAX = 64;   // largest d (V and a tile of rows in LDS)

// out[i * d + j] += sum over the sampled rows r = 0, stride, 2 stride, ... of (X[r][i] - mean[i]) (X[r][j] - mean[j])
__global__ __launch_bounds__(256) void cov_sample_kernel(const double* __restrict__ X, int64_t n_s, int64_t stride, int d,
                                                         const double* __restrict__ mean, double* __restrict__ out) {
  __shared__ double sx[FR_ROWS][FR_DMAX + 1];
  const int tid = threadIdx.x;
  const int64_t s0 = (int64_t)blockIdx.x * FR_ROWS;
  const int cnt = (int)min((int64_t)FR_ROWS, n_s - s0);
  for (int u = tid; u < FR_ROWS * d; u += 256) {
    const int r = u / d, k = u - r * d;
    sx[r][k] = r < cnt ? X[(s0 + r) * stride * d + k] - mean[k] : 0.0;
  }
  __syncthreads();
  for (int p = tid; p < d * d; p += 256) {
    const int i = p / d, j = p - i * d;
    if (j < i) continue;  // (the upper triangle; the host mirrors it)
    double acc = 0.0;
    for (int r = 0; r < FR_ROWS; ++r) acc = fma(sx[r][i], sx[r][j], acc);
    atomicAdd(out + p, acc);
  }
}

// out[r][:] = (X[r][:] - mean) V,  V row-major [d][d] (column j = the j-th axis of the new frame).  One workgroup per 64 rows:
// the rows and V in LDS, thread (row, quarter of the columns) -- V is read at wave-uniform addresses -- and the tile leaves
// through LDS with coalesced stores.
__global__ __launch_bounds__(256) void rotate_rows_kernel(const double* __restrict__ X, int64_t N, int d, const double* __restrict__ mean,
                                                          const double* __restrict__ V, double* __restrict__ out) {
  __shared__ double sv[FR_DMAX * FR_DMAX];
  __shared__ double sx[FR_ROWS][FR_DMAX + 1];
  const int tid = threadIdx.x;
  const int r = tid & 63, g = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int u = tid; u < d * d; u += 256) sv[u] = V[u];
  constexpr int CPG = FR_DMAX / 4;  // columns per thread at most
  const int cpg = (d + 3) >> 2;
  const int j0 = g * cpg;
  for (int64_t t = blockIdx.x; t * FR_ROWS < N; t += gridDim.x) {
    const int64_t row0 = t * FR_ROWS;
    const int cnt = (int)min((int64_t)FR_ROWS, N - row0);
    __syncthreads();  // (V staged; the previous tile has left sx)
    for (int u = tid; u < cnt * d; u += 256) {
      const int rr = u / d, k = u - rr * d;
      sx[rr][k] = X[row0 * d + u] - mean[k];
    }
    __syncthreads();
    double acc[CPG];
#pragma unroll
    for (int c = 0; c < CPG; ++c) acc[c] = 0.0;
    for (int k = 0; k < d; ++k) {
      const double xv = sx[r][k];
      const double* vk = sv + k * d + j0;
#pragma unroll
      for (int c = 0; c < CPG; ++c)
        if (c < cpg && j0 + c < d) acc[c] = fma(xv, vk[c], acc[c]);
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPG; ++c)
      if (c < cpg && j0 + c < d) sx[r][j0 + c] = acc[c];
    __syncthreads();
    for (int u = tid; u < cnt * d; u += 256) {
      const int rr = u / d, k = u - rr * d;
      out[row0 * d + u] = sx[rr][k];
    }
  }
}

}  // namespace meld

using namespace meld;

extern "C" int meld_frame_max_dims(void) { return FR_DMAX; }

// cov[d * d] (upper triangle, row-major; the caller zeroes it and mirrors the result) += the scatter matrix of the rows
// 0, stride, 2 stride, ... (n_s of them) of X about `mean`
extern "C" int meld_cov_sample_f64(const double* X, int64_t N, int d, const double* mean, int64_t stride, double* cov, meld_stream_t stream) {
  MELD_CHECK_ARG(X && mean && cov && N > 0 && d > 0 && d <= FR_DMAX && stride >= 1, "meld_cov_sample_f64: bad arguments (d <= %d)", FR_DMAX);
  const int64_t n_s = (N + stride - 1) / stride;
  hipLaunchKernelGGL(cov_sample_kernel, dim3((unsigned)ceil_div(n_s, FR_ROWS)), dim3(256), 0, S(stream), X, n_s, stride, d, mean, cov);
  MELD_LAUNCH_CHECK("cov_sample_kernel");
  return MELD_OK;
}

// out[N][d] = (X - mean) V   (V row-major [d][d]; out must not alias X)
extern "C" int meld_rotate_rows_f64(const double* X, int64_t N, int d, const double* mean, const double* V, double* out, meld_stream_t stream) {
  MELD_CHECK_ARG(X && mean && V && out && N > 0 && d > 0 && d <= FR_DMAX && X != out, "meld_rotate_rows_f64: bad arguments (d <= %d)", FR_DMAX);
  const unsigned tiles = (unsigned)ceil_div(N, FR_ROWS);
  hipLaunchKernelGGL(rotate_rows_kernel, dim3(tiles < 4096u ? tiles : 4096u), dim3(256), 0, S(stream), X, N, d, mean, V, out);
  MELD_LAUNCH_CHECK("rotate_rows_kernel");
  return MELD_OK;
}
