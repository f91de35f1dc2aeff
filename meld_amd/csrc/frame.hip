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

// out[r][:] = (X[r][:] - mean) V,  V row-major [d][d] (column j = the j-th axis of the new frame), on v_mfma_f64_16x16x4_f64: a
// workgroup walks a contiguous range of 64-row tiles, wave w the rows 16 w .. 16 w + 15 of a tile against all the columns.  V sits
// in LDS (zero beyond d) and is read as B fragments (V[4 ks + lane / 16][16 cb + lane % 16]); the rows pass through LDS once on the
// way in (coalesced loads, centred; the loads of the NEXT tile are in flight while this one is multiplied) and once on the way out
// (coalesced stores).  D fragment: lane holds rows lane / 16 + 4 i (i = 0..3) of column lane % 16.  (A vector-FMA version with V in
// LDS was bound by its LDS reads, 14 per 13 FMAs; loads issued one by one behind a run-time trip count cost as much again: 1.1 ms at
// 1M x 50 against 0.15 ms of memory traffic.)
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int FR_KS = FR_DMAX / 4;   // K steps of 4
constexpr int FR_CB = FR_DMAX / 16;  // column blocks of 16
constexpr int FR_LD = FR_DMAX + 1;
constexpr int FR_NQ = FR_ROWS * FR_DMAX / 256;  // elements of a tile per thread
__global__ __launch_bounds__(256) void rotate_rows_kernel(const double* __restrict__ X, int64_t N, int d, const double* __restrict__ mean,
                                                          const double* __restrict__ V, double* __restrict__ out, int64_t tiles_per_wg) {
  __shared__ double sx[FR_ROWS][FR_LD];
  __shared__ double sv[FR_DMAX][FR_LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, lk = lane >> 4;
  const int ks_n = (d + 3) >> 2, cb_n = (d + 15) >> 4;
  for (int u = tid; u < FR_DMAX * FR_DMAX; u += 256) {
    const int k = u >> 6, c = u & 63;
    sv[k][c] = (k < d && c < d) ? V[k * d + c] : 0.0;
  }
  const int dk = 4 * ks_n;  // columns of the tile the MFMAs read (zero from d on)
  const double mk = (lane < d) ? mean[lane] : 0.0;  // (a thread always handles column tid % 64)
  const int64_t n_tiles = (N + FR_ROWS - 1) / FR_ROWS;
  const int64_t t_begin = (int64_t)blockIdx.x * tiles_per_wg, t_end = min(n_tiles, t_begin + tiles_per_wg);
  // (constant trip count: all the loads of a thread are in flight together)
  double xv[FR_NQ];
  auto fetch = [&](int64_t t) __attribute__((always_inline)) {
    const int64_t row0 = t * FR_ROWS;
#pragma unroll
    for (int q = 0; q < FR_NQ; ++q) {
      const int u = tid + 256 * q, rr = u >> 6, k = u & 63;
      xv[q] = (t < t_end && row0 + rr < N && k < d) ? X[(row0 + rr) * d + k] : mk;
    }
  };
  fetch(t_begin);
  for (int64_t t = t_begin; t < t_end; ++t) {
    const int64_t row0 = t * FR_ROWS;
    const int cnt = (int)min((int64_t)FR_ROWS, N - row0);
    __syncthreads();  // (V staged; the previous tile has left sx)
#pragma unroll
    for (int q = 0; q < FR_NQ; ++q) {
      const int u = tid + 256 * q, rr = u >> 6, k = u & 63;
      if (k < dk) sx[rr][k] = xv[q] - mk;  // (padding rows and columns: mk - mk = 0)
    }
    fetch(t + 1);
    __syncthreads();
    f64x4 acc[FR_CB];
#pragma unroll
    for (int cb = 0; cb < FR_CB; ++cb) acc[cb] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < FR_KS; ++ks) {
      if (ks < ks_n) {  // (uniform)
        const double a = sx[16 * w + l16][4 * ks + lk];
#pragma unroll
        for (int cb = 0; cb < FR_CB; ++cb)
          if (cb < cb_n) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sv[4 * ks + lk][16 * cb + l16], acc[cb], 0, 0, 0);
      }
    }
    __syncthreads();  // (every wave has read its rows)
#pragma unroll
    for (int cb = 0; cb < FR_CB; ++cb)
      if (cb < cb_n) {
#pragma unroll
        for (int i = 0; i < 4; ++i) sx[16 * w + lk + 4 * i][16 * cb + l16] = acc[cb][i];
      }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FR_NQ; ++q) {
      const int u = tid + 256 * q, rr = u >> 6, k = u & 63;
      if (rr < cnt && k < d) out[(row0 + rr) * d + k] = sx[rr][k];
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
  const int64_t per = ceil_div((int64_t)tiles, (int64_t)768);  // (three workgroups per CU by their LDS)
  hipLaunchKernelGGL(rotate_rows_kernel, dim3((unsigned)ceil_div((int64_t)tiles, per)), dim3(256), 0, S(stream), X, N, d, mean, V, out, per);
  MELD_LAUNCH_CHECK("rotate_rows_kernel");
  return MELD_OK;
}
