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

// out[r][j] = (X[r][:] - mean) . A_j,  A = the new axes as ROWS, At[d][FR_DMAX] zero-padded.  A workgroup walks a contiguous range
// of 64-row tiles; lane = row: every lane holds its (centred) row in registers, wave w computes a quarter of the columns, and the
// axes are read with SCALAR loads (constant address space, wave-uniform addresses: eight doubles per s_load_dwordx16) and enter the
// FMAs as scalar operands -- no LDS traffic for them at all.  The rows pass through LDS once on the way in (coalesced loads; the
// loads of the NEXT tile are in flight while this one is multiplied) and once on the way out (coalesced stores).
// (Measured on the way, 1M x 50: V in LDS read at wave-uniform addresses, 14 LDS reads per 13 FMAs: 1.1 ms; the same with the tile
// loads issued one by one behind a run-time trip count: 1.1 ms more; v_mfma_f64_16x16x4_f64 with V as B fragments: 0.93 ms, 0.75 of
// them the MFMAs -- the fp64 matrix instruction runs far below its nominal rate here; 0.19 ms is what the memory side takes; this
// version 0.69 ms: ~24 cycles per wave-wide fp64 FMA.  Round 6, the roles exchanged -- lane = column with its axis in registers,
// the ROWS through scalar registers (s_load_dwordx16 of the row, its doubles as the scalar operand of 8 K8 FMAs, no LDS, no barrier,
// the mean folded in as x . a - mean . a): correct, 0.7 ms again at d = 50 and 4.2 ms at d = 100 against 1.8 for the library GEMM --
// 400 MB through the scalar data caches arrive at ~0.6 TB/s; the same with a row across the lanes and v_readlane (two per double) in
// front of every FMA: 0.5 ms at d = 50, 2.9 at d = 100.  Not kept: 0.2 ms for a second kernel and a guard against cancellation.)
typedef double f64x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) f64x8* const_f64x8_ptr;
constexpr int FR_LD = FR_DMAX + 1;
constexpr int FR_NQ = FR_ROWS * FR_DMAX / 256;  // elements of a tile per thread
constexpr int FR_CPG = FR_DMAX / 4;             // columns per wave at most
__global__ __launch_bounds__(256) void rotate_rows_kernel(const double* __restrict__ X, int64_t N, int d, const double* __restrict__ mean,
                                                          const double* __restrict__ At, double* __restrict__ out, int64_t tiles_per_wg) {
  __shared__ double sx[FR_ROWS][FR_LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cpg = (d + 3) >> 2;  // columns of this wave: j0 .. j0 + cpg - 1
  const int j0 = w * cpg;
  const int k8n = (d + 7) >> 3;  // chunks of eight coordinates
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
    __syncthreads();  // (the previous tile has left sx)
#pragma unroll
    for (int q = 0; q < FR_NQ; ++q) {
      const int u = tid + 256 * q, rr = u >> 6, k = u & 63;
      sx[rr][k] = xv[q] - mk;  // (padding rows and columns: mk - mk = 0)
    }
    fetch(t + 1);
    __syncthreads();
    double xr[FR_DMAX];  // my row
#pragma unroll
    for (int k = 0; k < FR_DMAX; ++k) xr[k] = sx[lane][k];
    double res[FR_CPG];
    // four columns at a time: four independent FMA chains (one chain alone waits out the fp64 FMA latency at every step -- the
    // workgroup has one wave per SIMD; half rows in registers at three waves per SIMD spill and take 1.2 ms)
#pragma unroll
    for (int c4 = 0; c4 < FR_CPG; c4 += 4) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      if (c4 < cpg) {  // (uniform)
        const_f64x8_ptr arow[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)  // (columns beyond the wave's share or beyond d read a valid row; their result is dropped)
          arow[i] = (const_f64x8_ptr)(uintptr_t)(At + (size_t)min(j0 + c4 + i, d - 1) * FR_DMAX);
#pragma unroll
        for (int k8 = 0; k8 < FR_DMAX / 8; ++k8) {
          if (k8 < k8n) {  // (uniform; the axes are zero beyond d)
            f64x8 av[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = arow[i][k8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[i] = fma(xr[8 * k8 + e], av[i][e], acc[i]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) res[c4 + i] = acc[i];
    }
    __syncthreads();  // (every wave has read its rows)
#pragma unroll
    for (int c = 0; c < FR_CPG; ++c)
      if (c < cpg && j0 + c < d) sx[lane][j0 + c] = res[c];
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

// out[N][d] = (X - mean) A^T: row j of At[d][meld_frame_max_dims()] (zero-padded) = the j-th axis of the new frame; out must not alias X
extern "C" int meld_rotate_rows_f64(const double* X, int64_t N, int d, const double* mean, const double* At, double* out, meld_stream_t stream) {
  MELD_CHECK_ARG(X && mean && At && out && N > 0 && d > 0 && d <= FR_DMAX && X != out, "meld_rotate_rows_f64: bad arguments (d <= %d)", FR_DMAX);
  const int64_t tiles = ceil_div(N, FR_ROWS);
  const int64_t per = ceil_div(tiles, (int64_t)1024);
  hipLaunchKernelGGL(rotate_rows_kernel, dim3((unsigned)ceil_div(tiles, per)), dim3(256), 0, S(stream), X, N, d, mean, At, out, per);
  MELD_LAUNCH_CHECK("rotate_rows_kernel");
  return MELD_OK;
}
