// spmm.hip -- the graph-Laplacian operator: Chebyshev three-term recurrence step and helpers.
//
// Replaces [UPSTREAM pygsp filters.approximations.cheby_op] (called from reference
// meld/filter.py:59) and supplies the matrix-vector product for the lmax estimate
// ([UPSTREAM pygsp Graph.estimate_lmax], reference meld/filter.py:39).
//
// pygsp materialises factor = (2/a1)(L - a2 I) and runs  T_k = factor T_{k-1} - T_{k-2};
// r += c_k T_k  as separate scipy passes (SpMM, subtract, scale, add: >= 5 vector passes plus a
// second sparse matrix).  Here one kernel does
//     y = alpha * (dw .* x - W x) + beta * x + gamma * z ;   r += coef * y
// reading W (fp64 values + int32 columns) exactly once per step and never forming L or factor.
//
// Kernel shape ("CSR-stream"): a workgroup owns RB consecutive rows = one contiguous span of
// the CSR value/column arrays.  All 256 threads stream that span with fully coalesced loads,
// gather x[col] (16 B per nonzero for p = 2), and stage the products in LDS; then TPR = 256/RB
// lanes per row sum the row's LDS segment and a 2-step shuffle finishes the row.  Rows of any
// length work: the span is processed in LDS-sized chunks with the row accumulators kept in
// registers.  HBM-bound: 12 B/nonzero streamed + vector traffic (roofline section).
#include "common.hpp"

#include <cstdlib>

#include <algorithm>

namespace meld {

constexpr int DOT_SLOTS = 64;

template <int P>
struct Vec {
  double v[P];
};

template <int P>
__device__ __forceinline__ Vec<P> load_row(const double* __restrict__ base, int64_t row, int ld, int colofs) {
  Vec<P> out;
  const double* p = base + row * ld + colofs;
  if constexpr (P == 1) {
    out.v[0] = p[0];
  } else {
#pragma unroll
    for (int c = 0; c < P; c += 2) {
      const double2 t = *reinterpret_cast<const double2*>(p + c);
      out.v[c] = t.x;
      out.v[c + 1] = t.y;
    }
  }
  return out;
}

template <int P>
__device__ __forceinline__ void store_row(double* __restrict__ base, int64_t row, int ld, int colofs,
                                          const Vec<P>& val) {
  double* p = base + row * ld + colofs;
  if constexpr (P == 1) {
    p[0] = val.v[0];
  } else {
#pragma unroll
    for (int c = 0; c < P; c += 2) *reinterpret_cast<double2*>(p + c) = make_double2(val.v[c], val.v[c + 1]);
  }
}

#ifndef SPMM_U
#define SPMM_U 8  // (value, column, gather) chains in flight per thread (measured at 1M, P = 2: 2 -> 224, 4 -> 206, 8 -> 194 us)
#endif

template <int P, int RB>
__global__ __launch_bounds__(256) void cheby_step_kernel(
    const int64_t* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
    const double* __restrict__ dw, int64_t n_rows, int ld, int colofs, const double* __restrict__ x_full,
    int64_t x_row_offset, const double* z, double* y, double* r, double alpha, double beta, double gamma,
    double coef, double* __restrict__ dots, int chunk, const double* __restrict__ coef_dev) {
  // coef_dev (device-resident Lanczos): alpha and gamma come from device memory, written by the
  // previous iteration's scalar kernel, so that no host round trip separates the iterations
  if (coef_dev != nullptr) {
    alpha = coef_dev[3];
    gamma = coef_dev[4];
  }
  extern __shared__ __attribute__((aligned(16))) double prod[];  // [chunk][P]
  __shared__ int64_t s_rowptr[RB + 1];
  __shared__ double s_dot[2][4];
  constexpr int TPR = 256 / RB;
  const int tid = threadIdx.x;
  // XCD-contiguous row blocks: workgroup b is dispatched to XCD b % 8 (observed placement; used for
  // locality only), so XCD x takes the x-th contiguous eighth of the rows and its private 4 MiB L2
  // keeps the slice of the iterate its rows gather from.
  const int nb = gridDim.x;
  const int per = (nb + 7) >> 3;
  int64_t rb = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (rb >= nb) rb = nb;  // padding workgroup of the last XCD: owns no rows
  const int64_t row0 = min(rb * RB, n_rows);
  if (tid <= RB) s_rowptr[tid] = rowptr[min(row0 + tid, n_rows)];

  const int rr = tid / TPR;
  const int sub = tid % TPR;
  const int64_t row = row0 + rr;
  const bool own = (row < n_rows) && (sub == 0);

  // epilogue operands: issue the loads now, consume them after the reduction
  Vec<P> xl, zl, rl;
  double dwi = 0.0;
#pragma unroll
  for (int c = 0; c < P; ++c) xl.v[c] = zl.v[c] = rl.v[c] = 0.0;
  if (own) {
    xl = load_row<P>(x_full, x_row_offset + row, ld, colofs);
    dwi = dw[row];
    if (gamma != 0.0) zl = load_row<P>(z, row, ld, colofs);
    if (r != nullptr) rl = load_row<P>(r, row, ld, colofs);
  }
  __syncthreads();

  const int64_t e0 = s_rowptr[0];
  const int64_t e1 = s_rowptr[RB];
  const int64_t rs = s_rowptr[rr];
  const int64_t re = s_rowptr[rr + 1];
  Vec<P> acc;
#pragma unroll
  for (int c = 0; c < P; ++c) acc.v[c] = 0.0;

  for (int64_t cs = e0; cs < e1; cs += chunk) {
    const int64_t ce = min(cs + (int64_t)chunk, e1);
    // stream the span: SPMM_U independent (value, column, gather) chains per thread per trip
    for (int64_t eb = cs + tid; eb < ce; eb += SPMM_U * 256) {
      double v[SPMM_U];
      int j[SPMM_U];
#pragma unroll
      for (int u = 0; u < SPMM_U; ++u) {
        const int64_t e = eb + u * 256;
        const bool ok = e < ce;
        // streamed once: non-temporal loads keep the CSR arrays out of the L1 the gathers live in
        v[u] = ok ? __builtin_nontemporal_load(val + e) : 0.0;
        j[u] = ok ? __builtin_nontemporal_load(col + e) : 0;
      }
      Vec<P> xj[SPMM_U];
#pragma unroll
      for (int u = 0; u < SPMM_U; ++u) xj[u] = load_row<P>(x_full, j[u], ld, colofs);
#pragma unroll
      for (int u = 0; u < SPMM_U; ++u) {
        const int64_t e = eb + u * 256;
        if (e < ce) {
          Vec<P> pr;
#pragma unroll
          for (int c = 0; c < P; ++c) pr.v[c] = v[u] * xj[u].v[c];
          store_row<P>(prod, e - cs, P, 0, pr);
        }
      }
    }
    __syncthreads();
    const int64_t lo = max(rs, cs), hi = min(re, ce);
    for (int64_t e = lo + sub; e < hi; e += TPR) {
      const Vec<P> t = load_row<P>(prod, e - cs, P, 0);
#pragma unroll
      for (int c = 0; c < P; ++c) acc.v[c] += t.v[c];
    }
    if (ce < e1) __syncthreads();
  }

#pragma unroll
  for (int off = 1; off < TPR; off <<= 1) {
#pragma unroll
    for (int c = 0; c < P; ++c) acc.v[c] += __shfl_xor(acc.v[c], off, 64);
  }

  double d_yx = 0.0, d_yy = 0.0;
  if (own) {
    Vec<P> yv;
#pragma unroll
    for (int c = 0; c < P; ++c) {
      const double lx = dwi * xl.v[c] - acc.v[c];  // (L x)_i
      yv.v[c] = alpha * lx + beta * xl.v[c] + gamma * zl.v[c];
      rl.v[c] += coef * yv.v[c];
      d_yx += yv.v[c] * xl.v[c];
      d_yy += yv.v[c] * yv.v[c];
    }
    store_row<P>(y, row, ld, colofs, yv);
    if (r != nullptr) store_row<P>(r, row, ld, colofs, rl);
  }
  if (dots != nullptr) {
    d_yx = wave_sum(d_yx);
    d_yy = wave_sum(d_yy);
    if ((tid & 63) == 0) {
      s_dot[0][tid >> 6] = d_yx;
      s_dot[1][tid >> 6] = d_yy;
    }
    __syncthreads();
    if (tid < 2) {
      const double s = s_dot[tid][0] + s_dot[tid][1] + s_dot[tid][2] + s_dot[tid][3];
      atomicAdd(&dots[tid * DOT_SLOTS + (blockIdx.x % DOT_SLOTS)], s);
    }
  }
}

__global__ __launch_bounds__(256) void scale_kernel(const double* __restrict__ x, double a, double* __restrict__ r,
                                                    int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    r[i] = a * x[i];
}

__global__ __launch_bounds__(256) void axpby_kernel(double a, const double* __restrict__ x, double b,
                                                    double* __restrict__ y, int64_t n, double* __restrict__ nrm2,
                                                    const double* __restrict__ a_dev) {
  __shared__ double s_part[4];
  if (a_dev != nullptr) a = *a_dev;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = a * x[i] + b * y[i];
    y[i] = v;
    acc += v * v;
  }
  if (nrm2 != nullptr) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
      atomicAdd(&nrm2[blockIdx.x % DOT_SLOTS], s_part[0] + s_part[1] + s_part[2] + s_part[3]);
  }
}

__global__ __launch_bounds__(256) void normalize_rows_l1_kernel(const double* __restrict__ in,
                                                                double* __restrict__ out, int64_t n_rows, int p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  double s = 0.0;
  for (int c = 0; c < p; ++c) s += fabs(in[i * p + c]);
  if (s == 0.0) s = 1.0;
  for (int c = 0; c < p; ++c) out[i * p + c] = in[i * p + c] / s;
}

// Scalar steps of the device-resident Lanczos recurrence (one wave each).
// state: [0] s_cur (v_k = s_cur u_k), [1] s_prev, [2] beta_{k-1}, [3] alpha argument of the next
// SpMV (= s_cur), [4] its gamma argument (= -beta_{k-1} s_prev), [5] a of w = y + a u (= -alpha_k s_cur)
__global__ __launch_bounds__(64) void lanczos_alpha_kernel(double* __restrict__ state, const double* __restrict__ dots,
                                                           double* __restrict__ nrm2, double* __restrict__ alphas, int it) {
  double v = dots[threadIdx.x];  // <y, u> partial sums, DOT_SLOTS == 64
  v = wave_sum(v);
  nrm2[threadIdx.x] = 0.0;  // the axpby kernel that follows accumulates into these slots
  if (threadIdx.x == 0) {
    const double alpha = v * state[0];
    alphas[it] = alpha;
    state[5] = -alpha * state[0];
  }
}
// alpha step + axpy in one launch (the device-resident loops): every workgroup reduces the 64 partial sums of <y, u>
// itself (same order, same value), so the scalar kernel between the SpMV and the axpy -- a launch and its gap per
// iteration, 40 to 100 iterations per estimate -- is gone; workgroup 0 records alpha.  nrm2 must be zero on entry (the
// caller at the first iteration, lanczos_beta_kernel afterwards).
__global__ __launch_bounds__(256) void lanczos_axpy_fused_kernel(double* __restrict__ state, const double* __restrict__ dots,
                                                                 double* __restrict__ nrm2, double* __restrict__ alphas, int it,
                                                                 const double* __restrict__ x, double* __restrict__ y, int64_t n) {
  __shared__ double s_a;
  __shared__ double s_part[4];
  if (threadIdx.x < 64) {
    const double v = wave_sum(dots[threadIdx.x]);
    if (threadIdx.x == 0) {
      const double s_cur = state[0];
      const double alpha = v * s_cur;
      s_a = -alpha * s_cur;
      if (blockIdx.x == 0) {
        alphas[it] = alpha;
        state[5] = -alpha * s_cur;
      }
    }
  }
  __syncthreads();
  const double a = s_a;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = a * x[i] + y[i];
    y[i] = v;
    acc += v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&nrm2[blockIdx.x % DOT_SLOTS], s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}
// Device-resident loop on the tiled layout (meld_pt_lanczos_steps): no scalar kernel between the iterations.  The SpMV derives
// s_k and beta_{k-1} itself (PtLanczos, common.hpp); this is the axpy of lanczos_axpy_fused_kernel with the parity buffers of
// that scheme: it reads s_k from state_cur, <y, u> from dots_cur, adds |w|^2 into nrm2_cur (cleared by the SpMV before it) and
// clears dots_next, the slots the NEXT SpMV adds into.
__global__ __launch_bounds__(256) void lanczos_axpy_pp_kernel(const double* __restrict__ state_cur, const double* __restrict__ dots_cur,
                                                              double* __restrict__ nrm2_cur, double* __restrict__ dots_next,
                                                              double* __restrict__ alphas, int it, const double* __restrict__ x,
                                                              double* __restrict__ y, int64_t n, const int* __restrict__ stop) {
  __shared__ double s_a;
  __shared__ double s_part[4];
  if (stop != nullptr && *stop != 0) return;  // (uniform) stop request: see PtLanczos
  if (threadIdx.x < 64) {
    const double v = wave_sum(dots_cur[threadIdx.x]);
    if (threadIdx.x == 0) {
      const double s_cur = state_cur[0];
      const double alpha = v * s_cur;
      s_a = -alpha * s_cur;
      if (blockIdx.x == 0) alphas[it] = alpha;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 64 + 2 * DOT_SLOTS) dots_next[threadIdx.x - 64] = 0.0;
  __syncthreads();
  const double a = s_a;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = a * x[i] + y[i];
    y[i] = v;
    acc += v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&nrm2_cur[blockIdx.x % DOT_SLOTS], s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}
// first call of a run (it_begin == 0): the parity buffers as iteration 0 expects them -- |u_0|^2 = 1 / state[0]^2 in slot 0 of the
// "previous" sums, s_{-1} = 0, every accumulator clear
__global__ __launch_bounds__(256) void lanczos_pp_init_kernel(const double* __restrict__ state, double* __restrict__ pp) {
  // pp: dots[2][2 DOT_SLOTS] | nrm2[2][DOT_SLOTS] | st[2][8]
  const int t = threadIdx.x;
  for (int i = t; i < 6 * DOT_SLOTS + 16; i += 256) pp[i] = 0.0;
  __syncthreads();
  if (t == 0) {
    const double inv = state[0];
    pp[4 * DOT_SLOTS + DOT_SLOTS] = 1.0 / (inv * inv);  // nrm2[1][0]: what "iteration -1" left
  }
}
// end of a batch: beta of the last iteration (the next SpMV would have recorded it)
__global__ __launch_bounds__(64) void lanczos_pp_last_beta_kernel(const double* __restrict__ nrm2_last, double* __restrict__ betas, int it_last,
                                                                  const int* __restrict__ stop) {
  if (stop != nullptr && *stop != 0) return;
  const double v = wave_sum(nrm2_last[threadIdx.x]);
  if (threadIdx.x == 0) betas[it_last] = sqrt(v);
}
__global__ __launch_bounds__(64) void lanczos_beta_kernel(double* __restrict__ state, const double* __restrict__ nrm2,
                                                          double* __restrict__ dots, double* __restrict__ betas, int it,
                                                          double* __restrict__ nrm2_clear) {
  double v = nrm2[threadIdx.x];
  v = wave_sum(v);
  // (device-resident loops: read above by every lane, the next fused axpy accumulates into these slots)
  if (nrm2_clear != nullptr) nrm2_clear[threadIdx.x] = 0.0;
  dots[threadIdx.x] = 0.0;  // the next SpMV accumulates <y, u> and <y, y> into these slots
  dots[DOT_SLOTS + threadIdx.x] = 0.0;
  if (threadIdx.x == 0) {
    const double beta = sqrt(v);
    betas[it] = beta;
    const double s_cur = state[0];
    state[1] = s_cur;
    state[2] = beta;
    state[0] = 1.0 / beta;
    state[3] = 1.0 / beta;
    state[4] = -beta * s_cur;
  }
}

// One-reduction form of the iteration for the row-sharded driver.  With u_k = w_{k-1} kept un-normalised
// (v_k = u_k / n_k, n_k = |u_k| = beta_{k-1}) the SpMV z = L u_k needs no scalar, and both sums of an iteration --
// <z, u_k> (SpMV) and |u_k|^2 (accumulated by the axpy that formed u_k) -- sit in one buffer acc = dots (2 slots
// blocks) | nrm2 (1 block), so ONE all-reduce per iteration serves both:  alpha_k = <z, u_k> / |u_k|^2,
//   u_{k+1} = w_k = z / n_k - (alpha_k / n_k) u_k - (n_k / n_{k-1}) u_{k-1}.
// |w|^2 is still summed from the vector itself (the shortcut |y|^2 - alpha^2 is unstable); beta_{k-1} = n_k becomes
// known one iteration late, which only delays the convergence check by one iteration.
// state: [2] n_{k-1} -> n_k, [5] c1 = 1 / n_k, [6] c2 = -alpha_k / n_k, [7] c3 = -n_k / n_{k-1} (0 at it == 0)
__global__ __launch_bounds__(64) void lanczos_fold_kernel(double* __restrict__ state, double* __restrict__ acc,
                                                          double* __restrict__ alphas, double* __restrict__ betas, int it) {
  const double a = wave_sum(acc[threadIdx.x]);
  const double b = wave_sum(acc[2 * DOT_SLOTS + threadIdx.x]);
  acc[threadIdx.x] = 0.0;  // the next SpMV / axpy accumulate into these slots
  acc[DOT_SLOTS + threadIdx.x] = 0.0;
  acc[2 * DOT_SLOTS + threadIdx.x] = 0.0;
  if (threadIdx.x == 0) {
    const double n = sqrt(b), alpha = a / b, n_prev = state[2];
    alphas[it] = alpha;
    if (it > 0) betas[it - 1] = n;
    state[5] = 1.0 / n;
    state[6] = -alpha / n;
    state[7] = it > 0 ? -n / n_prev : 0.0;
    state[2] = n;
  }
}
__global__ __launch_bounds__(256) void lanczos_axpy3_kernel(double* __restrict__ y, const double* __restrict__ u,
                                                            const double* __restrict__ u_prev, int64_t n,
                                                            const double* __restrict__ state, double* __restrict__ nrm2) {
  __shared__ double s_part[4];
  const double c1 = state[5], c2 = state[6], c3 = state[7];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double v = c1 * y[i] + c2 * u[i];
    if (c3 != 0.0) v += c3 * u_prev[i];
    y[i] = v;
    acc += v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&nrm2[blockIdx.x % DOT_SLOTS], s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

template <int P, int RB>
static int launch_cheby(const int64_t* rowptr, const int32_t* col, const double* val, const double* dw, int64_t n_rows,
                        int ld, int colofs, const double* x_full, int64_t x_row_offset, const double* z, double* y,
                        double* r, double alpha, double beta, double gamma, double coef, double* dots, int chunk,
                        hipStream_t st, const double* coef_dev = nullptr) {
  const unsigned grid = (unsigned)(ceil_div(ceil_div(n_rows, RB), 8) * 8);  // multiple of 8: bijective XCD remap
  const size_t lds = sizeof(double) * (size_t)chunk * P;
  hipLaunchKernelGGL((cheby_step_kernel<P, RB>), dim3(grid), dim3(256), lds, st, rowptr, col, val, dw, n_rows, ld,
                     colofs, x_full, x_row_offset, z, y, r, alpha, beta, gamma, coef, dots, chunk, coef_dev);
  return 0;
}

// Wide signals (the probe block of the filter-bank VertexFrequencyCluster, p = 64 columns): lanes = COLUMNS.  A wave owns a row at
// a time; for every nonzero the 64 lanes read the 8 p contiguous bytes of the neighbour's row of the iterate (whole cache lines,
// one request per 128 bytes instead of one per 16) and the matrix is streamed once for all columns -- against p / 2 launches of
// the two-column kernel, each of which streams it again (3.4 ms per 64-column product at 1M cells).  The row's (value, column)
// pairs are loaded eight at a time by the first lanes and handed round with v_readlane (scalar address arithmetic, the value as a
// scalar operand of the FMA); eight gathers in flight per lane.  Rows of a workgroup are consecutive and workgroups are
// XCD-contiguous, so the rows of the iterate a neighbourhood shares are served by that XCD's L2.
//     y = alpha (dw .* x - W x) + beta x + gamma z      (y and z may alias: read before write per element)
#ifndef MELD_WIDE_U
#define MELD_WIDE_U 16
#endif
#ifndef MELD_WIDE_ROWS
#define MELD_WIDE_ROWS 8
#endif
constexpr int WIDE_U = MELD_WIDE_U;        // gathers in flight per lane
constexpr int WIDE_ROWS = MELD_WIDE_ROWS;  // rows per wave (consecutive)
__global__ __launch_bounds__(256) void cheby_step_wide_kernel(const int64_t* __restrict__ rowptr, const int* __restrict__ col,
                                                              const double* __restrict__ val, const double* __restrict__ dw,
                                                              int64_t n_rows, int p, const double* __restrict__ x_full,
                                                              int64_t x_row_offset, const double* z, double* y, double alpha,
                                                              double beta, double gamma) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nb = gridDim.x;
  const int per = (nb + 7) >> 3;
  int64_t rb = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (rb >= nb) return;
  const int64_t row_first = (rb * 4 + wv) * WIDE_ROWS;
  if (row_first >= n_rows) return;  // (uniform)
  const int n_here = (int)min((int64_t)WIDE_ROWS, n_rows - row_first);
  const bool on = lane < p;
  const int lc = on ? lane : 0;
  // The wave's WIDE_ROWS consecutive rows are ONE stream of entries [E0, E1): the (value, column) pairs are fetched a chunk ahead
  // of the gathers they feed and the gathers a chunk ahead of the FMAs that consume them, so the two dependent memory latencies
  // of a chunk overlap with the previous chunk's work and no bubble opens at a row boundary.  Row boundaries, degrees and the
  // rows' own operands are loaded once, up front (lane r: row r), and handed round with v_readlane.
  static_assert(WIDE_ROWS + 1 <= 64, "row pointers of a wave are held one per lane");
  const int64_t rp = rowptr[min(row_first + min(lane, n_here), n_rows)];
  const double dwl = dw[min(row_first + min(lane, n_here - 1), n_rows - 1)];
  double xi[WIDE_ROWS], zi[WIDE_ROWS];
#pragma unroll
  for (int r = 0; r < WIDE_ROWS; ++r) {
    const int64_t row = min(row_first + r, n_rows - 1);
    xi[r] = x_full[(x_row_offset + row) * p + lc];
    zi[r] = gamma != 0.0 ? __builtin_nontemporal_load(z + row * p + lc) : 0.0;  // (streamed once: leave the L2 to the gathered rows)
  }
  auto rp_at = [&](int r) __attribute__((always_inline)) {
    return (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(rp >> 32), r) << 32) |
                     (unsigned)__builtin_amdgcn_readlane((int)rp, r));
  };
  const int64_t E0 = rp_at(0), E1 = rp_at(n_here);
  auto fetch_pairs = [&](int64_t e0, double& vv, int& cc) __attribute__((always_inline)) {
    const int64_t e = min(e0 + (lane & (WIDE_U - 1)), E1 - 1);
    vv = (e0 + (lane & (WIDE_U - 1)) < E1) ? __builtin_nontemporal_load(val + max(e, (int64_t)0)) : 0.0;
    cc = E1 > E0 ? __builtin_nontemporal_load(col + max(e, (int64_t)0)) : 0;
  };
  auto gather = [&](int cc, double (&xj)[WIDE_U]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < WIDE_U; ++u) xj[u] = x_full[(int64_t)__builtin_amdgcn_readlane(cc, u) * p + lc];
  };
  int r_cur = 0;
  int64_t re_cur = rp_at(1);
  double acc = 0.0;
  auto flush_rows_until = [&](int64_t e) __attribute__((always_inline)) {  // (uniform) close every row that ends at or before entry e
    while (r_cur < n_here && e >= re_cur) {
      // xi / zi of row r_cur: a uniform select over the register array
      double xr = xi[0], zr = zi[0];
#pragma unroll
      for (int r = 1; r < WIDE_ROWS; ++r) {
        xr = (r == r_cur) ? xi[r] : xr;
        zr = (r == r_cur) ? zi[r] : zr;
      }
      const int dlo = __builtin_amdgcn_readlane(__double2loint(dwl), r_cur), dhi = __builtin_amdgcn_readlane(__double2hiint(dwl), r_cur);
      if (on) __builtin_nontemporal_store(alpha * (__hiloint2double(dhi, dlo) * xr - acc) + beta * xr + gamma * zr, y + (row_first + r_cur) * p + lane);
      acc = 0.0;
      ++r_cur;
      re_cur = r_cur < n_here ? rp_at(r_cur + 1) : E1 + 1;
    }
  };
  double vA, vB;
  int cA, cB;
  double xA[WIDE_U], xB[WIDE_U];
  fetch_pairs(E0, vA, cA);
  fetch_pairs(E0 + WIDE_U, vB, cB);
  gather(cA, xA);
  for (int64_t e0 = E0; e0 < E1; e0 += 2 * WIDE_U) {
    // chunk A = [e0, e0 + U), chunk B = [e0 + U, e0 + 2 U); the pairs of the chunk after B and B's gathers go out before A is summed
    gather(cB, xB);
    double vN;
    int cN;
    fetch_pairs(e0 + 2 * WIDE_U, vN, cN);
#pragma unroll
    for (int u = 0; u < WIDE_U; ++u) {
      flush_rows_until(e0 + u);
      const int lo = __builtin_amdgcn_readlane(__double2loint(vA), u), hi = __builtin_amdgcn_readlane(__double2hiint(vA), u);
      acc = fma(__hiloint2double(hi, lo), xA[u], acc);
    }
    gather(cN, xA);
    double vM;
    int cM;
    fetch_pairs(e0 + 3 * WIDE_U, vM, cM);
#pragma unroll
    for (int u = 0; u < WIDE_U; ++u) {
      flush_rows_until(e0 + WIDE_U + u);
      const int lo = __builtin_amdgcn_readlane(__double2loint(vB), u), hi = __builtin_amdgcn_readlane(__double2hiint(vB), u);
      acc = fma(__hiloint2double(hi, lo), xB[u], acc);
    }
    vA = vN;
    cA = cN;
    vB = vM;
    cB = cM;
  }
  flush_rows_until(E1 + 1);  // the rows that are left (the last one, and rows without entries)
}

// (Round 6, measured and not kept: the same step on PANELS of 16 columns -- four rows per wave, lane = (row group, column), one launch
// per panel so that the 128 MB of iterate a launch gathers from stay in the 256 MiB Infinity Cache, the matrix streamed `nt` once per
// panel.  Same results (1.9e-16), 4.89 ms per 64-column product against 1.89 ms for the kernel above.  The gathers move nnz x 8 B x p
// through the L2s whatever the panel width -- 20 GB per product, of which the L2s absorb half -- and what caps them is the L2 <-> fabric
// rate (~6.3 TB/s: the rate of a copy, wherever the bytes come from), not HBM: keeping the iterate in the Infinity Cache does not lift
// it, and four passes add 1.4 GB of matrix.  The lever is the L2 hit rate, i.e. the row order, as round 5's window measurements said.)
}  // namespace meld

using namespace meld;

// One step of the recurrence for a WIDE signal, 3 <= p <= 64 columns, row-major [rows, p] (lanes = columns; see
// cheby_step_wide_kernel).  Same operator as meld_cheby_step without the accumulator and the dot products:
//   y = alpha (dw .* x - W x) + beta x + gamma z;   x_full: the whole iterate [n_total, p], x_row_offset: first local row in it.
extern "C" int meld_cheby_step_wide(const int64_t* rowptr, const int32_t* col, const double* val, const double* dw, int64_t n_rows,
                                    int p, const double* x_full, int64_t x_row_offset, const double* z, double* y, double alpha,
                                    double beta, double gamma, meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && col && val && dw && x_full && y && n_rows >= 0 && p >= 1 && p <= 64, "meld_cheby_step_wide: bad arguments (1 <= p <= 64)");
  MELD_CHECK_ARG(gamma == 0.0 || z != nullptr, "meld_cheby_step_wide: z is required when gamma != 0");
  if (n_rows == 0) return MELD_OK;
  const int64_t nblk = ceil_div(n_rows, 4 * WIDE_ROWS);
  const unsigned grid = (unsigned)(ceil_div(nblk, 8) * 8);
  // (profiling hook, never set in production: MELD_WIDE_PADLDS=<bytes> of unused dynamic LDS lowers the workgroups per CU, i.e. the
  // rows in flight per XCD -- the window of the iterate its L2 has to hold)
  static const size_t pad_lds = [] { const char* e = meld_dev_getenv("MELD_WIDE_PADLDS"); return e ? (size_t)atoi(e) : (size_t)0; }();
  if (pad_lds > 65536) {
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cheby_step_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad_lds); done = true; }
  }
  hipLaunchKernelGGL(cheby_step_wide_kernel, dim3(grid), dim3(256), pad_lds, S(stream), rowptr, col, val, dw, n_rows, p, x_full, x_row_offset, z,
                     y, alpha, beta, gamma);
  MELD_LAUNCH_CHECK("cheby_step_wide_kernel");
  return MELD_OK;
}

extern "C" int meld_spmm_dot_slots(void) { return DOT_SLOTS; }

extern "C" int meld_cheby_step(const int64_t* rowptr, const int32_t* col, const double* val, const double* dw,
                               int64_t n_rows, int64_t nnz_hint, int p, const double* x_full, int64_t x_row_offset,
                               const double* z, double* y, double* r, double alpha, double beta, double gamma,
                               double coef, double* dots, meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && dw && x_full && y && n_rows >= 0 && p >= 1, "meld_cheby_step: bad arguments");
  MELD_CHECK_ARG(gamma == 0.0 || z != nullptr, "meld_cheby_step: z is required when gamma != 0");
  MELD_CHECK_ARG(dots == nullptr || p == 1, "meld_cheby_step: dots are only produced for p == 1");
  hipStream_t st = S(stream);
  if (dots) MELD_HIP_CALL(hipMemsetAsync(dots, 0, sizeof(double) * 2 * DOT_SLOTS, st));
  if (n_rows == 0) return MELD_OK;  // a rank of the row-sharded driver that owns no rows: nothing to compute, the
                                    // partial sums stay zero and the caller still takes part in every collective
  constexpr int RB = 32;
  // LDS chunk: ~1.3x the mean span of RB rows, multiple of 256, within [512, 3072] entries
  const double mean_span = (nnz_hint > 0) ? (double)nnz_hint / (double)n_rows * RB : 1024.0;
  auto pick_chunk = [&](int P) {
    int64_t c = (int64_t)(mean_span * 1.3) / 256 * 256 + 256;
    const int64_t cmax = (48 * 1024) / (8 * P) / 256 * 256;
    return (int)std::max<int64_t>(512, std::min<int64_t>(c, cmax));
  };
  int colofs = 0;
  if (p % 2 == 0) {
    while (colofs + 4 <= p) {
      launch_cheby<4, RB>(rowptr, col, val, dw, n_rows, p, colofs, x_full, x_row_offset, z, y, r, alpha, beta, gamma,
                          coef, nullptr, pick_chunk(4), st);
      colofs += 4;
    }
    if (colofs + 2 <= p) {
      launch_cheby<2, RB>(rowptr, col, val, dw, n_rows, p, colofs, x_full, x_row_offset, z, y, r, alpha, beta, gamma,
                          coef, nullptr, pick_chunk(2), st);
      colofs += 2;
    }
  } else {
    for (; colofs < p; ++colofs)
      launch_cheby<1, RB>(rowptr, col, val, dw, n_rows, p, colofs, x_full, x_row_offset, z, y, r, alpha, beta, gamma,
                          coef, dots, pick_chunk(1), st);
  }
  MELD_LAUNCH_CHECK("cheby_step_kernel");
  return MELD_OK;
}

// n_iter iterations of the Lanczos recurrence of L = diag(dw) - W with every scalar on the device.
// Replaces the per-iteration host round trips of the lmax estimate ([UPSTREAM pygsp
// Graph.estimate_lmax], reference meld/filter.py:39): one SpMV, one wave-sized scalar kernel, one
// axpby, one scalar kernel per iteration, nothing read back until the caller looks at alphas/betas.
extern "C" int meld_lanczos_steps(const int64_t* rowptr, const int32_t* col, const double* val, const double* dw,
                                  int64_t n_rows, int64_t nnz_hint, double* v0, double* v1, double* v2, double* state,
                                  double* alphas, double* betas, int it_begin, int n_iter, double* scratch,
                                  meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && dw && v0 && v1 && v2 && state && alphas && betas && scratch && n_rows > 0 && it_begin >= 0 &&
                     n_iter >= 0,
                 "meld_lanczos_steps: bad arguments");
  hipStream_t st = S(stream);
  double* V[3] = {v0, v1, v2};
  double* dots = scratch;                  // 2 * DOT_SLOTS, zero on entry of iteration 0 (caller) and re-zeroed by the beta kernel
  double* nrm2 = scratch + 2 * DOT_SLOTS;  // DOT_SLOTS
  constexpr int RB = 32;
  const double mean_span = (nnz_hint > 0) ? (double)nnz_hint / (double)n_rows * RB : 1024.0;
  const int chunk = (int)std::max<int64_t>(512, std::min<int64_t>((int64_t)(mean_span * 1.3) / 256 * 256 + 256, (48 * 1024) / 8 / 256 * 256));
  const unsigned grid_ax = (unsigned)std::min<int64_t>(2048, ceil_div(n_rows, 256));
  for (int it = it_begin; it < it_begin + n_iter; ++it) {
    // roles rotate with the iteration: u_prev = V[it % 3], u = V[(it + 1) % 3], y = V[(it + 2) % 3]
    double* u_prev = V[it % 3];
    double* u = V[(it + 1) % 3];
    double* y = V[(it + 2) % 3];
    // y = s_cur L u - beta_{k-1} s_prev u_prev ;  dots <- <y, u>
    launch_cheby<1, RB>(rowptr, col, val, dw, n_rows, 1, 0, u, 0, u_prev, y, nullptr, 0.0, 0.0, 0.0, 0.0, dots, chunk, st,
                        state);
    // alpha_k = s_cur <y, u>;  w = y - alpha v_k (in y) ;  nrm2 <- |w|^2
    // (the beta step stays a one-wave launch of its own: folded into the axpy behind a last-workgroup ticket it needs a
    // __threadfence per workgroup, which on this part writes the L2 back -- measured 4.9 vs 3.6 ms per estimate)
    hipLaunchKernelGGL(lanczos_axpy_fused_kernel, dim3(grid_ax), dim3(256), 0, st, state, dots, nrm2, alphas, it, u, y, n_rows);
    hipLaunchKernelGGL(lanczos_beta_kernel, dim3(1), dim3(64), 0, st, state, nrm2, dots, betas, it, nrm2);
  }
  MELD_LAUNCH_CHECK("meld_lanczos_steps");
  return MELD_OK;
}

// The four phases of one device-resident Lanczos iteration as separate entry points, for the row-sharded
// driver: it interleaves them with the all-reduces of the partial sums (dots after the SpMV, nrm2 after the
// axpy) and the all-gather of the new vector, all stream-ordered, so that a sharded iteration needs no host
// round trip either.  x_full [n_total] is the gathered iterate, the other vectors are the local rows.
extern "C" int meld_lanczos_spmv(const int64_t* rowptr, const int32_t* col, const double* val, const double* dw,
                                 int64_t n_rows, int64_t nnz_hint, const double* x_full, int64_t x_row_offset,
                                 const double* z_local, double* y_local, const double* state, double* dots,
                                 meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && dw && x_full && z_local && y_local && state && dots && n_rows >= 0, "meld_lanczos_spmv: bad arguments");
  if (n_rows == 0) return MELD_OK;  // empty shard (the beta phase has zeroed the partial sums)
  constexpr int RB = 32;
  const double mean_span = (nnz_hint > 0) ? (double)nnz_hint / (double)n_rows * RB : 1024.0;
  const int chunk = (int)std::max<int64_t>(512, std::min<int64_t>((int64_t)(mean_span * 1.3) / 256 * 256 + 256, (48 * 1024) / 8 / 256 * 256));
  launch_cheby<1, RB>(rowptr, col, val, dw, n_rows, 1, 0, x_full, x_row_offset, z_local, y_local, nullptr, 0.0, 0.0, 0.0, 0.0,
                      dots, chunk, S(stream), state);
  MELD_LAUNCH_CHECK("meld_lanczos_spmv");
  return MELD_OK;
}
// The same two drivers on the panel-tiled layout (spmm_tiled.hip).
extern "C" int meld_pt_lanczos_steps(const meld_pt_layout_t* layout, const int64_t* rowptr, const double* dw, int64_t n_rows,
                                     double* v0, double* v1, double* v2, double* state, double* alphas, double* betas,
                                     int it_begin, int n_iter, double* scratch, const int32_t* stop, meld_stream_t stream) {
  MELD_CHECK_ARG(layout && rowptr && dw && v0 && v1 && v2 && state && alphas && betas && scratch && n_rows > 0 &&
                     it_begin >= 0 && n_iter >= 0,
                 "meld_pt_lanczos_steps: bad arguments");
  hipStream_t st = S(stream);
  double* V[3] = {v0, v1, v2};
  // Two launches per iteration: the SpMV (which derives its scalars from the previous axpy's partial sums and records beta of the
  // previous iteration, PtLanczos in common.hpp) and the axpy.  (The one-wave beta kernel that used to sit between them cost a
  // launch and its gaps, ~7 us of ~90 per iteration; folding it into the axpy behind a last-workgroup ticket had cost more than
  // it saved -- a device-scope release per workgroup.)  Parity buffers in scratch (8 DOT_SLOTS doubles):
  //   dots[2][2 DOT_SLOTS] | nrm2[2][DOT_SLOTS] | st[2][8];  iteration k adds <y, u> into dots[k & 1] and |w|^2 into nrm2[k & 1].
  double* dots_pp = scratch;
  double* nrm2_pp = scratch + 4 * DOT_SLOTS;
  double* st_pp = scratch + 6 * DOT_SLOTS;
  if (it_begin == 0) hipLaunchKernelGGL(lanczos_pp_init_kernel, dim3(1), dim3(256), 0, st, state, scratch);
  const unsigned grid_ax = (unsigned)std::min<int64_t>(2048, ceil_div(n_rows, 256));
  for (int it = it_begin; it < it_begin + n_iter; ++it) {
    double* u_prev = V[it % 3];
    double* u = V[(it + 1) % 3];
    double* y = V[(it + 2) % 3];
    const int cur = it & 1, prv = cur ^ 1;
    PtLanczos lz{nrm2_pp + prv * DOT_SLOTS, nrm2_pp + cur * DOT_SLOTS, st_pp + prv * 8, st_pp + cur * 8, betas, it, stop};
    const int rc = pt_step(layout, rowptr, dw, 1, u, 0, u_prev, y, nullptr, 0.0, 0.0, 0.0, 0.0, dots_pp + cur * 2 * DOT_SLOTS, nullptr, st, 0.0, &lz);
    if (rc != MELD_OK) return rc;
    hipLaunchKernelGGL(lanczos_axpy_pp_kernel, dim3(grid_ax), dim3(256), 0, st, st_pp + cur * 8, dots_pp + cur * 2 * DOT_SLOTS,
                       nrm2_pp + cur * DOT_SLOTS, dots_pp + prv * 2 * DOT_SLOTS, alphas, it, u, y, n_rows, stop);
  }
  if (n_iter > 0) {
    const int last = it_begin + n_iter - 1;
    hipLaunchKernelGGL(lanczos_pp_last_beta_kernel, dim3(1), dim3(64), 0, st, nrm2_pp + (last & 1) * DOT_SLOTS, betas, last, stop);
  }
  MELD_LAUNCH_CHECK("meld_pt_lanczos_steps");
  return MELD_OK;
}

extern "C" int meld_pt_lanczos_spmv(const meld_pt_layout_t* layout, const int64_t* rowptr, const double* dw, int64_t n_rows,
                                    const double* x_full, int64_t x_row_offset, const double* z_local, double* y_local,
                                    const double* state, double* dots, meld_stream_t stream) {
  MELD_CHECK_ARG(layout && rowptr && dw && x_full && z_local && y_local && state && dots && n_rows >= 0,
                 "meld_pt_lanczos_spmv: bad arguments");
  if (n_rows == 0) return MELD_OK;
  const int rc = pt_step(layout, rowptr, dw, 1, x_full, x_row_offset, z_local, y_local, nullptr, 0.0, 0.0, 0.0, 0.0, dots,
                         state, S(stream));
  if (rc != MELD_OK) return rc;
  MELD_LAUNCH_CHECK("meld_pt_lanczos_spmv");
  return MELD_OK;
}
extern "C" int meld_lanczos_alpha(double* state, const double* dots, double* nrm2, double* alphas, int it,
                                  meld_stream_t stream) {
  MELD_CHECK_ARG(state && dots && nrm2 && alphas && it >= 0, "meld_lanczos_alpha: bad arguments");
  hipLaunchKernelGGL(lanczos_alpha_kernel, dim3(1), dim3(64), 0, S(stream), state, dots, nrm2, alphas, it);
  MELD_LAUNCH_CHECK("lanczos_alpha_kernel");
  return MELD_OK;
}
extern "C" int meld_lanczos_axpy(const double* x_local, double* y_local, int64_t n_rows, const double* state, double* nrm2,
                                 meld_stream_t stream) {
  MELD_CHECK_ARG(x_local && y_local && state && nrm2 && n_rows >= 0, "meld_lanczos_axpy: bad arguments");
  if (n_rows == 0) return MELD_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(2048, ceil_div(n_rows, 256));
  hipLaunchKernelGGL(axpby_kernel, dim3(grid), dim3(256), 0, S(stream), 0.0, x_local, 1.0, y_local, n_rows, nrm2, state + 5);
  MELD_LAUNCH_CHECK("axpby_kernel");
  return MELD_OK;
}
extern "C" int meld_lanczos_beta(double* state, const double* nrm2, double* dots, double* betas, int it,
                                 meld_stream_t stream) {
  MELD_CHECK_ARG(state && nrm2 && dots && betas && it >= 0, "meld_lanczos_beta: bad arguments");
  hipLaunchKernelGGL(lanczos_beta_kernel, dim3(1), dim3(64), 0, S(stream), state, nrm2, dots, betas, it, (double*)nullptr);
  MELD_LAUNCH_CHECK("lanczos_beta_kernel");
  return MELD_OK;
}

// One-reduction Lanczos iteration (row-sharded driver): acc = [3 * slots] = <z, u> | (unused) | |u|^2 partial sums, summed
// over the ranks by ONE all-reduce between meld_lanczos_spmv (state[3] = 1, state[4] = 0: z = L u) and this call;
// meld_lanczos_axpy3 then forms u_{k+1} in place of z and accumulates |u_{k+1}|^2 into acc + 2 * slots.
extern "C" int meld_lanczos_fold(double* state, double* acc, double* alphas, double* betas, int it, meld_stream_t stream) {
  MELD_CHECK_ARG(state && acc && alphas && betas && it >= 0, "meld_lanczos_fold: bad arguments");
  hipLaunchKernelGGL(lanczos_fold_kernel, dim3(1), dim3(64), 0, S(stream), state, acc, alphas, betas, it);
  MELD_LAUNCH_CHECK("lanczos_fold_kernel");
  return MELD_OK;
}
extern "C" int meld_lanczos_axpy3(double* y_local, const double* u_local, const double* u_prev_local, int64_t n_rows,
                                  const double* state, double* nrm2, meld_stream_t stream) {
  MELD_CHECK_ARG(y_local && u_local && u_prev_local && state && nrm2 && n_rows >= 0, "meld_lanczos_axpy3: bad arguments");
  if (n_rows == 0) return MELD_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(2048, ceil_div(n_rows, 256));
  hipLaunchKernelGGL(lanczos_axpy3_kernel, dim3(grid), dim3(256), 0, S(stream), y_local, u_local, u_prev_local, n_rows, state, nrm2);
  MELD_LAUNCH_CHECK("lanczos_axpy3_kernel");
  return MELD_OK;
}

extern "C" int meld_scale_f64(const double* x, double a, double* r, int64_t n, meld_stream_t stream) {
  MELD_CHECK_ARG(x && r && n >= 0, "meld_scale_f64: bad arguments");
  if (n == 0) return MELD_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(4096, ceil_div(n, 256));
  hipLaunchKernelGGL(scale_kernel, dim3(grid), dim3(256), 0, S(stream), x, a, r, n);
  MELD_LAUNCH_CHECK("scale_kernel");
  return MELD_OK;
}

extern "C" int meld_axpby_f64(double a, const double* x, double b, double* y, int64_t n, double* nrm2,
                              meld_stream_t stream) {
  MELD_CHECK_ARG(x && y && n >= 0, "meld_axpby_f64: bad arguments");
  if (nrm2) MELD_HIP_CALL(hipMemsetAsync(nrm2, 0, sizeof(double) * DOT_SLOTS, S(stream)));
  if (n == 0) return MELD_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(2048, ceil_div(n, 256));
  hipLaunchKernelGGL(axpby_kernel, dim3(grid), dim3(256), 0, S(stream), a, x, b, y, n, nrm2, (const double*)nullptr);
  MELD_LAUNCH_CHECK("axpby_kernel");
  return MELD_OK;
}

extern "C" int meld_normalize_rows_l1(const double* in, double* out, int64_t n_rows, int p, meld_stream_t stream) {
  MELD_CHECK_ARG(in && out && n_rows > 0 && p > 0, "meld_normalize_rows_l1: bad arguments");
  hipLaunchKernelGGL(normalize_rows_l1_kernel, dim3((unsigned)ceil_div(n_rows, 256)), dim3(256), 0, S(stream), in, out,
                     n_rows, p);
  MELD_LAUNCH_CHECK("normalize_rows_l1_kernel");
  return MELD_OK;
}
