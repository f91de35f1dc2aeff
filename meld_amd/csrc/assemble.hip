// assemble.hip -- directed kernel rows -> symmetric weight matrix W (CSR) and degrees.
//
// Replaces, for the sparse kNN graph, [UPSTREAM graphtools BaseGraph.symmetrize_kernel
// (K + K^T)/2, BaseGraph.apply_anisotropy K_ij/(d_i d_j)^a, PyGSPGraph._build_weight_from_kernel
// (zero the diagonal)] and [UPSTREAM pygsp Graph.compute_laplacian: dw = W 1] -- the graph
// construction that reference meld/meld.py:117-118,273 delegates to graphtools.
//
// Method: every kept directed entry (i, j, v) is emitted twice as COO -- (i,j,v/2) and (j,i,v/2)
// -- with key = (row << 32) | col; one radix sort + one reduce-by-key gives (K + K^T)/2 with
// sorted, duplicate-free rows.  The diagonal never enters the COO stream: K_ii = 1 always
// (d_ii = 0), so it is carried analytically (row sum = 1 + off-diagonal sum; W has no diagonal).
// The sort / scan / reduce-by-key primitives are rocPRIM device algorithms compiled into this
// library; they are graph assembly (run once per fit), not the per-step hot loop.
#include "common.hpp"

#include <rocprim/rocprim.hpp>

namespace meld {

// wave per complete row: compact the kept candidates into the row's COO slots
__global__ __launch_bounds__(256) void coo_emit_rows_kernel(int64_t q_begin, int64_t q_count,
                                                            const int* __restrict__ cand_idx,
                                                            const double* __restrict__ cand_val, int ksel, int cap,
                                                            const int64_t* __restrict__ keep_off, int64_t M,
                                                            unsigned long long* __restrict__ keys,
                                                            double* __restrict__ vals) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= q_count) return;
  const int64_t base = keep_off[q];
  if (keep_off[q + 1] == base) return;  // flagged or empty row
  const unsigned long long gi = (unsigned long long)(q_begin + q);
  int written = 0;
  for (int c0 = 0; c0 < ksel; c0 += 64) {
    const int c = c0 + lane;
    double v = 0.0;
    int j = 0;
    if (c < ksel) {
      v = cand_val[(size_t)q * ksel + c];
      j = cand_idx[(size_t)q * cap + c];
    }
    const bool keep = v > 0.0;
    const unsigned long long b = __ballot(keep);
    if (keep) {
      const int64_t slot = base + written + __popcll(b & ((1ull << lane) - 1ull));
      const double hv = 0.5 * v;
      keys[slot] = (gi << 32) | (unsigned long long)(unsigned)j;
      vals[slot] = hv;
      keys[M + slot] = ((unsigned long long)(unsigned)j << 32) | gi;
      vals[M + slot] = hv;
    }
    written += __popcll(b);
  }
}

// thread per fallback entry
__global__ __launch_bounds__(256) void coo_emit_fallback_kernel(int64_t q_begin, const int* __restrict__ flag_rows,
                                                                int n_flag, const int64_t* __restrict__ fb_off,
                                                                const int* __restrict__ fb_col,
                                                                const double* __restrict__ fb_val, int64_t fb_total,
                                                                int64_t fb_base, int64_t M,
                                                                unsigned long long* __restrict__ keys,
                                                                double* __restrict__ vals) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= fb_total) return;
  // f = last index with fb_off[f] <= e
  int lo = 0, hi = n_flag;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (fb_off[mid] <= e)
      lo = mid;
    else
      hi = mid;
  }
  const unsigned long long gi = (unsigned long long)(q_begin + flag_rows[lo]);
  const unsigned long long j = (unsigned long long)(unsigned)fb_col[e];
  const double hv = 0.5 * fb_val[e];
  const int64_t slot = fb_base + e;
  keys[slot] = (gi << 32) | j;
  vals[slot] = hv;
  keys[M + slot] = (j << 32) | gi;
  vals[M + slot] = hv;
}

__global__ __launch_bounds__(256) void csr_rowptr_kernel(const unsigned long long* __restrict__ ukeys, int64_t nnz,
                                                         int64_t row_begin, int64_t n_rows,
                                                         int64_t* __restrict__ rowptr) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_rows) return;
  const unsigned long long target = (unsigned long long)(row_begin + r) << 32;
  int64_t lo = 0, hi = nnz;  // first index with key >= target
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (ukeys[mid] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  rowptr[r] = lo;
}

__global__ __launch_bounds__(256) void csr_cols_kernel(const unsigned long long* __restrict__ ukeys, int64_t nnz,
                                                       int* __restrict__ col) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < nnz) col[e] = (int)(ukeys[e] & 0xffffffffull);
}

// 8 lanes per row
__global__ __launch_bounds__(256) void csr_row_sums_kernel(const int64_t* __restrict__ rowptr,
                                                           const double* __restrict__ val, int64_t n_rows,
                                                           double diag, double* __restrict__ out) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int g = threadIdx.x & 7;
  double s = 0.0;
  if (r < n_rows) {
    const int64_t b = rowptr[r], e = rowptr[r + 1];
    for (int64_t k = b + g; k < e; k += 8) s += val[k];
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (r < n_rows && g == 0) out[r] = diag + s;
}

__global__ __launch_bounds__(256) void csr_anisotropy_kernel(const int64_t* __restrict__ rowptr,
                                                             const int* __restrict__ col, double* __restrict__ val,
                                                             int64_t n_rows, const double* __restrict__ ksum_all,
                                                             int64_t row_off, double anisotropy) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int g = threadIdx.x & 7;
  if (r >= n_rows) return;
  const double di = ksum_all[row_off + r];
  const int64_t b = rowptr[r], e = rowptr[r + 1];
  for (int64_t k = b + g; k < e; k += 8) {
    const double dd = di * ksum_all[col[k]];
    val[k] = (anisotropy == 1.0) ? val[k] / dd : val[k] / pow(dd, anisotropy);
  }
}

}  // namespace meld

using namespace meld;

extern "C" size_t meld_scan_temp_bytes(int64_t n) {
  size_t bytes = 0;
  int32_t* in = nullptr;
  int64_t* out = nullptr;
  (void)rocprim::exclusive_scan(nullptr, bytes, in, out, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>());
  return bytes + 256;
}

// out[0..n] = exclusive scan of in[0..n-1] with out[n] = total.  `in` must have n+1 readable
// entries? No: we scan n+1 items through a transform iterator that yields 0 for the last one.
namespace {
struct PadZero {
  const int32_t* p;
  int64_t n;
  __host__ __device__ int64_t operator()(int64_t i) const { return i < n ? (int64_t)p[i] : 0; }
};
}  // namespace

extern "C" int meld_exclusive_scan_i32_i64(const int32_t* in, int64_t* out, int64_t n, void* temp, size_t temp_bytes,
                                           meld_stream_t stream) {
  MELD_CHECK_ARG(in && out && temp && n >= 0, "meld_exclusive_scan_i32_i64: bad arguments");
  auto it = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int64_t>(0), PadZero{in, n});
  size_t bytes = temp_bytes;
  MELD_HIP_CALL(rocprim::exclusive_scan(temp, bytes, it, out, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(),
                                        S(stream)));
  return MELD_OK;
}

extern "C" int meld_coo_emit(int64_t q_begin, int64_t q_count, const int32_t* cand_idx, const double* cand_val,
                             const int32_t* cand_cnt, int ksel, int cap, const int64_t* keep_off,
                             const int32_t* flag_rows,
                             int32_t n_flag, const int64_t* fb_off, const int32_t* fb_col, const double* fb_val,
                             int64_t fb_base, int64_t M, uint64_t* keys, double* vals, meld_stream_t stream) {
  (void)cand_cnt;
  MELD_CHECK_ARG(cand_idx && cand_val && keep_off && keys && vals && q_count > 0, "meld_coo_emit: bad arguments");
  MELD_CHECK_ARG(cap >= ksel, "meld_coo_emit: row stride cap=%d smaller than ksel=%d", cap, ksel);
  hipLaunchKernelGGL(coo_emit_rows_kernel, dim3((unsigned)ceil_div(q_count, 4)), dim3(256), 0, S(stream), q_begin,
                     q_count, cand_idx, cand_val, ksel, cap, keep_off, M,
                     reinterpret_cast<unsigned long long*>(keys), vals);
  MELD_LAUNCH_CHECK("coo_emit_rows_kernel");
  const int64_t fb_total = M - fb_base;
  if (n_flag > 0 && fb_total > 0) {
    MELD_CHECK_ARG(flag_rows && fb_off && fb_col && fb_val, "meld_coo_emit: missing fallback arrays");
    hipLaunchKernelGGL(coo_emit_fallback_kernel, dim3((unsigned)ceil_div(fb_total, 256)), dim3(256), 0, S(stream),
                       q_begin, flag_rows, n_flag, fb_off, fb_col, fb_val, fb_total, fb_base, M,
                       reinterpret_cast<unsigned long long*>(keys), vals);
    MELD_LAUNCH_CHECK("coo_emit_fallback_kernel");
  }
  return MELD_OK;
}

extern "C" size_t meld_sort_temp_bytes(int64_t n) {
  size_t bytes = 0;
  uint64_t* k = nullptr;
  double* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0u, 64u);
  return bytes + 256;
}

extern "C" int meld_sort_pairs_u64_f64(const uint64_t* keys_in, uint64_t* keys_out, const double* vals_in,
                                       double* vals_out, int64_t n, int end_bit, void* temp, size_t temp_bytes,
                                       meld_stream_t stream) {
  MELD_CHECK_ARG(keys_in && keys_out && vals_in && vals_out && temp && n >= 0 && end_bit > 0 && end_bit <= 64,
                 "meld_sort_pairs_u64_f64: bad arguments");
  if (n == 0) return MELD_OK;
  size_t bytes = temp_bytes;
  MELD_HIP_CALL(rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u,
                                          (unsigned)end_bit, S(stream)));
  return MELD_OK;
}

extern "C" size_t meld_merge_temp_bytes(int64_t n) {
  size_t bytes = 0;
  uint64_t* k = nullptr;
  double* v = nullptr;
  int64_t* c = nullptr;
  (void)rocprim::reduce_by_key(nullptr, bytes, k, v, (size_t)n, k, v, c, rocprim::plus<double>(),
                               rocprim::equal_to<uint64_t>());
  return bytes + 256;
}

extern "C" int meld_coo_merge(const uint64_t* keys_sorted, const double* vals_sorted, int64_t n, uint64_t* ukeys,
                              double* uvals, int64_t* n_unique, void* temp, size_t temp_bytes,
                              meld_stream_t stream) {
  MELD_CHECK_ARG(keys_sorted && vals_sorted && ukeys && uvals && n_unique && temp && n >= 0,
                 "meld_coo_merge: bad arguments");
  if (n == 0) {
    MELD_HIP_CALL(hipMemsetAsync(n_unique, 0, sizeof(int64_t), S(stream)));
    return MELD_OK;
  }
  size_t bytes = temp_bytes;
  MELD_HIP_CALL(rocprim::reduce_by_key(temp, bytes, keys_sorted, vals_sorted, (size_t)n, ukeys, uvals, n_unique,
                                       rocprim::plus<double>(), rocprim::equal_to<uint64_t>(), S(stream)));
  return MELD_OK;
}

extern "C" int meld_csr_from_keys(const uint64_t* ukeys, int64_t nnz, int64_t row_begin, int64_t n_rows,
                                  int64_t* rowptr, int32_t* col, meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && n_rows > 0 && nnz >= 0 && (nnz == 0 || (ukeys && col)), "meld_csr_from_keys: bad arguments");
  hipLaunchKernelGGL(csr_rowptr_kernel, dim3((unsigned)ceil_div(n_rows + 1, 256)), dim3(256), 0, S(stream),
                     reinterpret_cast<const unsigned long long*>(ukeys), nnz, row_begin, n_rows, rowptr);
  MELD_LAUNCH_CHECK("csr_rowptr_kernel");
  if (nnz > 0) {
    hipLaunchKernelGGL(csr_cols_kernel, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, S(stream),
                       reinterpret_cast<const unsigned long long*>(ukeys), nnz, col);
    MELD_LAUNCH_CHECK("csr_cols_kernel");
  }
  return MELD_OK;
}

extern "C" int meld_csr_row_sums(const int64_t* rowptr, const double* val, int64_t n_rows, double diag, double* out,
                                 meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && out && n_rows > 0, "meld_csr_row_sums: bad arguments");
  hipLaunchKernelGGL(csr_row_sums_kernel, dim3((unsigned)ceil_div(n_rows * 8, 256)), dim3(256), 0, S(stream), rowptr,
                     val, n_rows, diag, out);
  MELD_LAUNCH_CHECK("csr_row_sums_kernel");
  return MELD_OK;
}

extern "C" int meld_csr_anisotropy(const int64_t* rowptr, const int32_t* col, double* val, int64_t n_rows,
                                   const double* ksum_all, int64_t ksum_row_offset, double anisotropy,
                                   meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && ksum_all && n_rows > 0, "meld_csr_anisotropy: bad arguments");
  if (anisotropy == 0.0) return MELD_OK;
  hipLaunchKernelGGL(csr_anisotropy_kernel, dim3((unsigned)ceil_div(n_rows * 8, 256)), dim3(256), 0, S(stream),
                     rowptr, col, val, n_rows, ksum_all, ksum_row_offset, anisotropy);
  MELD_LAUNCH_CHECK("csr_anisotropy_kernel");
  return MELD_OK;
}
