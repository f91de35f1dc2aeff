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

// W_ij = K_ij / (ksum_i ksum_j)^a in place; dw_out (optional): the row sums of the result, in the order csr_row_sums_kernel adds
// them (lane g of the row's eight takes entries g, g + 8, ...; then the same three exchanges), so the separate pass over the values
// that computed the degrees is gone without moving a bit.  Four entries per lane at a time: columns, values, then the four
// gathers of ksum_j together -- the plain loop waited out a column load and a dependent gather per entry, five times per lane.
__global__ __launch_bounds__(256) void csr_anisotropy_kernel(const int64_t* __restrict__ rowptr,
                                                             const int* __restrict__ col, double* __restrict__ val,
                                                             int64_t n_rows, const double* __restrict__ ksum_all,
                                                             int64_t row_off, double anisotropy, double* __restrict__ dw_out) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int g = threadIdx.x & 7;
  double s = 0.0;
  if (r < n_rows) {
    const double di = ksum_all[row_off + r];
    const int64_t b = rowptr[r], e = rowptr[r + 1];
    constexpr int U = 4;
    for (int64_t k0 = b + g; k0 < e; k0 += 8 * U) {
      int c[U];
      double v[U], kj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t k = k0 + 8 * u;
        c[u] = k < e ? col[k] : 0;
        v[u] = k < e ? val[k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) kj[u] = ksum_all[c[u]];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t k = k0 + 8 * u;
        if (k < e) {
          const double dd = di * kj[u];
          const double w = (anisotropy == 1.0) ? v[u] / dd : v[u] / pow(dd, anisotropy);
          val[k] = w;
          s += w;
        }
      }
    }
  }
  if (dw_out != nullptr) {  // (uniform)
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (r < n_rows && g == 0) dw_out[r] = 0.0 + s;
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

// ---------------------------------------------------------------------------------------------
// Bucket assembly: unsorted COO -> CSR without a global sort.  The keys of one row are few (tens), so the
// entries are sent to fixed-size row buckets first (one pass, a per-row cursor advanced by atomics -- the order
// inside a bucket is whatever the atomics made it) and then every row is sorted by column and its duplicates
// summed inside ONE wave
// (bitonic network on (column : slot) keys, the values fetched by slot).  The result does not depend on the
// bucket order as long as no (row, column) occurs more than twice: a + b is commutative.  A third duplicate
// (never produced by (K + K^T)/2, where (i, j) occurs once per direction) or a row of more than 256 entries
// raises a flag and the caller takes the sort-based path, whose summation order is defined by the stable sort.
// 1M cells, 53 M entries in, 39 M out: 2.5 ms instead of 4.0 (6 radix passes over 16-byte pairs + reduce-by-key).
// ---------------------------------------------------------------------------------------------
// (Entries of one row often sit next to each other in the stream -- the rows' own entries are emitted row by row --
// and their atomics would serialise on one address: the lanes of a wave that hold a run of equal rows send ONE atomic.)
__device__ __forceinline__ void coo_row_run(int64_t r, int lane, bool* leader, int* rank, int* run_len) {
  const int64_t prev = __shfl_up(r, 1, 64);
  const bool head = lane == 0 || prev != r;
  const unsigned long long hb = __ballot(head);
  const unsigned long long below = hb & ((2ull << lane) - 1ull);  // heads at or below this lane (never empty: lane 0 is one)
  const int lead = 63 - __clzll((long long)below);
  const unsigned long long above = lane == 63 ? 0ull : (hb >> (lane + 1));
  const int next = above ? lane + 1 + (__ffsll((long long)above) - 1) : 64;
  *leader = head;
  *rank = lane - lead;
  *run_len = next - lane;  // (meaningful on the leader)
}

// row r's bucket = slots [r * CSR_BUCKET, (r + 1) * CSR_BUCKET) of tcol / tval; cursor[r] counts what was sent to it
constexpr int CSR_BUCKET = 256;

__global__ __launch_bounds__(256) void coo_scatter_rows_kernel(const unsigned long long* __restrict__ keys,
                                                               const double* __restrict__ vals, int64_t n, int64_t row_begin,
                                                               int64_t n_rows, int* __restrict__ cursor, int* __restrict__ tcol,
                                                               double* __restrict__ tval) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int64_t r = -1 - lane;  // (lanes past the end / rows outside the slice: runs of one, nothing sent)
  unsigned long long k = 0;
  if (e < n) {
    k = keys[e];
    const int64_t rr = (int64_t)(k >> 32) - row_begin;
    if (rr >= 0 && rr < n_rows) r = rr;
  }
  bool leader;
  int rank, run_len;
  coo_row_run(r, lane, &leader, &rank, &run_len);
  int first = 0;
  if (leader && r >= 0) first = atomicAdd(cursor + r, run_len);
  const int slot = __shfl(first, lane - rank, 64) + rank;
  if (r >= 0 && slot < CSR_BUCKET) {  // (an overfull bucket shows in cursor[r]: the caller takes the sort-based path)
    tcol[r * CSR_BUCKET + slot] = (int)(unsigned)k;
    tval[r * CSR_BUCKET + slot] = vals[e];
  }
}

// Row-sharded symmetrisation without a host round trip: the transposed entries a rank owes to the OTHER ranks go into a
// fixed-capacity send buffer, one segment of `cap` slots per owner (owner = row / rows_per_rank), keys and values of a
// segment next to each other -- send[o][0][slot] = key, send[o][1][slot] = value bits -- so that ONE equal-split
// all-to-all moves everything.  Unused slots keep the sentinel key ~0 (row 2^32 - 1: outside every slice, ignored by
// coo_scatter_rows_kernel).  counts[o] ends as the number of entries rank o is owed, whether they fitted or not.
// (Slots are claimed per WORKGROUP: a workgroup counts what its 256 x PR_PER entries owe every owner in LDS, reserves the ranges
// with one global atomic per owner and writes behind them.  One atomic per owner and wave on the `world` shared counters was
// what the kernel spent its time on: 0.67 ms for the 3.3 M entries of a 1/8 shard of 1M cells, every one of ~10^5 returning
// device-scope atomics to the same few addresses; 64 x fewer now.)
constexpr int PR_PER = 16;       // entries per thread
constexpr int PR_WORLD_MAX = 64;  // owners the LDS counters hold (a node has 8 GPUs)
__global__ __launch_bounds__(256) void coo_partition_remote_kernel(const unsigned long long* __restrict__ keys,
                                                                   const double* __restrict__ vals, int64_t n,
                                                                   int64_t rows_per_rank, int world, int self_rank, int64_t cap,
                                                                   int* __restrict__ counts, long long* __restrict__ send) {
  __shared__ int s_cnt[PR_WORLD_MAX];
  __shared__ int s_base[PR_WORLD_MAX];
  if (threadIdx.x < PR_WORLD_MAX) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t e0 = (int64_t)blockIdx.x * (256 * PR_PER) + threadIdx.x;
  unsigned long long k[PR_PER];
  int pos[PR_PER];  // owner << 24 | slot within the workgroup's range (a workgroup holds 4096 entries), or -1
#pragma unroll
  for (int u = 0; u < PR_PER; ++u) {
    const int64_t e = e0 + (int64_t)u * 256;
    pos[u] = -1;
    k[u] = 0;
    if (e < n) {
      k[u] = keys[e];
      const int o = (int)min((int64_t)(k[u] >> 32) / rows_per_rank, (int64_t)world - 1);
      if (o != self_rank) pos[u] = (o << 24) | atomicAdd(&s_cnt[o], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x < world && s_cnt[threadIdx.x] > 0) s_base[threadIdx.x] = atomicAdd(counts + threadIdx.x, s_cnt[threadIdx.x]);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PR_PER; ++u) {
    if (pos[u] >= 0) {
      const int o = pos[u] >> 24;
      const int64_t slot = (int64_t)s_base[o] + (pos[u] & 0xFFFFFF);
      if (slot < cap) {
        send[((int64_t)o * 2 + 0) * cap + slot] = (long long)k[u];
        send[((int64_t)o * 2 + 1) * cap + slot] = __double_as_longlong(vals[e0 + (int64_t)u * 256]);
      }
    }
  }
}

// Emit + scatter in one pass (single GPU, all rows local): the kept candidates of row q go straight into the row buckets --
// its own entries into the first keep_cnt[q] slots of bucket q (no atomic: cursor[] starts at keep_cnt[]), the transposed
// copy of each into the bucket of its column's row behind an atomic slot -- instead of being written as 2 M (key, value)
// pairs (512 MB at 1M cells) that the scatter kernel reads back.  cursor ends as meld_coo_scatter_rows leaves it.
__global__ __launch_bounds__(256) void coo_emit_scatter_rows_kernel(int64_t q_count, const int* __restrict__ cand_idx,
                                                                    const double* __restrict__ cand_val, int ksel, int cap,
                                                                    const int* __restrict__ keep_cnt, int* __restrict__ cursor,
                                                                    int* __restrict__ tcol, double* __restrict__ tval) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= q_count) return;
  if (keep_cnt[q] == 0) return;  // flagged (its entries come from the exact sweep) or empty row
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
      const int pos = written + __popcll(b & ((1ull << lane) - 1ull));
      const double hv = 0.5 * v;
      tcol[q * CSR_BUCKET + pos] = j;  // (pos < keep_cnt[q] <= ksel <= CSR_BUCKET)
      tval[q * CSR_BUCKET + pos] = hv;
      const int slot = atomicAdd(cursor + j, 1);
      if (slot < CSR_BUCKET) {  // (an overfull bucket shows in cursor[j]: the caller takes the sort-based path)
        tcol[(int64_t)j * CSR_BUCKET + slot] = (int)q;
        tval[(int64_t)j * CSR_BUCKET + slot] = hv;
      }
    }
    written += __popcll(b);
  }
}
// ... and the entries of the rows the exact sweep recomputed (thread per entry): fb_off[k] .. fb_off[k + 1] belong to row
// flag_rows[k], whose cursor started at their count
__global__ __launch_bounds__(256) void coo_emit_scatter_fallback_kernel(const int* __restrict__ flag_rows, int n_flag,
                                                                        const int64_t* __restrict__ fb_off,
                                                                        const int* __restrict__ fb_col,
                                                                        const double* __restrict__ fb_val, int64_t fb_total,
                                                                        int* __restrict__ cursor, int* __restrict__ tcol,
                                                                        double* __restrict__ tval) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= fb_total) return;
  int lo = 0, hi = n_flag - 1;  // row of entry e: last k with fb_off[k] <= e
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (fb_off[mid] <= e) lo = mid; else hi = mid - 1;
  }
  const int64_t q = flag_rows[lo];
  const int64_t pos = e - fb_off[lo];
  const int j = fb_col[e];
  const double hv = 0.5 * fb_val[e];
  if (pos < CSR_BUCKET) {
    tcol[q * CSR_BUCKET + pos] = j;
    tval[q * CSR_BUCKET + pos] = hv;
  }
  const int slot = atomicAdd(cursor + j, 1);
  if (slot < CSR_BUCKET) {
    tcol[(int64_t)j * CSR_BUCKET + slot] = (int)q;
    tval[(int64_t)j * CSR_BUCKET + slot] = hv;
  }
}

template <int SL, typename K>
__device__ __forceinline__ void csr_bitonic_sort(K (&key)[SL], int lane) {
  constexpr int NE = 64 * SL;
#pragma unroll
  for (int k = 2; k <= NE; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {
        const int je = j >> 6;
#pragma unroll
        for (int e = 0; e < SL; ++e) {
          if ((e & je) == 0) {
            const bool asc = ((64 * e + lane) & k) == 0;
            const K a = key[e], b = key[e | je];
            const bool sw = asc ? (b < a) : (a < b);
            key[e] = sw ? b : a;
            key[e | je] = sw ? a : b;
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < SL; ++e) {
          const int i = 64 * e + lane;
          const K o = __shfl_xor(key[e], j, 64);
          const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
          key[e] = keep_min ? (o < key[e] ? o : key[e]) : (o < key[e] ? key[e] : o);
        }
      }
    }
  }
}

// sorts the bucket [base, base + n) by column in place, sums pairs of equal columns, returns the number of
// distinct columns; *bad is set when a column occurs more than twice
// symm: how the two directions of an entry combine [UPSTREAM graphtools BaseGraph.symmetrize_kernel; the bucket holds HALVES]:
//   0  "+"    (K + K^T) / 2                     = a/2 + b/2 (an entry present in one direction only keeps its half)
//   1  "*"    K o K^T                           = 4 (a/2)(b/2); one-directional entries vanish
//   2  "mnn"  theta min(K, K^T) + (1 - theta) max(K, K^T), the missing direction counting as 0
// All three are symmetric functions of the pair, so W stays bitwise symmetric.
template <int SL>
__device__ __forceinline__ int csr_row_merge(int* __restrict__ tcol, double* __restrict__ tval, int64_t base, int n, int lane,
                                             bool* bad, int symm, double theta) {
  // keys (column : slot), slot < CSR_BUCKET = 2^8.  Columns below 2^24 - 1 -- every graph of fewer than 16.7 M cells -- fit a 32-bit
  // key with the slot: the network then compares and selects single words (the kernel is bound by the vector instructions of
  // its compare-exchanges, not by the exchanges themselves); same order, same result.  Decided per row, wave-uniformly.
  static_assert(CSR_BUCKET <= 256, "the slot shares a 32-bit key with a 24-bit column");
  unsigned cin[SL];
  bool wide = false;
#pragma unroll
  for (int e = 0; e < SL; ++e) {
    const int i = lane + 64 * e;
    cin[e] = i < n ? (unsigned)tcol[base + i] : 0u;
    wide |= cin[e] >= 0xFFFFFFu;
  }
  double v[SL];
  unsigned c[SL];
  if (!__any(wide)) {
    unsigned key[SL];
#pragma unroll
    for (int e = 0; e < SL; ++e) {
      const int i = lane + 64 * e;
      key[e] = i < n ? ((cin[e] << 8) | (unsigned)i) : ~0u;
    }
    csr_bitonic_sort<SL>(key, lane);
#pragma unroll
    for (int e = 0; e < SL; ++e) {
      const int i = lane + 64 * e;
      c[e] = i < n ? (key[e] >> 8) : 0xffffffffu;
      v[e] = i < n ? tval[base + (key[e] & 255u)] : 0.0;
    }
  } else {
    unsigned long long key[SL];
#pragma unroll
    for (int e = 0; e < SL; ++e) {
      const int i = lane + 64 * e;
      key[e] = i < n ? (((unsigned long long)cin[e] << 32) | (unsigned)i) : ~0ull;
    }
    csr_bitonic_sort<SL>(key, lane);
#pragma unroll
    for (int e = 0; e < SL; ++e) {
      const int i = lane + 64 * e;
      c[e] = (unsigned)(key[e] >> 32);
      v[e] = i < n ? tval[base + (unsigned)key[e]] : 0.0;
    }
  }
  // neighbours in sorted order: position i - 1, i + 1, i + 2 (across the lane / slot boundary)
  int total = 0;
  bool any_bad = false;
  unsigned head_col[SL];
  double head_val[SL];
  int head_pos[SL];
  bool is_head[SL];
#pragma unroll
  for (int e = 0; e < SL; ++e) {
    const int i = lane + 64 * e;
    unsigned prev = __shfl_up(c[e], 1, 64);
    if (lane == 0) prev = 0xffffffffu;
    if (e > 0) {
      const unsigned pl = __shfl(c[e - 1], 63, 64);
      if (lane == 0) prev = pl;
    }
    unsigned n1 = __shfl_down(c[e], 1, 64), n2 = __shfl_down(c[e], 2, 64);
    double v1 = __shfl_down(v[e], 1, 64);
    if (e + 1 < SL) {
      const unsigned f0 = __shfl(c[e + 1], 0, 64), f1 = __shfl(c[e + 1], 1, 64);
      const double w0 = __shfl(v[e + 1], 0, 64);
      if (lane == 63) { n1 = f0; n2 = f1; v1 = w0; }
      if (lane == 62) n2 = f0;
    } else {
      if (lane == 63) { n1 = 0xffffffffu; n2 = 0xffffffffu; }
      if (lane == 62) n2 = 0xffffffffu;
    }
    const bool valid = i < n;
    const bool head = valid && (i == 0 || prev != c[e]);
    const bool pair = valid && (i + 1 < n) && n1 == c[e];
    const bool triple = valid && (i + 2 < n) && n2 == c[e];
    any_bad |= triple;
    double hv = pair ? v[e] + v1 : v[e];
    bool keep_head = head;
    if (symm == 1) {
      hv = pair ? 4.0 * v[e] * v1 : 0.0;
      keep_head = head && pair;
    } else if (symm == 2) {
      hv = pair ? 2.0 * (theta * fmin(v[e], v1) + (1.0 - theta) * fmax(v[e], v1)) : 2.0 * (1.0 - theta) * v[e];
      keep_head = head && hv != 0.0;
    }
    is_head[e] = keep_head;
    head_col[e] = c[e];
    head_val[e] = hv;
    const unsigned long long hb = __ballot(keep_head);
    head_pos[e] = total + __popcll(hb & ((1ull << lane) - 1ull));
    total += __popcll(hb);
  }
  // (every load of the bucket precedes its first store: the sums depend on all of them)
#pragma unroll
  for (int e = 0; e < SL; ++e) {
    if (is_head[e]) {
      tcol[base + head_pos[e]] = (int)head_col[e];
      tval[base + head_pos[e]] = head_val[e];
    }
  }
  if (__any(any_bad)) *bad = true;
  return total;
}

__global__ __launch_bounds__(256) void csr_rows_sort_merge_kernel(const int* __restrict__ cursor, int64_t n_rows,
                                                                  int* __restrict__ tcol, double* __restrict__ tval,
                                                                  int* __restrict__ ucnt, int* __restrict__ flags, int symm, double theta) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int64_t base = r * CSR_BUCKET;
  const int n = cursor[r];
  if (n > CSR_BUCKET) {  // more entries than the bucket (and the in-register network) holds
    if (lane == 0) {
      atomicOr(flags, 1);
      ucnt[r] = 0;
    }
    return;
  }
  bool bad = false;
  int total;
  if (n <= 64)
    total = csr_row_merge<1>(tcol, tval, base, n, lane, &bad, symm, theta);
  else if (n <= 128)
    total = csr_row_merge<2>(tcol, tval, base, n, lane, &bad, symm, theta);
  else
    total = csr_row_merge<4>(tcol, tval, base, n, lane, &bad, symm, theta);
  if (lane == 0) {
    ucnt[r] = total;
    if (bad) atomicOr(flags, 2);
  }
}

// 16 lanes per row: the merged head of the bucket -> [rowptr[r], rowptr[r + 1])
__global__ __launch_bounds__(256) void csr_compact_rows_kernel(const int64_t* __restrict__ rowptr, int64_t n_rows,
                                                               const int* __restrict__ tcol, const double* __restrict__ tval,
                                                               int* __restrict__ col, double* __restrict__ val) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int g = threadIdx.x & 15;
  if (r >= n_rows) return;
  const int64_t src = r * CSR_BUCKET, dst = rowptr[r];
  const int len = (int)(rowptr[r + 1] - dst);
  for (int k = g; k < len; k += 16) {
    col[dst + k] = tcol[src + k];
    val[dst + k] = tval[src + k];
  }
}

// The same copy with the rows' sums on the way out: 8 lanes per row taking entries g, g + 8, ... and the three exchanges of
// csr_row_sums_kernel, so sums[r] = diag + sum equals what meld_csr_row_sums returns for the compacted rows, bit for bit, and the
// pass over the values that computed the kernel's row sums is gone.
__global__ __launch_bounds__(256) void csr_compact_rows_sums_kernel(const int64_t* __restrict__ rowptr, int64_t n_rows,
                                                                    const int* __restrict__ tcol, const double* __restrict__ tval,
                                                                    int* __restrict__ col, double* __restrict__ val, double diag,
                                                                    double* __restrict__ sums) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int g = threadIdx.x & 7;
  double s = 0.0;
  if (r < n_rows) {
    const int64_t src = r * CSR_BUCKET, dst = rowptr[r];
    const int len = (int)(rowptr[r + 1] - dst);
    for (int k = g; k < len; k += 8) {
      const double v = tval[src + k];
      col[dst + k] = tcol[src + k];
      val[dst + k] = v;
      s += v;
    }
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (r < n_rows && g == 0) sums[r] = diag + s;
}

extern "C" int meld_csr_bucket_slots(void) { return CSR_BUCKET; }

extern "C" int meld_coo_scatter_rows(const uint64_t* keys, const double* vals, int64_t n, int64_t row_begin, int64_t n_rows,
                                     int32_t* cursor, int32_t* tcol, double* tval, meld_stream_t stream) {
  MELD_CHECK_ARG(cursor && n_rows > 0 && n >= 0 && (n == 0 || (keys && vals && tcol && tval)),
                 "meld_coo_scatter_rows: bad arguments");
  MELD_HIP_CALL(hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)n_rows, S(stream)));
  if (n == 0) return MELD_OK;
  hipLaunchKernelGGL(coo_scatter_rows_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, S(stream),
                     reinterpret_cast<const unsigned long long*>(keys), vals, n, row_begin, n_rows, cursor, tcol, tval);
  MELD_LAUNCH_CHECK("coo_scatter_rows_kernel");
  return MELD_OK;
}

extern "C" int meld_coo_partition_remote(const uint64_t* keys, const double* vals, int64_t n, int64_t rows_per_rank, int world,
                                         int self_rank, int64_t cap, int32_t* counts, int64_t* send, meld_stream_t stream) {
  MELD_CHECK_ARG(counts && send && n >= 0 && (n == 0 || (keys && vals)) && rows_per_rank > 0 && world >= 1 && self_rank >= 0 &&
                     self_rank < world && cap >= 0,
                 "meld_coo_partition_remote: bad arguments");
  MELD_HIP_CALL(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)world, S(stream)));
  if (cap > 0) MELD_HIP_CALL(hipMemsetAsync(send, 0xFF, sizeof(int64_t) * 2 * (size_t)world * (size_t)cap, S(stream)));
  if (n == 0) return MELD_OK;
  MELD_CHECK_ARG(world <= PR_WORLD_MAX, "meld_coo_partition_remote: %d ranks (at most %d)", world, PR_WORLD_MAX);
  hipLaunchKernelGGL(coo_partition_remote_kernel, dim3((unsigned)ceil_div(n, 256 * PR_PER)), dim3(256), 0, S(stream),
                     reinterpret_cast<const unsigned long long*>(keys), vals, n, rows_per_rank, world, self_rank, cap, counts,
                     reinterpret_cast<long long*>(send));
  MELD_LAUNCH_CHECK("coo_partition_remote_kernel");
  return MELD_OK;
}

// cursor[n_rows] must hold, on entry, the number of OWN entries of every row (keep_cnt of complete rows, the sweep's count of
// flagged rows); see coo_emit_scatter_rows_kernel.  Rows = all N cells (q_begin = 0): the transposed entries land in local buckets.
extern "C" int meld_coo_emit_scatter(int64_t q_count, const int32_t* cand_idx, const double* cand_val, int ksel, int cap,
                                     const int32_t* keep_cnt, const int32_t* flag_rows, int32_t n_flag, const int64_t* fb_off,
                                     const int32_t* fb_col, const double* fb_val, int64_t fb_total, int32_t* cursor, int32_t* tcol,
                                     double* tval, meld_stream_t stream) {
  MELD_CHECK_ARG(cand_idx && cand_val && keep_cnt && cursor && tcol && tval && q_count > 0, "meld_coo_emit_scatter: bad arguments");
  MELD_CHECK_ARG(cap >= ksel && ksel <= CSR_BUCKET, "meld_coo_emit_scatter: ksel=%d must fit a row bucket (%d) and the row stride cap=%d",
                 ksel, CSR_BUCKET, cap);
  hipLaunchKernelGGL(coo_emit_scatter_rows_kernel, dim3((unsigned)ceil_div(q_count, 4)), dim3(256), 0, S(stream), q_count, cand_idx,
                     cand_val, ksel, cap, keep_cnt, cursor, tcol, tval);
  MELD_LAUNCH_CHECK("coo_emit_scatter_rows_kernel");
  if (n_flag > 0 && fb_total > 0) {
    MELD_CHECK_ARG(flag_rows && fb_off && fb_col && fb_val, "meld_coo_emit_scatter: missing fallback arrays");
    hipLaunchKernelGGL(coo_emit_scatter_fallback_kernel, dim3((unsigned)ceil_div(fb_total, 256)), dim3(256), 0, S(stream), flag_rows,
                       n_flag, fb_off, fb_col, fb_val, fb_total, cursor, tcol, tval);
    MELD_LAUNCH_CHECK("coo_emit_scatter_fallback_kernel");
  }
  return MELD_OK;
}

extern "C" int meld_csr_rows_sort_merge(const int32_t* cursor, int64_t n_rows, int32_t* tcol, double* tval, int32_t* ucnt,
                                        int32_t* flags, int symm, double theta, meld_stream_t stream) {
  MELD_CHECK_ARG(cursor && tcol && tval && ucnt && flags && n_rows > 0, "meld_csr_rows_sort_merge: bad arguments");
  MELD_CHECK_ARG(symm >= 0 && symm <= 2 && (symm != 2 || (theta >= 0.0 && theta <= 1.0)),
                 "meld_csr_rows_sort_merge: symm must be 0 (+), 1 (*) or 2 (mnn, theta in [0, 1])");
  MELD_HIP_CALL(hipMemsetAsync(flags, 0, sizeof(int32_t), S(stream)));
  hipLaunchKernelGGL(csr_rows_sort_merge_kernel, dim3((unsigned)ceil_div(n_rows, 4)), dim3(256), 0, S(stream), cursor, n_rows,
                     tcol, tval, ucnt, flags, symm, theta);
  MELD_LAUNCH_CHECK("csr_rows_sort_merge_kernel");
  return MELD_OK;
}

extern "C" int meld_csr_compact_rows(const int64_t* rowptr, int64_t n_rows, const int32_t* tcol, const double* tval,
                                     int32_t* col, double* val, meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && n_rows > 0 && tcol && tval && col && val, "meld_csr_compact_rows: bad arguments");
  hipLaunchKernelGGL(csr_compact_rows_kernel, dim3((unsigned)ceil_div(n_rows * 16, 256)), dim3(256), 0, S(stream), rowptr,
                     n_rows, tcol, tval, col, val);
  MELD_LAUNCH_CHECK("csr_compact_rows_kernel");
  return MELD_OK;
}

extern "C" int meld_csr_compact_rows_sums(const int64_t* rowptr, int64_t n_rows, const int32_t* tcol, const double* tval, int32_t* col,
                                          double* val, double diag, double* sums, meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && n_rows > 0 && tcol && tval && col && val && sums, "meld_csr_compact_rows_sums: bad arguments");
  hipLaunchKernelGGL(csr_compact_rows_sums_kernel, dim3((unsigned)ceil_div(n_rows * 8, 256)), dim3(256), 0, S(stream), rowptr, n_rows, tcol,
                     tval, col, val, diag, sums);
  MELD_LAUNCH_CHECK("csr_compact_rows_sums_kernel");
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
                     rowptr, col, val, n_rows, ksum_all, ksum_row_offset, anisotropy, (double*)nullptr);
  MELD_LAUNCH_CHECK("csr_anisotropy_kernel");
  return MELD_OK;
}

// The same, and the degrees dw[r] = sum_j W_rj of the result in the same pass (equal, bit for bit, to meld_csr_row_sums(diag = 0)
// called afterwards).
extern "C" int meld_csr_anisotropy_degrees(const int64_t* rowptr, const int32_t* col, double* val, int64_t n_rows,
                                           const double* ksum_all, int64_t ksum_row_offset, double anisotropy, double* dw,
                                           meld_stream_t stream) {
  MELD_CHECK_ARG(rowptr && ksum_all && dw && n_rows > 0, "meld_csr_anisotropy_degrees: bad arguments");
  if (anisotropy == 0.0) return meld_csr_row_sums(rowptr, val, n_rows, 0.0, dw, stream);
  hipLaunchKernelGGL(csr_anisotropy_kernel, dim3((unsigned)ceil_div(n_rows * 8, 256)), dim3(256), 0, S(stream),
                     rowptr, col, val, n_rows, ksum_all, ksum_row_offset, anisotropy, dw);
  MELD_LAUNCH_CHECK("csr_anisotropy_kernel");
  return MELD_OK;
}
