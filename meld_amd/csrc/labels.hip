// labels.hip -- sample labels on the device: factorisation of fixed-width label words and the scaled one-hot signal.
//
// Reference: MELD._create_sample_indicators (meld/meld.py:143-191: `np.unique(labels)`, two boolean rows / a
// LabelBinarizer) and the column normalisation of transform (meld/meld.py:229-232).  What the filter needs of them is
// a code per cell (which of the p sample labels) and the label counts; the [N, p] indicator matrix is a scaled one-hot.
//
// meld_factorize_labels: labels arrive as fixed-width words (numpy 'U' / 'S' / 64-bit integer arrays viewed as int32
// [N, W]).  A workgroup builds a DICTIONARY of the distinct labels of its chunk in LDS -- entries are compared word
// by word, so two labels share a code iff they are equal (no hashing, no collision) -- with the first row and the
// count of every entry; one more workgroup merges the chunk dictionaries the same way.  p is small (the reference
// binarises sample labels: a handful), the dictionary holds up to FZ_MAXG entries; more distinct labels (or wider
// ones than FZ_MAXW words) are reported and the caller factorises on the host.
// (It replaces torch.unique + a stable argsort + an equality pass -- a rocPRIM merge sort of N keys, ~5.5 ms of GPU
// time at 1M cells that ran beside the candidate search and took CUs from it; these three launches take ~30 us.)
#include "common.hpp"

namespace meld {

constexpr int FZ_MAXG = 64;     // dictionary entries (distinct labels)
constexpr int FZ_MAXW = 16;     // words per label (64 bytes: numpy <U16 / S64)
constexpr int FZ_THREADS = 256;
constexpr int FZ_ROWS = 4;      // rows per thread and chunk (1024-row chunks: ~1000 workgroups at 1M cells; with 16 the 245 workgroups
                                // walked 16 batches each, 0.2 ms beside the search)
constexpr int FZ_CHUNK = FZ_THREADS * FZ_ROWS;

struct FzDict {
  int words[FZ_MAXG][FZ_MAXW];
  unsigned long long first[FZ_MAXG];
  unsigned long long count[FZ_MAXG];
  int n;         // entries
  int pick;      // the thread that inserts next
  int overflow;  // more than FZ_MAXG distinct labels
};

// One batch: every thread brings (at most) one label L[0..W) with its first row and weight; on return `code` is the
// label's dictionary entry (0 after an overflow).  Labels the dictionary does not hold yet are inserted one per round,
// by the lowest thread that holds one -- a deterministic order.  All threads of the workgroup call it together.
__device__ __forceinline__ int fz_batch(FzDict& D, bool valid, const int (&L)[FZ_MAXW], int W, unsigned long long row,
                                        unsigned long long weight) {
  const int tid = threadIdx.x;
  int code = -1;
  auto equal = [&](int e) {
    bool eq = true;
#pragma unroll
    for (int w = 0; w < FZ_MAXW; ++w)
      if (w < W) eq = eq && (D.words[e][w] == L[w]);
    return eq;
  };
  const int n0 = D.n;  // (uniform: written before the last barrier)
  if (valid)
    for (int e = 0; e < n0 && code < 0; ++e)
      if (equal(e)) code = e;
  while (true) {
    if (tid == 0) D.pick = FZ_THREADS;
    __syncthreads();
    if (valid && code < 0) atomicMin(&D.pick, tid);
    __syncthreads();
    const int pick = D.pick;
    const int n = D.n;
    __syncthreads();  // (everyone has read pick / n before they change)
    if (pick == FZ_THREADS) break;
    if (n == FZ_MAXG) {  // no room: every unresolved label maps to entry 0 and the caller is told
      if (tid == 0) D.overflow = 1;
      if (valid && code < 0) code = 0;
      __syncthreads();
      break;
    }
    if (tid == pick) {
#pragma unroll
      for (int w = 0; w < FZ_MAXW; ++w)
        if (w < W) D.words[n][w] = L[w];
      D.first[n] = ~0ull;
      D.count[n] = 0ull;
      D.n = n + 1;
    }
    __syncthreads();
    if (valid && code < 0 && equal(n)) code = n;
  }
  if (valid) {
    atomicMin(&D.first[code], row);
    atomicAdd(&D.count[code], weight);
  }
  return valid ? code : 0;
}

// chunk dictionaries: workgroup c factorises rows [c * FZ_CHUNK, (c + 1) * FZ_CHUNK)
__global__ __launch_bounds__(FZ_THREADS) void fz_local_kernel(const int* __restrict__ words, long long n_rows, int W,
                                                              int* __restrict__ local_codes, int* __restrict__ d_words,
                                                              unsigned long long* __restrict__ d_first,
                                                              unsigned long long* __restrict__ d_count, int* __restrict__ d_n,
                                                              int* __restrict__ status) {
  __shared__ FzDict D;
  const int tid = threadIdx.x;
  if (tid == 0) {
    D.n = 0;
    D.overflow = 0;
  }
  __syncthreads();
  const long long base = (long long)blockIdx.x * FZ_CHUNK;
  for (int k = 0; k < FZ_ROWS; ++k) {
    const long long r = base + (long long)k * FZ_THREADS + tid;
    const bool valid = r < n_rows;
    int L[FZ_MAXW];
#pragma unroll
    for (int w = 0; w < FZ_MAXW; ++w) L[w] = (valid && w < W) ? words[r * W + w] : 0;
    const int code = fz_batch(D, valid, L, W, (unsigned long long)r, 1ull);
    if (valid) local_codes[r] = code;
  }
  __syncthreads();
  const int n = D.n;
  if (tid == 0) {
    d_n[blockIdx.x] = n;
    if (D.overflow) atomicOr(status, 1);
  }
  for (int i = tid; i < n * FZ_MAXW; i += FZ_THREADS) d_words[(size_t)blockIdx.x * FZ_MAXG * FZ_MAXW + i] = D.words[i / FZ_MAXW][i % FZ_MAXW];
  if (tid < n) {
    d_first[(size_t)blockIdx.x * FZ_MAXG + tid] = D.first[tid];
    d_count[(size_t)blockIdx.x * FZ_MAXG + tid] = D.count[tid];
  }
}

// merge: one workgroup folds the chunk dictionaries into one; remap[c][e] = global entry of chunk c's entry e;
// head = [status, n_groups, first[FZ_MAXG], count[FZ_MAXG]] (one buffer for the host's single read-back)
__global__ __launch_bounds__(FZ_THREADS) void fz_merge_kernel(const int* __restrict__ d_words, const unsigned long long* __restrict__ d_first,
                                                              const unsigned long long* __restrict__ d_count, const int* __restrict__ d_n,
                                                              int n_chunks, int W, int* __restrict__ remap, const int* __restrict__ status,
                                                              long long* __restrict__ head) {
  __shared__ FzDict D;
  const int tid = threadIdx.x;
  if (tid == 0) {
    D.n = 0;
    D.overflow = 0;
  }
  __syncthreads();
  // (thread = chunk, round e = the chunks' e-th entries: as many rounds as the longest chunk dictionary -- two for two labels --
  // instead of one per 256 slots of the n_chunks x FZ_MAXG table: 149 -> a few us at 1M cells)
  __shared__ int s_more;
  for (int c0 = 0; c0 < n_chunks; c0 += FZ_THREADS) {
    const int c = c0 + tid;
    const int nc = c < n_chunks ? d_n[c] : 0;
    for (int e = 0; e < FZ_MAXG; ++e) {
      if (tid == 0) s_more = 0;
      __syncthreads();
      const bool valid = e < nc;
      if (valid) s_more = 1;
      __syncthreads();
      if (!s_more) break;  // (uniform; the barrier inside fz_batch separates this read from the next round's reset)
      const int i = c * FZ_MAXG + e;
      int L[FZ_MAXW];
#pragma unroll
      for (int w = 0; w < FZ_MAXW; ++w) L[w] = (valid && w < W) ? d_words[(size_t)i * FZ_MAXW + w] : 0;
      const int code = fz_batch(D, valid, L, W, valid ? d_first[i] : 0ull, valid ? d_count[i] : 0ull);
      if (valid) remap[i] = code;
    }
  }
  __syncthreads();
  if (tid == 0) {
    head[0] = (long long)(status[0] | D.overflow);
    head[1] = D.n;
  }
  if (tid < FZ_MAXG) {
    head[2 + tid] = tid < D.n ? (long long)D.first[tid] : -1;
    head[2 + FZ_MAXG + tid] = tid < D.n ? (long long)D.count[tid] : 0;
  }
}

__global__ __launch_bounds__(256) void fz_apply_kernel(const int* __restrict__ local_codes, const int* __restrict__ remap,
                                                       const int* __restrict__ rank, long long n_rows, long long* __restrict__ codes) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  const int g = remap[(r / FZ_CHUNK) * FZ_MAXG + local_codes[r]];
  codes[r] = rank ? rank[g] : g;
}

// out[i, :] = 0 except out[i, c] = scale[c] (1 without scale), c = codes[src(i)], src(i) = perm[i] (i without perm);
// rows [n_rows, n_pad) are zero (the isolated padding rows of a row shard)
__global__ __launch_bounds__(256) void indicator_signal_kernel(const long long* __restrict__ codes, const double* __restrict__ scale,
                                                               const long long* __restrict__ perm, long long n_rows, long long n_pad,
                                                               int p, double* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pad) return;
  int c = -1;
  double v = 0.0;
  if (i < n_rows) {
    c = (int)codes[perm ? perm[i] : i];
    v = scale ? scale[c] : 1.0;
  }
  double* o = out + i * p;
  if (p == 2) {
    *reinterpret_cast<double2*>(o) = make_double2(c == 0 ? v : 0.0, c == 1 ? v : 0.0);
  } else {
    for (int k = 0; k < p; ++k) o[k] = (k == c) ? v : 0.0;
  }
}

// out[perm[i], :] = in[i, :] -- results back in the caller's row order
__global__ __launch_bounds__(256) void scatter_rows_kernel(const double* __restrict__ in, const long long* __restrict__ perm,
                                                           long long n_rows, int p, double* __restrict__ out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p == 2) {
    if (t >= n_rows) return;
    *reinterpret_cast<double2*>(out + perm[t] * 2) = *reinterpret_cast<const double2*>(in + t * 2);
  } else {
    if (t >= n_rows * p) return;
    const long long i = t / p;
    const int k = (int)(t - i * p);
    out[perm[i] * p + k] = in[t];
  }
}

}  // namespace meld

using namespace meld;

extern "C" int meld_factorize_max_groups(void) { return FZ_MAXG; }
extern "C" int meld_factorize_max_words(void) { return FZ_MAXW; }

extern "C" size_t meld_factorize_temp_bytes(int64_t n_rows) {
  const size_t chunks = (size_t)ceil_div(n_rows > 0 ? n_rows : 1, FZ_CHUNK);
  // local codes, chunk dictionaries (words, first, count, n), remap, status word
  return (size_t)n_rows * 4 + chunks * (FZ_MAXG * FZ_MAXW * 4 + FZ_MAXG * 16 + 4 + FZ_MAXG * 4) + 1024;
}

static void fz_carve(void* temp, int64_t n_rows, int** local_codes, int** d_words, unsigned long long** d_first,
                     unsigned long long** d_count, int** d_n, int** remap, int** status) {
  const size_t chunks = (size_t)ceil_div(n_rows > 0 ? n_rows : 1, FZ_CHUNK);
  char* p = reinterpret_cast<char*>(temp);
  auto take = [&](size_t bytes) {
    char* q = p;
    p += (bytes + 63) & ~(size_t)63;
    return q;
  };
  // (8-byte arrays first: the carve keeps 64-byte alignment, sizes are rounded up inside the budget of temp_bytes + slack)
  *d_first = reinterpret_cast<unsigned long long*>(take(chunks * FZ_MAXG * 8));
  *d_count = reinterpret_cast<unsigned long long*>(take(chunks * FZ_MAXG * 8));
  *d_words = reinterpret_cast<int*>(take(chunks * FZ_MAXG * FZ_MAXW * 4));
  *remap = reinterpret_cast<int*>(take(chunks * FZ_MAXG * 4));
  *d_n = reinterpret_cast<int*>(take(chunks * 4));
  *status = reinterpret_cast<int*>(take(4));
  *local_codes = reinterpret_cast<int*>(take((size_t)n_rows * 4));
}

// words: int32 [n_rows, n_words] label words.  head (int64 [2 + 2 FZ_MAXG], device): [0] status (1: more than FZ_MAXG
// distinct labels -- nothing else is valid), [1] number of groups G, [2 .. 2 + MAXG) first row of every group,
// [2 + MAXG ..) group sizes.  Groups are numbered in the (deterministic) order the merge met them.
extern "C" int meld_factorize_labels(const int32_t* words, int64_t n_rows, int n_words, void* temp, size_t temp_bytes,
                                     int64_t* head, meld_stream_t stream) {
  MELD_CHECK_ARG(words && temp && head && n_rows > 0, "meld_factorize_labels: null argument or no rows");
  MELD_CHECK_ARG(n_words >= 1 && n_words <= FZ_MAXW, "meld_factorize_labels: labels of %d words (1..%d supported)", n_words, FZ_MAXW);
  MELD_CHECK_ARG(temp_bytes >= meld_factorize_temp_bytes(n_rows), "meld_factorize_labels: temp too small");
  MELD_CHECK_ARG(ceil_div(n_rows, FZ_CHUNK) * FZ_MAXG < (1ll << 30), "meld_factorize_labels: too many rows");
  int *local_codes, *d_words, *d_n, *remap, *status;
  unsigned long long *d_first, *d_count;
  fz_carve(temp, n_rows, &local_codes, &d_words, &d_first, &d_count, &d_n, &remap, &status);
  const int chunks = (int)ceil_div(n_rows, FZ_CHUNK);
  MELD_HIP_CALL(hipMemsetAsync(status, 0, 4, S(stream)));
  fz_local_kernel<<<chunks, FZ_THREADS, 0, S(stream)>>>(words, n_rows, n_words, local_codes, d_words, d_first, d_count, d_n, status);
  MELD_LAUNCH_CHECK("fz_local_kernel");
  fz_merge_kernel<<<1, FZ_THREADS, 0, S(stream)>>>(d_words, d_first, d_count, d_n, chunks, n_words, remap, status,
                                                   reinterpret_cast<long long*>(head));
  MELD_LAUNCH_CHECK("fz_merge_kernel");
  return MELD_OK;
}

// codes[i] = rank[group of row i] (rank: int32 [G] on the device -- the position of every group among the sorted labels,
// which only the host can tell -- or NULL for the group numbers themselves); temp as left by meld_factorize_labels
extern "C" int meld_factorize_codes(const void* temp, int64_t n_rows, const int32_t* rank, int64_t* codes, meld_stream_t stream) {
  MELD_CHECK_ARG(temp && codes && n_rows > 0, "meld_factorize_codes: null argument");
  int *local_codes, *d_words, *d_n, *remap, *status;
  unsigned long long *d_first, *d_count;
  fz_carve(const_cast<void*>(temp), n_rows, &local_codes, &d_words, &d_first, &d_count, &d_n, &remap, &status);
  fz_apply_kernel<<<(unsigned)ceil_div(n_rows, 256), 256, 0, S(stream)>>>(local_codes, remap, rank, n_rows, reinterpret_cast<long long*>(codes));
  MELD_LAUNCH_CHECK("fz_apply_kernel");
  return MELD_OK;
}

// The [n_pad, p] signal of the filter from the label codes (reference meld/meld.py:169-189 + the normalisation of
// :229-232), already in the device's row order: row i takes the code of cell perm[i] (perm may be NULL).
extern "C" int meld_indicator_signal(const int64_t* codes, const double* scale, const int64_t* perm, int64_t n_rows, int64_t n_pad,
                                     int p, double* out, meld_stream_t stream) {
  MELD_CHECK_ARG(codes && out && n_rows >= 0 && n_pad >= n_rows && p >= 1, "meld_indicator_signal: bad argument");
  if (n_pad == 0) return MELD_OK;
  indicator_signal_kernel<<<(unsigned)ceil_div(n_pad, 256), 256, 0, S(stream)>>>(
      reinterpret_cast<const long long*>(codes), scale, reinterpret_cast<const long long*>(perm), n_rows, n_pad, p, out);
  MELD_LAUNCH_CHECK("indicator_signal_kernel");
  return MELD_OK;
}

// out[perm[i], :] = in[i, :] for i < n_rows (fp64 rows of p columns): the densities back in the caller's cell order
extern "C" int meld_scatter_rows_f64(const double* in, const int64_t* perm, int64_t n_rows, int p, double* out, meld_stream_t stream) {
  MELD_CHECK_ARG(in && perm && out && n_rows >= 0 && p >= 1, "meld_scatter_rows_f64: bad argument");
  if (n_rows == 0) return MELD_OK;
  const int64_t work = (p == 2) ? n_rows : n_rows * p;
  scatter_rows_kernel<<<(unsigned)ceil_div(work, 256), 256, 0, S(stream)>>>(in, reinterpret_cast<const long long*>(perm), n_rows, p, out);
  MELD_LAUNCH_CHECK("scatter_rows_kernel");
  return MELD_OK;
}
