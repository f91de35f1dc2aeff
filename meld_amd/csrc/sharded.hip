// sharded.hip -- the row-sharded recurrences with the host out of the loop ("RCCL calls issued
// ... from C with a passed ncclComm_t").
//
// One process per GPU; cells are row-sharded (meld_amd/distributed.py).  A Chebyshev step / a Lanczos iteration on a shard is
// one kernel over the local rows followed by an all-gather of the new local slice of the iterate (+ ONE all-reduce of partial
// sums per Lanczos iteration).  Driven from Python that was 3-5 calls per step around a 20-26 us kernel: per rank the host
// spent ~15 ms per fit issuing 65 steps (tools/shard_emulate.py).  Here a whole filter / a whole batch of Lanczos iterations is
// ONE call that enqueues kernel, ncclAllGather, kernel, ... back to back on the caller's stream.
//
// RCCL is reached through dlopen("librccl.so.1") -- the copy the process has already loaded when torch.distributed runs on the
// "nccl" backend -- so libmeld_hip.so carries no link-time dependency on it and loads (and passes its ABI test) on a box
// without RCCL.  The communicator is the library's own (meld_rccl_comm_create from a unique id the ranks share, e.g. broadcast
// through torch.distributed); torch's process group keeps serving everything that is not on the per-step path.
#include "common.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <mutex>

namespace meld {
namespace {

// the slice of rccl.h this file needs (types only: the symbols come from dlsym)
typedef struct { char internal[128]; } nccl_unique_id_t;
typedef void* nccl_comm_t;
enum { NCCL_SUCCESS = 0 };
enum { NCCL_UINT8 = 1, NCCL_FLOAT64 = 8 };
enum { NCCL_SUM = 0 };

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(nccl_unique_id_t*) = nullptr;
  int (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id_t, int) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);  // the copy the process already holds, if any
      if (r.handle) break;
    }
    for (const char* n : names) {
      if (r.handle) break;
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.handle) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.handle, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.AllReduce;
  });
  return r;
}

#define MELD_RCCL_CALL(expr, what)                                                                    \
  do {                                                                                                \
    const int e__ = (expr);                                                                           \
    if (e__ != NCCL_SUCCESS) {                                                                        \
      Rccl& r__ = rccl();                                                                             \
      set_err("%s failed: RCCL error %d (%s)", what, e__, r__.GetErrorString ? r__.GetErrorString(e__) : "?"); \
      return MELD_ERR_HIP;                                                                            \
    }                                                                                                 \
  } while (0)

struct Comm {
  nccl_comm_t comm;
  int world, rank;
};

// all-gather of `bytes` per rank, IN PLACE: rank g's slice of full already holds its contribution
int gather_in_place(Comm* c, void* full, size_t bytes, hipStream_t st) {
  Rccl& r = rccl();
  const char* mine = reinterpret_cast<const char*>(full) + (size_t)c->rank * bytes;
  MELD_RCCL_CALL(r.AllGather(mine, full, bytes, NCCL_UINT8, c->comm, st), "ncclAllGather");
  return MELD_OK;
}

}  // namespace
}  // namespace meld

using namespace meld;

extern "C" int meld_rccl_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int meld_rccl_unique_id(void* id_host) {
  MELD_CHECK_ARG(id_host, "meld_rccl_unique_id: null argument");
  Rccl& r = rccl();
  MELD_CHECK_ARG(r.ok, "meld_rccl_unique_id: librccl.so.1 could not be loaded");
  nccl_unique_id_t id;
  MELD_RCCL_CALL(r.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id_host, &id, sizeof(id));
  return MELD_OK;
}

extern "C" int meld_rccl_comm_create(const void* id_host, int world, int rank, void** comm_out) {
  MELD_CHECK_ARG(id_host && comm_out && world >= 1 && rank >= 0 && rank < world, "meld_rccl_comm_create: bad arguments");
  Rccl& r = rccl();
  MELD_CHECK_ARG(r.ok, "meld_rccl_comm_create: librccl.so.1 could not be loaded");
  nccl_unique_id_t id;
  memcpy(&id, id_host, sizeof(id));
  nccl_comm_t c = nullptr;
  MELD_RCCL_CALL(r.CommInitRank(&c, world, id, rank), "ncclCommInitRank");
  *comm_out = new Comm{c, world, rank};
  return MELD_OK;
}

extern "C" int meld_rccl_comm_destroy(void* comm) {
  if (!comm) return MELD_OK;
  Comm* c = reinterpret_cast<Comm*>(comm);
  Rccl& r = rccl();
  if (r.ok && c->comm) r.CommDestroy(c->comm);
  delete c;
  return MELD_OK;
}

extern "C" int meld_rccl_all_gather(void* comm, const void* send, void* recv, size_t bytes_per_rank, meld_stream_t stream) {
  MELD_CHECK_ARG(comm && send && recv, "meld_rccl_all_gather: null argument");
  Comm* c = reinterpret_cast<Comm*>(comm);
  MELD_RCCL_CALL(rccl().AllGather(send, recv, bytes_per_rank, NCCL_UINT8, c->comm, S(stream)), "ncclAllGather");
  return MELD_OK;
}

extern "C" int meld_rccl_all_reduce_sum_f64(void* comm, double* buf, size_t count, meld_stream_t stream) {
  MELD_CHECK_ARG(comm && buf, "meld_rccl_all_reduce_sum_f64: null argument");
  Comm* c = reinterpret_cast<Comm*>(comm);
  MELD_RCCL_CALL(rccl().AllReduce(buf, buf, count, NCCL_FLOAT64, NCCL_SUM, c->comm, S(stream)), "ncclAllReduce");
  return MELD_OK;
}

// Steps k = 2 .. n_coef - 1 of the Chebyshev recurrence on a row shard, in one call:
//   T_k(local rows) = alpha2 L T_{k-1} + beta2 T_{k-1} - T_{k-2};  r += c_k T_k;  all-gather of the local slice of T_k
// [UPSTREAM pygsp cheby_op, reference meld/filter.py:59; the partitioning of SURVEY.md section 8e].
// t_a / t_b: the two FULL-length iterates [world * rows_pad, p] holding T_0 / T_1 (gathered) on entry, used as ping-pong
// buffers (T_k overwrites the local rows of T_{k-2}, then the slices are gathered in place); r [rows_pad, p]: the local rows
// of the result, already holding c_0 / 2 T_0 + c_1 T_1.  layout: the shard's panel-tiled layout, or NULL for the CSR-stream
// kernel (col / val are only read then).  On the tiled kernel the accumulator is touched every other step (as
// meld_pt_cheby_run).  coeffs: n_coef doubles on the HOST.  *last (optional) = 1 if t_b holds the last T, 0 if t_a does.
extern "C" int meld_cheby_run_sharded(void* comm, const meld_pt_layout_t* layout, const int64_t* rowptr, const int32_t* col,
                                      const double* val, const double* dw, int64_t n_rows, int64_t nnz, int64_t rows_pad,
                                      int64_t row_begin, int p, double* t_a, double* t_b, double* r, const double* coeffs,
                                      int n_coef, double alpha2, double beta2, int* last, meld_stream_t stream) {
  MELD_CHECK_ARG(comm && rowptr && dw && t_a && t_b && r && coeffs && n_rows >= 0 && rows_pad >= n_rows && row_begin >= 0 &&
                     p >= 1 && n_coef >= 2 && (layout || (col && val) || n_rows == 0),
                 "meld_cheby_run_sharded: bad arguments");
  Comm* c = reinterpret_cast<Comm*>(comm);
  MELD_CHECK_ARG(row_begin == (int64_t)c->rank * rows_pad, "meld_cheby_run_sharded: row_begin %lld is not rank %d's slice of %lld rows",
                 (long long)row_begin, c->rank, (long long)rows_pad);
  hipStream_t st = S(stream);
  double* t_old = t_a;
  double* t_cur = t_b;
  int which = 1;
  const size_t slice = (size_t)rows_pad * p * sizeof(double);
  auto step = [&](int k, bool touch_r, double coef, double coef_x) -> int {
    double* loc = t_old + (size_t)row_begin * p;  // T_k overwrites the local rows of T_{k-2} (z and y alias)
    if (n_rows > 0) {
      if (layout) {
        const int rc = pt_step(layout, rowptr, dw, p, t_cur, row_begin, loc, loc, touch_r ? r : nullptr, alpha2, beta2, -1.0, coef,
                               nullptr, nullptr, st, coef_x);
        if (rc != MELD_OK) return rc;
      } else {
        const int rc = meld_cheby_step(rowptr, col, val, dw, n_rows, nnz, p, t_cur, row_begin, loc, loc, r, alpha2, beta2, -1.0,
                                       coeffs[k], nullptr, stream);
        if (rc != MELD_OK) return rc;
      }
    }
    const int rc = gather_in_place(c, t_old, slice, st);
    if (rc != MELD_OK) return rc;
    std::swap(t_old, t_cur);
    which ^= 1;
    return MELD_OK;
  };
  int k = 2;
  if (!layout) {
    for (; k < n_coef; ++k) {
      const int rc = step(k, true, coeffs[k], 0.0);
      if (rc != MELD_OK) return rc;
    }
  } else {
    if ((n_coef - 2) % 2 == 1) {  // an odd number of steps: the first one alone
      const int rc = step(k, true, coeffs[k], 0.0);
      if (rc != MELD_OK) return rc;
      ++k;
    }
    for (; k + 1 < n_coef; k += 2) {
      int rc = step(k, false, 0.0, 0.0);  // T_k: no accumulator traffic
      if (rc != MELD_OK) return rc;
      // T_{k+1}, and r += c_{k+1} T_{k+1} + c_k T_k (T_k = this step's own rows of the gathered iterate)
      rc = step(k + 1, true, coeffs[k + 1], coeffs[k]);
      if (rc != MELD_OK) return rc;
    }
  }
  if (last) *last = which;
  MELD_LAUNCH_CHECK("meld_cheby_run_sharded");
  return MELD_OK;
}

// Iterations [it_begin, it_begin + n_iter) of the one-reduction Lanczos recurrence on a row shard (the iteration of
// meld_lanczos_fold / meld_lanczos_axpy3, include/meld_hip.h; [UPSTREAM pygsp estimate_lmax], reference meld/filter.py:39), in
// one call: per iteration the SpMV of the local rows, ONE all-reduce of the 3 x slots partial sums, the one-wave scalar kernel,
// the three-term update and the all-gather of the new vector.  v0 / v1 / v2: FULL-length rotating vectors [world * rows_pad];
// state [8], acc [3 * meld_spmm_dot_slots()], alphas / betas as for the phase entry points.
extern "C" int meld_lanczos_steps_sharded(void* comm, const meld_pt_layout_t* layout, const int64_t* rowptr, const int32_t* col,
                                          const double* val, const double* dw, int64_t n_rows, int64_t nnz, int64_t rows_pad,
                                          int64_t row_begin, double* v0, double* v1, double* v2, double* state, double* acc,
                                          double* alphas, double* betas, int it_begin, int n_iter, meld_stream_t stream) {
  MELD_CHECK_ARG(comm && rowptr && dw && v0 && v1 && v2 && state && acc && alphas && betas && n_rows >= 0 && rows_pad >= n_rows &&
                     it_begin >= 0 && n_iter >= 0 && (layout || (col && val) || n_rows == 0),
                 "meld_lanczos_steps_sharded: bad arguments");
  Comm* c = reinterpret_cast<Comm*>(comm);
  MELD_CHECK_ARG(row_begin == (int64_t)c->rank * rows_pad, "meld_lanczos_steps_sharded: row_begin is not this rank's slice");
  Rccl& rc_ = rccl();
  hipStream_t st = S(stream);
  double* V[3] = {v0, v1, v2};
  const int slots = meld_spmm_dot_slots();
  for (int k = it_begin; k < it_begin + n_iter; ++k) {
    double* u_prev = V[k % 3];
    double* u = V[(k + 1) % 3];
    double* y = V[(k + 2) % 3];
    int rc;
    if (layout)
      rc = meld_pt_lanczos_spmv(layout, rowptr, dw, n_rows, u, row_begin, u_prev + row_begin, y + row_begin, state, acc, stream);
    else
      rc = meld_lanczos_spmv(rowptr, col, val, dw, n_rows, nnz, u, row_begin, u_prev + row_begin, y + row_begin, state, acc, stream);
    if (rc != MELD_OK) return rc;
    MELD_RCCL_CALL(rc_.AllReduce(acc, acc, (size_t)3 * slots, NCCL_FLOAT64, NCCL_SUM, c->comm, st), "ncclAllReduce");
    rc = meld_lanczos_fold(state, acc, alphas, betas, k, stream);
    if (rc != MELD_OK) return rc;
    rc = meld_lanczos_axpy3(y + row_begin, u + row_begin, u_prev + row_begin, rows_pad, state, acc + 2 * slots, stream);
    if (rc != MELD_OK) return rc;
    rc = gather_in_place(c, y, (size_t)rows_pad * sizeof(double), st);
    if (rc != MELD_OK) return rc;
  }
  return MELD_OK;
}
