// common.hpp -- shared helpers for libmeld_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/meld_hip.h"

namespace meld {

// thread-local last-error string, reported through meld_last_error()
char* err_buf();
void set_err(const char* fmt, ...);

static inline hipStream_t S(meld_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// wave = 64 lanes on CDNA4; hard-coded on purpose (cdna_hip_programming.md section 1)
constexpr int WAVE = 64;

#define MELD_CHECK_ARG(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      meld::set_err(__VA_ARGS__);        \
      return MELD_ERR_INVALID;           \
    }                                    \
  } while (0)

#define MELD_LAUNCH_CHECK(name)                                               \
  do {                                                                        \
    hipError_t e__ = hipGetLastError();                                       \
    if (e__ != hipSuccess) {                                                  \
      meld::set_err("%s: launch failed: %s", name, hipGetErrorString(e__));   \
      return MELD_ERR_HIP;                                                    \
    }                                                                         \
  } while (0)

#define MELD_HIP_CALL(expr)                                                   \
  do {                                                                        \
    hipError_t e__ = (expr);                                                  \
    if (e__ != hipSuccess) {                                                  \
      meld::set_err("%s failed: %s", #expr, hipGetErrorString(e__));          \
      return MELD_ERR_HIP;                                                    \
    }                                                                         \
  } while (0)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Development switches of the library (A-B measurements, ablations, counters): read only under MELD_DEV=1, so that a stray
// variable in a production environment cannot change which kernels run (the same rule as meld_amd/_options.py on the host side)
static inline const char* meld_dev_getenv(const char* name) {
  const char* dev = getenv("MELD_DEV");
  return (dev && dev[0] == '1' && dev[1] == 0) ? getenv(name) : nullptr;
}

// Device-resident Lanczos on the tiled layout without a scalar kernel between the iterations: the SpMV of iteration k derives its
// own scalars from the partial sums the axpy of iteration k - 1 left (beta_{k-1} = sqrt(sum nrm2_prev), s_k = 1 / beta_{k-1},
// y = s_k L u_k - beta_{k-1} s_{k-1} u_{k-1}); block 0 records beta_{k-1} and s_k and clears the slots the next axpy adds into.
struct PtLanczos {
  const double* nrm2_prev;   // DOT_SLOTS partial sums of |w_{k-1}|^2 (k = 0: |u_0|^2 in slot 0)
  double* nrm2_zero;         // DOT_SLOTS slots to clear
  const double* state_prev;  // [0] = s_{k-1} (k = 0: 0)
  double* state_cur;         // [0] <- s_k
  double* betas;             // betas[it - 1] <- beta_{k-1} (it > 0)
  int it;
  const int* stop;  // (optional) nonzero when the launch starts: it does nothing (a stop request of the host's convergence check)
};

// one recurrence step on the panel-tiled layout (spmm_tiled.hip); coef_dev: device-resident Lanczos scalars
int pt_step(const meld_pt_layout_t* L, const int64_t* rowptr, const double* dw, int p, const double* x_full,
            int64_t x_row_offset, const double* z, double* y, double* r, double alpha, double beta, double gamma,
            double coef, double* dots, const double* coef_dev, hipStream_t st, double coef_x = 0.0, const PtLanczos* lz = nullptr);

}  // namespace meld
