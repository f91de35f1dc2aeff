// refine.hip -- exact fp64 re-evaluation of the kNN candidates and the alpha-decay kernel.
//
// Replaces the "Calculating affinities" block of graphtools
// ([UPSTREAM kNNGraph.build_kernel_to_data]: bandwidth = distances[:, knn], radius = bandwidth *
// (-log thresh)^(1/decay), re-search of rows whose farthest neighbour is inside the radius,
// K = exp(-(d/bw)^decay), K < thresh dropped), reached from reference meld/meld.py:273.
//
// Net semantics reproduced here (the tests restate them by O(N^2) brute force):
//   bw_i  = (knn+1)-th smallest euclidean distance of row i, self counted, clipped to eps
//   K_ij  = exp(-(d_ij / bw_i)^decay)  for every j with K_ij >= thresh
//
// Completeness proof per row (why a fixed-size candidate list is enough).  Let tau be the fp32
// squared distance of the ksel-th (last) candidate and E a bound on |d2_fp32 - d2_exact| (fp32
// FMA-chain bound E = KP * 2^-21 * max_i |x~_i|^2, or the split-fp16 bound 2^-16 * max |x~|^2).  Every reference that is NOT a candidate
// has d2_fp32 >= tau, hence exact d2 >= tau - E.  If radius^2 + E <= tau, no reference inside
// the radius (and therefore none of the knn+1 nearest, since radius >= bw) was missed, so bw
// and the row of K computed from the candidates are exact.  Rows that fail the test are
// flagged and recomputed by the exact fp64 sweep below -- graphtools' re-search fallback.
#include "common.hpp"

#include <algorithm>
#include <cfloat>

namespace meld {

constexpr int RB_FALL = 8;  // flagged rows per workgroup in the exact sweep
#ifndef RF_U
#define RF_UC 7  // rounds of 64 bytes per quad in flight in refine_kernel<true> (d = 50: the whole row)
#define RF_U 5  // 16-byte loads of a candidate row in flight per lane in refine_kernel (measured at d = 50: 5 -> 3.3 ms,
               // 8 -> 4.1 ms, 12 -> 5.7 ms: more registers per lane cost more occupancy than they add in flight)
#endif

// Summation order of the exact squared distances (refine_kernel, radius_exact_kernel and pair_distances_kernel must agree bit
// for bit: the sweep CONFIRMS bandwidths the other two produce).  Coordinate pairs (2 kk, 2 kk + 1), kk = 0 .. d / 2 - 1:
//   * even d <= RF_DMAX ("four chains"): pair kk belongs to slot kk & 3; a slot adds its pairs in rising order, the even
//     coordinate into one FMA chain and the odd one into another; u_j = even_j + odd_j; d2 = (u_0 + u_1) + (u_2 + u_3).
//     This is the order in which four neighbouring lanes that read 64 contiguous bytes of a candidate row per instruction
//     accumulate it (refine_kernel<true>), written out in scalar form by dist2_four_chains below.
//   * any other d: one even and one odd chain over all coordinates, then their sum (the order of rounds 1-5).
constexpr int RF_DMAX = 256;
__host__ __device__ __forceinline__ bool rf_four_chains(int d) { return (d & 1) == 0 && d <= RF_DMAX; }

__device__ __forceinline__ double dist2_four_chains(const double* __restrict__ xa, const double* __restrict__ xb, int d) {
  const double2* a2 = reinterpret_cast<const double2*>(xa);
  const double2* b2 = reinterpret_cast<const double2*>(xb);
  const int nk = d >> 1;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  int kk = 0;
  for (; kk + 4 <= nk; kk += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double2 a = a2[kk + j], b = b2[kk + j];
      const double t0 = a.x - b.x, t1 = a.y - b.y;
      s[j] = fma(t0, t0, s[j]);
      s1[j] = fma(t1, t1, s1[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (kk + j < nk) {
      const double2 a = a2[kk + j], b = b2[kk + j];
      const double t0 = a.x - b.x, t1 = a.y - b.y;
      s[j] = fma(t0, t0, s[j]);
      s1[j] = fma(t1, t1, s1[j]);
    }
  }
  return ((s[0] + s1[0]) + (s[1] + s1[1])) + ((s[2] + s1[2]) + (s[3] + s1[3]));
}

__device__ __forceinline__ double decay_kernel(double dist, double bw, double decay) {
  // decay = +inf: graphtools' decay=None, the unweighted kNN graph -- 1 for the knn + 1 nearest (self
  // included; bw is the distance of the last of them), 0 beyond; the radius factor is then 1
  if (isinf(decay)) return dist <= bw ? 1.0 : 0.0;
  double v = exp(-pow(dist / bw, decay));
  if (v != v) v = 1.0;  // graphtools: NaN -> 1
  return v;
}

// one wave per query row; lane c owns candidates c and c + 64.
// COOP (even d <= RF_DMAX; round 6): the candidate rows are read by FOUR lanes each -- 64 contiguous bytes per quad and
// instruction, 16 candidates per pass -- instead of by their own lane.  With a lane per candidate every load instruction
// touched as many cache lines as there were candidates (47 of a row's 59 listed ones are gathered at 1M x 50), and the kernel ran
// at the rate of the vector L1, not of anything behind it: 1.5e9 tag accesses (TCP_TOTAL_CACHE_ACCESSES) in 7.4 M cycles on 256
// CUs, TCP busy (TCP_GATE_EN1) 96 % of the time, L2 hit rate 85 %, 2.5-5 GB from the fabric for 18.7 GB gathered
// (profiles/r06_refine_pmc.txt).  The query row sits in LDS; the four lanes' partial sums meet by two DPP exchanges.
template <bool COOP>
__global__ __launch_bounds__(256) void refine_kernel(
    const double* __restrict__ X, int d, int64_t q_begin, int64_t q_count, const int* __restrict__ cand_idx,
    const float* __restrict__ cand_d2, const int* __restrict__ cand_cnt, const float* __restrict__ cand_thr, int ksel,
    int cap, int knn, double decay, double thresh, double radius_factor, const float* __restrict__ norm2_max, double err_coef,
    const float* __restrict__ norm2, double err_coef_lin, double bw_scale, const double* __restrict__ bw_fixed, int max_rank,
    double* __restrict__ bw_out, double* __restrict__ cand_val, int* __restrict__ keep_cnt,
    int* __restrict__ flag_rows, int* __restrict__ n_flag, const int* __restrict__ rows, int out_cap,
    int* __restrict__ cand_idx_out) {
  const int lane = threadIdx.x & 63;
  const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= q_count) return;
  // second-stage call: candidate row q belongs to local row rows[q]; results go to that row and the
  // candidate indices are copied into the first-stage index array so that later stages see one list
  const int64_t orow = rows ? (int64_t)rows[q] : q;
  const int64_t gi = q_begin + orow;
  const double* xi = X + gi * d;
  const size_t ro = (size_t)q * cap;
  // (COOP: everything the row needs from its candidate list is requested at once, whatever its length turns out to be -- the
  // chain count -> entry knn -> entries cost three round trips to L2 per row, a third of the kernel's time without its gather)
  int idx_raw[2] = {0, 0};
  float d2_raw[2] = {0.0f, 0.0f}, d2_knn = 0.0f;
  if constexpr (COOP) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int cc = min(lane + 64 * e, cap - 1);
      idx_raw[e] = cand_idx[ro + cc];
      d2_raw[e] = cand_d2[ro + cc];
    }
    d2_knn = cand_d2[ro + min(knn, cap - 1)];
  }
  const int n = min(cand_cnt[q], ksel);

  // search-error allowance: constant part + the part that scales with this row's own norm
  // (|q.r - q~.r~| <= c_lin |x_q| max|x_r|, the per-row form of the Cauchy-Schwarz bound)
  double E = err_coef * (double)norm2_max[0];
  if (norm2 != nullptr && err_coef_lin > 0.0) E += err_coef_lin * sqrt((double)norm2[gi] * (double)norm2_max[0]);
  // Candidates that cannot lie inside the radius are not gathered: the list is sorted by approximate
  // d2, the exact (knn+1)-th distance^2 is <= approx[knn] + E, so radius^2 <= rf^2 (approx[knn] + E), and
  // a candidate with approx > rf^2 (approx[knn] + E) + E has exact d2 > radius^2 (>= bandwidth^2): it
  // ranks behind the bandwidth entry and its kernel value is below thresh either way.
  // (graphtools' `bandwidth_scale` s multiplies the bandwidth, hence the radius: the bound uses max(rf s, 1) so that it never
  // falls below the bandwidth entry itself; a FIXED bandwidth -- graphtools' `bandwidth=` -- fixes the radius outright)
  double skip_above = INFINITY;
  const double skip_factor = fmax(radius_factor * bw_scale, 1.0);
  if (bw_fixed != nullptr) {
    const double rfix = fmax(bw_fixed[gi] * bw_scale, DBL_EPSILON) * radius_factor;
    skip_above = rfix * rfix * (1.0 + 1e-12) + E;
  } else if (n > knn) {
    skip_above = skip_factor * skip_factor * ((double)(COOP ? d2_knn : cand_d2[ro + knn]) + E) + E;
  }

  double dist[2];
  int idx[2];
  if constexpr (COOP) {
    __shared__ __attribute__((aligned(16))) double xis[4][RF_DMAX];
    __shared__ double dls[4][128];
    const int w = threadIdx.x >> 6;
    for (int k = lane; k < d; k += 64) xis[w][k] = xi[k];
    bool gth[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = lane + 64 * e;
      idx[e] = c < n ? idx_raw[e] : 0x7fffffff;
      gth[e] = c < n && (double)d2_raw[e] <= skip_above;
    }
    // (the list is sorted by approximate d2: the gathered candidates are a prefix of it)
    const int n_g = __popcll(__ballot(gth[0])) + __popcll(__ballot(gth[1]));
    const int grp = lane >> 2, j = lane & 3, nk = d >> 1;
    const double2* X2 = reinterpret_cast<const double2*>(X);
    const double2* xi2 = reinterpret_cast<const double2*>(&xis[w][0]);
    for (int c0 = 0; c0 < n_g; c0 += 16) {
      const int c = c0 + grp;
      const int i_lo = __shfl(idx[0], c & 63, 64), i_hi = __shfl(idx[1], c & 63, 64);
      const bool ok = c < n_g;
      const double2* xj2 = X2 + (int64_t)(ok ? (c < 64 ? i_lo : i_hi) : 0) * nk;
      double s = 0.0, s1 = 0.0;
      for (int k0 = 0; k0 < nk; k0 += 4 * RF_UC) {
        double2 b[RF_UC];
#pragma unroll
        for (int u = 0; u < RF_UC; ++u) {
          const int k = k0 + 4 * u + j;
          if (ok && k < nk) b[u] = xj2[k];
        }
#pragma unroll
        for (int u = 0; u < RF_UC; ++u) {
          const int k = k0 + 4 * u + j;
          if (ok && k < nk) {
            const double2 a = xi2[k];
            const double t0 = a.x - b[u].x, t1 = a.y - b[u].y;
            s = fma(t0, t0, s);
            s1 = fma(t1, t1, s1);
          }
        }
      }
      double u_ = s + s1;
      u_ += __shfl_xor(u_, 1, 64);
      u_ += __shfl_xor(u_, 2, 64);
      if (ok && j == 0) dls[w][c] = sqrt(u_);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) dist[e] = gth[e] ? dls[w][lane + 64 * e] : INFINITY;
  } else {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int c = lane + 64 * e;
    if (c < n) idx[e] = cand_idx[ro + c];
    if (c < n && (double)cand_d2[ro + c] <= skip_above) {
      const double* xj = X + (int64_t)idx[e] * d;
      double s = 0.0;
      if ((d & 1) == 0) {
        // rows are 16-byte aligned when d is even: 16-byte loads.  Summation order (shared with the
        // exact sweep below, which must reproduce these distances bit for bit): even coordinates in
        // one FMA chain, odd coordinates in another, then the sum of the two.
        const double2* xi2 = reinterpret_cast<const double2*>(xi);
        const double2* xj2 = reinterpret_cast<const double2*>(xj);
        double s1 = 0.0;
        const int nk = d / 2;
        int k = 0;
        // RF_U 16-byte loads of the candidate row in flight per lane (the plain loop issued one, waited for
        // it, and a wave spent ~25 L2 / HBM round trips per row: 5.9 ms at 1M); same summation order as before
        for (; k + RF_U <= nk; k += RF_U) {
          double2 b[RF_U];
#pragma unroll
          for (int u = 0; u < RF_U; ++u) b[u] = xj2[k + u];
#pragma unroll
          for (int u = 0; u < RF_U; ++u) {
            const double2 a = xi2[k + u];
            const double t0 = a.x - b[u].x, t1 = a.y - b[u].y;
            s = fma(t0, t0, s);
            s1 = fma(t1, t1, s1);
          }
        }
        for (; k < nk; ++k) {
          const double2 a = xi2[k], b = xj2[k];
          const double t0 = a.x - b.x, t1 = a.y - b.y;
          s = fma(t0, t0, s);
          s1 = fma(t1, t1, s1);
        }
        s += s1;
      } else {
        double s1 = 0.0;
        for (int k = 0; k + 1 < d; k += 2) {
          const double t0 = xi[k] - xj[k], t1 = xi[k + 1] - xj[k + 1];
          s = fma(t0, t0, s);
          s1 = fma(t1, t1, s1);
        }
        const double t = xi[d - 1] - xj[d - 1];
        s = fma(t, t, s);
        s += s1;
      }
      dist[e] = sqrt(s);
    } else {
      if (c >= n) idx[e] = 0x7fffffff;
      dist[e] = INFINITY;
    }
  }
  }

  // rank of each candidate by (dist, idx); the entry with rank == knn is the bandwidth.  The list is sorted by
  // approximate d2, so the candidates that were gathered are a prefix of it: the rest sit at +inf behind every finite
  // entry (their own rank is never used: their kernel value is 0 and the bandwidth entry, approximate rank <= knn,
  // is finite), and only the prefix has to be compared against
  const unsigned long long g0 = __ballot(dist[0] < INFINITY), g1 = __ballot(dist[1] < INFINITY);
  const int n_fin = g1 ? 64 + (64 - __clzll((long long)g1)) : (g0 ? 64 - __clzll((long long)g0) : 0);
  int rk[2] = {0, 0};
  const int ne = (n_fin > 64) ? 2 : 1;
  // Lists of at most 64 entries (every list, with ksel = 64) whose exact distances are all different -- nearly all of them -- are
  // ranked by `<` alone, four instructions per source entry: with a tie the ranks of the tied entries coincide and the sum of the
  // ranks falls short of n (n - 1) / 2, and only then is the list ranked again by (distance, index).
  bool ranked = false;
  if (ne == 1 && n <= 64) {
    int r0 = 0;
    for (int l2 = 0; l2 < n_fin; ++l2) {
      // (l2 is uniform: the source entry travels through scalar registers -- v_readlane, not the LDS crossbar)
      const double de = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dist[0]), l2), __builtin_amdgcn_readlane(__double2loint(dist[0]), l2));
      r0 += (de < dist[0]) ? 1 : 0;
    }
    int sum = dist[0] < INFINITY ? r0 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (sum == n_fin * (n_fin - 1) / 2) {
      rk[0] = r0;
      ranked = true;
    }
  }
  for (int e2 = 0; e2 < (ranked ? 0 : ne); ++e2) {
    const int lim = min(64, n_fin - 64 * e2);
    for (int l2 = 0; l2 < lim; ++l2) {
      const double de = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(dist[e2]), l2), __builtin_amdgcn_readlane(__double2loint(dist[e2]), l2));
      const int ie = __builtin_amdgcn_readlane(idx[e2], l2);
      rk[0] += (de < dist[0] || (de == dist[0] && ie < idx[0])) ? 1 : 0;
      if (n > 64) rk[1] += (de < dist[1] || (de == dist[1] && ie < idx[1])) ? 1 : 0;  // (lanes' second entries exist for lists beyond 64 only)
    }
  }
  double bw = 0.0;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const unsigned long long b = __ballot((lane + 64 * e) < n && rk[e] == knn);
    if (b) bw = __shfl(dist[e], __ffsll((long long)b) - 1, 64);
  }
  if (bw_fixed != nullptr) bw = bw_fixed[gi];
  bw = fmax(bw, DBL_EPSILON);       // what is recorded: the k-th neighbour distance (or the given bandwidth), unscaled
  const double bw_raw = bw;
  bw = fmax(bw * bw_scale, DBL_EPSILON);  // what the kernel uses [UPSTREAM build_kernel_to_data: bandwidth * bandwidth_scale, then max(., eps)]

  // completeness test in squared-distance space: everything inside the radius -- and, for the adaptive bandwidth, the
  // bandwidth entry itself (a scale below 1 / rf puts the radius inside it) -- must be certified present
  const double radius = bw * radius_factor;
  const double reach = bw_fixed != nullptr ? radius : fmax(radius, bw_raw);
  // tau: every reference that is not in the list has approximate d2 >= tau -- the last entry of a full
  // list, or the threshold the search published for a row it cut at the kernel radius (cand_thr)
  double tau = INFINITY;
  if (cand_cnt[q] >= ksel) tau = (double)cand_d2[ro + ksel - 1];
  if (cand_thr != nullptr) tau = fmin(tau, (double)cand_thr[q]);
  bool complete = (reach * reach + E <= tau);
  // graphtools' knn_max (max_rank = knn_max + 1, self counted; 0 = none): a row keeps its max_rank nearest cells at most.  The
  // list then only has to hold THOSE for sure: if the entry of rank max_rank - 1 is certified (no reference outside the list can
  // be closer), everything the row keeps is present even where the list does not reach the radius.
  if (max_rank > 0) {
    double d_m = INFINITY;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned long long b = __ballot((lane + 64 * e) < n && rk[e] == max_rank - 1 && dist[e] < INFINITY);
      if (b) d_m = __shfl(dist[e], __ffsll((long long)b) - 1, 64);
    }
    if (d_m < INFINITY && d_m * d_m + E <= tau && d_m >= bw_raw) complete = true;
  }
  if (bw_fixed == nullptr && n <= knn) complete = false;  // cannot even define the bandwidth from this list

  int kept = 0;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int c = lane + 64 * e;
    double v = 0.0;
    if (c < n && complete) {
      v = decay_kernel(dist[e], bw, decay);
      if (v < thresh || (int64_t)idx[e] == gi) v = 0.0;  // diagonal handled analytically (K_ii = 1)
      if (max_rank > 0 && rk[e] >= max_rank) v = 0.0;    // beyond the knn_max nearest
    }
    if (c < ksel) {
      cand_val[(size_t)orow * ksel + c] = v;
      if (cand_idx_out) cand_idx_out[(size_t)orow * out_cap + c] = (c < n) ? idx[e] : 0;
    }
    kept += __popcll(__ballot(v > 0.0));
  }
  if (lane == 0) {
    bw_out[orow] = bw_raw;
    keep_cnt[orow] = complete ? kept : 0;
    if (!complete) {
      const int pos = atomicAdd(n_flag, 1);
      flag_rows[pos] = (int)orow;
    }
  }
}

// Exact fp64 sweep for flagged rows: RB_FALL rows per workgroup, every thread walks a strided
// subset of all N references.  mode 0 counts, mode 1 fills.
__global__ __launch_bounds__(256) void radius_exact_kernel(
    const double* __restrict__ X, int64_t N, int d, int64_t q_begin, const int* __restrict__ flag_rows,
    int n_flag, const double* __restrict__ bw_all, int knn, double decay, double thresh, int mode,
    int* __restrict__ fb_cnt, const int64_t* __restrict__ fb_off, int* __restrict__ fb_cursor,
    int* __restrict__ fb_col, double* __restrict__ fb_val, int* __restrict__ err_flag, int64_t ref_chunk,
    double radius_factor, double bw_scale) {
  extern __shared__ __attribute__((aligned(16))) double xq[];  // [RB_FALL][d]
  __shared__ int s_cnt[RB_FALL];
  __shared__ int s_lt[RB_FALL];
  // grid: x = group of RB_FALL flagged rows, y = chunk of the references (a handful of flagged rows must
  // not be swept by a handful of workgroups: 2 rows x 1M references took 195 ms on one CU)
  const int f0 = blockIdx.x * RB_FALL;
  const int nf = min(RB_FALL, n_flag - f0);
  const int64_t ref_lo = (int64_t)blockIdx.y * ref_chunk;
  const int64_t ref_hi = min(N, ref_lo + ref_chunk);
  int64_t gi[RB_FALL];
  double bw[RB_FALL], rad[RB_FALL], bw_chk[RB_FALL];
#pragma unroll
  for (int f = 0; f < RB_FALL; ++f) {
    const int q = flag_rows[f0 + (f < nf ? f : 0)];
    gi[f] = q_begin + q;
    bw_chk[f] = bw_all[q];                             // the recorded (unscaled) bandwidth: what the count of closer cells verifies
    bw[f] = fmax(bw_all[q] * bw_scale, DBL_EPSILON);   // the bandwidth of the kernel (refine_kernel's scaling)
    // beyond this distance the kernel value is certainly below thresh (the radius, with a margin far above the
    // rounding of pow/exp): exp and pow are evaluated for the few references inside it only -- evaluating them
    // for every (row, reference) pair made the sweep 20x slower than its distance arithmetic
    rad[f] = bw[f] * radius_factor * (1.0 + 1e-9);
  }
  for (int u = threadIdx.x; u < RB_FALL * d; u += blockDim.x) {
    const int f = u / d, k = u % d;
    xq[u] = X[gi[f] * d + k];
  }
  if (threadIdx.x < RB_FALL) {
    s_cnt[threadIdx.x] = 0;
    s_lt[threadIdx.x] = 0;
  }
  __syncthreads();

  int cnt[RB_FALL], lt[RB_FALL];
#pragma unroll
  for (int f = 0; f < RB_FALL; ++f) cnt[f] = lt[f] = 0;

  // Four chains (even d <= RF_DMAX): FOUR threads per reference, thread j = slot j of the order at the head of the file -- it takes
  // the coordinate pairs j, j + 4, ... (a quad reads 64 contiguous bytes of the reference per instruction, as refine_kernel<true>
  // does), the quad joins its partial sums by two exchanges and its first thread goes on with the distance.  Other d: a thread
  // per reference, one even and one odd chain.
  const bool four = rf_four_chains(d);
  const int tpr = four ? 4 : 1;  // threads per reference
  const int jq = four ? (threadIdx.x & 3) : 0;
  const int nk = d >> 1;
  const double2* xq2 = reinterpret_cast<const double2*>(xq);
  for (int64_t base = ref_lo; base < ref_hi; base += blockDim.x / tpr) {
    const int64_t ref_raw = base + (four ? (threadIdx.x >> 2) : threadIdx.x);
    const bool live = ref_raw < ref_hi;
    const int64_t ref = live ? ref_raw : ref_hi - 1;
    const double* xr = X + ref * d;
    double s[RB_FALL], s1[RB_FALL];
#pragma unroll
    for (int f = 0; f < RB_FALL; ++f) s[f] = s1[f] = 0.0;
    if (four) {
      const double2* xr2 = reinterpret_cast<const double2*>(xr);
      for (int kk = jq; kk < nk; kk += 4) {
        const double2 b = xr2[kk];
#pragma unroll
        for (int f = 0; f < RB_FALL; ++f) {
          const double2 a = xq2[f * nk + kk];
          const double t0 = a.x - b.x, t1 = a.y - b.y;
          s[f] = fma(t0, t0, s[f]);
          s1[f] = fma(t1, t1, s1[f]);
        }
      }
#pragma unroll
      for (int f = 0; f < RB_FALL; ++f) {
        double u = s[f] + s1[f];
        u += __shfl_xor(u, 1, 64);
        u += __shfl_xor(u, 2, 64);
        s[f] = u;
        s1[f] = 0.0;
      }
    } else {
    for (int k = 0; k < d; ++k) {
      const double xv = xr[k];
      if ((k & 1) == 0) {
#pragma unroll
        for (int f = 0; f < RB_FALL; ++f) {
          const double t = xq[f * d + k] - xv;
          s[f] = fma(t, t, s[f]);
        }
      } else {
#pragma unroll
        for (int f = 0; f < RB_FALL; ++f) {
          const double t = xq[f * d + k] - xv;
          s1[f] = fma(t, t, s1[f]);
        }
      }
    }
    }
    if (!live || jq != 0) continue;
#pragma unroll
    for (int f = 0; f < RB_FALL; ++f) s[f] += s1[f];
#pragma unroll
    for (int f = 0; f < RB_FALL; ++f) {
      if (f < nf) {
        const double dist = sqrt(s[f]);
        if (dist < bw_chk[f]) lt[f]++;
        if (dist > rad[f]) continue;
        const double v = decay_kernel(dist, bw[f], decay);
        if (v >= thresh && ref != gi[f]) {
          if (mode == 0) {
            cnt[f]++;
          } else {
            const int pos = atomicAdd(&fb_cursor[f0 + f], 1);
            fb_col[fb_off[f0 + f] + pos] = (int)ref;
            fb_val[fb_off[f0 + f] + pos] = v;
          }
        }
      }
    }
  }
  if (mode == 0) {
#pragma unroll
    for (int f = 0; f < RB_FALL; ++f) {
      if (cnt[f]) atomicAdd(&s_cnt[f], cnt[f]);
      if (lt[f]) atomicAdd(&s_lt[f], lt[f]);
    }
    __syncthreads();
    if (threadIdx.x < nf) {
      // per-row totals over the reference chunks: kept entries in fb_cnt, references strictly closer than
      // the bandwidth in fb_cursor (checked, and cleared for the fill pass, by radius_check_kernel)
      if (s_cnt[threadIdx.x]) atomicAdd(&fb_cnt[f0 + threadIdx.x], s_cnt[threadIdx.x]);
      if (s_lt[threadIdx.x]) atomicAdd(&fb_cursor[f0 + threadIdx.x], s_lt[threadIdx.x]);
    }
  }
}

// bw is the (knn+1)-th smallest distance iff at most knn references are strictly closer.  A row whose bandwidth
// was clipped to eps (more than knn exact duplicates of the cell: the true bandwidth is 0, graphtools uses eps and
// the duplicates get K = 1) is fine by definition.  A row that fails gets fb_cnt = -1 (and err_flag): its
// candidate list missed one of its knn nearest cells, the caller recomputes its bandwidth exactly and sweeps again.
__global__ void radius_check_kernel(int* __restrict__ lt_then_cursor, int n_flag, int knn, int* __restrict__ err_flag,
                                    const int* __restrict__ flag_rows, const double* __restrict__ bw_all,
                                    int* __restrict__ fb_cnt) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_flag) return;
  if (lt_then_cursor[f] > knn && bw_all[flag_rows[f]] > DBL_EPSILON) {
    atomicOr(err_flag, 1);
    fb_cnt[f] = -1;
  }
  lt_then_cursor[f] = 0;
}

// Distances of given (row, reference) pairs in the arithmetic of refine_kernel and radius_exact_kernel (even / odd coordinate
// chains of FMAs, their sum, the square root): what the caller ranks to recompute a bandwidth the sweep will then CONFIRM -- the
// sweep counts the references strictly closer than the bandwidth in this very arithmetic, and a bandwidth taken from a library
// norm differs from it by a few ulps at d ~ 50 (the row was flagged again: "could not settle the bandwidth").
__global__ __launch_bounds__(256) void pair_distances_kernel(const double* __restrict__ X, int d, const int64_t* __restrict__ rows,
                                                             const int64_t* __restrict__ cand, int64_t n, int kk,
                                                             double* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * kk) return;
  const double* xq = X + rows[e / kk] * d;
  const double* xr = X + cand[e] * d;
  if (rf_four_chains(d)) {
    out[e] = sqrt(dist2_four_chains(xq, xr, d));
    return;
  }
  double s = 0.0, s1 = 0.0;
  for (int k = 0; k < d; ++k) {
    const double t = xq[k] - xr[k];
    if ((k & 1) == 0) s = fma(t, t, s); else s1 = fma(t, t, s1);
  }
  out[e] = sqrt(s + s1);
}

}  // namespace meld

using namespace meld;

extern "C" int meld_knn_refine(const double* X, int64_t N, int d, int64_t q_begin, int64_t q_count,
                               const int32_t* cand_idx, const float* cand_d2, const int32_t* cand_cnt,
                               const float* cand_thr, int ksel,
                               int cap, int knn, double decay, double thresh, const float* norm2_max, double err_coef,
                               const float* norm2, double err_coef_lin, double* bw,
                               double* cand_val, int32_t* keep_cnt, int32_t* flag_rows, int32_t* n_flag,
                               const int32_t* rows, int out_cap, int32_t* cand_idx_out, double bw_scale, const double* bw_fixed,
                               int max_rank, meld_stream_t stream) {
  MELD_CHECK_ARG(X && cand_idx && cand_d2 && cand_cnt && norm2_max && bw && cand_val && keep_cnt && flag_rows && n_flag,
                 "meld_knn_refine: null pointer");
  MELD_CHECK_ARG(q_count > 0 && q_begin >= 0 && d > 0 && (rows != nullptr || q_begin + q_count <= N),
                 "meld_knn_refine: bad sizes");
  MELD_CHECK_ARG(rows == nullptr || (cand_idx_out != nullptr && out_cap >= ksel),
                 "meld_knn_refine: a row list needs cand_idx_out with row stride >= ksel");
  MELD_CHECK_ARG(ksel >= 1 && ksel <= 128, "meld_knn_refine: ksel=%d outside [1,128]", ksel);
  MELD_CHECK_ARG(knn >= 0 && decay > 0 && thresh > 0 && thresh <= 1, "meld_knn_refine: bad kernel parameters");
  MELD_CHECK_ARG(cap >= ksel, "meld_knn_refine: row stride cap=%d smaller than ksel=%d", cap, ksel);
  MELD_CHECK_ARG(err_coef >= 0 && err_coef_lin >= 0, "meld_knn_refine: error coefficients must be non-negative");
  MELD_CHECK_ARG(bw_scale > 0 && bw_scale < INFINITY, "meld_knn_refine: bandwidth_scale must be positive and finite");
  MELD_CHECK_ARG(max_rank == 0 || max_rank > knn, "meld_knn_refine: max_rank=%d must exceed knn=%d (or be 0)", max_rank, knn);
  const double radius_factor = pow(-log(thresh), 1.0 / decay);
#define RF_LAUNCH(COOPV)                                                                                                          \
  hipLaunchKernelGGL(refine_kernel<COOPV>, dim3((unsigned)ceil_div(q_count, 4)), dim3(256), 0, S(stream), X, d, q_begin, q_count,  \
                     cand_idx, cand_d2, cand_cnt, cand_thr, ksel, cap, knn, decay, thresh, radius_factor, norm2_max, err_coef,     \
                     norm2, err_coef_lin, bw_scale, bw_fixed, max_rank, bw, cand_val, keep_cnt, flag_rows, n_flag, rows, out_cap, \
                     cand_idx_out)
  if (rf_four_chains(d))
    RF_LAUNCH(true);
  else
    RF_LAUNCH(false);
#undef RF_LAUNCH
  MELD_LAUNCH_CHECK("refine_kernel");
  return MELD_OK;
}

extern "C" int meld_knn_radius_exact(const double* X, int64_t N, int d, int64_t q_begin, const int32_t* flag_rows,
                                     int32_t n_flag, const double* bw, int knn, double decay, double thresh, int mode,
                                     int32_t* fb_cnt, const int64_t* fb_off, int32_t* fb_cursor, int32_t* fb_col,
                                     double* fb_val, int32_t* err_flag, double bw_scale, meld_stream_t stream) {
  if (n_flag == 0) return MELD_OK;
  MELD_CHECK_ARG(bw_scale > 0 && bw_scale < INFINITY, "meld_knn_radius_exact: bandwidth_scale must be positive and finite");
  MELD_CHECK_ARG(X && flag_rows && bw && n_flag > 0 && N > 0 && d > 0, "meld_knn_radius_exact: bad arguments");
  MELD_CHECK_ARG(mode == 0 ? (fb_cnt && err_flag && fb_cursor) : (fb_off && fb_cursor && fb_col && fb_val),
                 "meld_knn_radius_exact: missing output for mode %d", mode);
  const size_t lds = sizeof(double) * RB_FALL * d;
  // rows x reference chunks: about four workgroups per CU however few rows are flagged
  const int64_t n_groups = ceil_div(n_flag, RB_FALL);
  const int64_t want_chunks = std::max<int64_t>(1, 1024 / n_groups);
  const int64_t ref_chunk = std::max<int64_t>(1024, ceil_div(ceil_div(N, want_chunks), 256) * 256);
  const int64_t n_chunks = ceil_div(N, ref_chunk);
  hipStream_t st = S(stream);
  if (mode == 0) {
    MELD_HIP_CALL(hipMemsetAsync(fb_cnt, 0, sizeof(int32_t) * n_flag, st));
    MELD_HIP_CALL(hipMemsetAsync(fb_cursor, 0, sizeof(int32_t) * n_flag, st));
  }
  hipLaunchKernelGGL(radius_exact_kernel, dim3((unsigned)n_groups, (unsigned)n_chunks), dim3(256), lds, st, X, N, d, q_begin,
                     flag_rows, n_flag, bw, knn, decay, thresh, mode, fb_cnt, fb_off, fb_cursor, fb_col, fb_val, err_flag,
                     ref_chunk, pow(-log(thresh), 1.0 / decay), bw_scale);
  if (mode == 0)
    hipLaunchKernelGGL(radius_check_kernel, dim3((unsigned)ceil_div(n_flag, 256)), dim3(256), 0, st, fb_cursor, n_flag, knn,
                       err_flag, flag_rows, bw, fb_cnt);
  MELD_LAUNCH_CHECK("radius_exact_kernel");
  return MELD_OK;
}

// out[i][c] = |X[rows[i]] - X[cand[i][c]]| in the summation order of meld_knn_refine / meld_knn_radius_exact (see the kernel).
extern "C" int meld_knn_pair_distances(const double* X, int d, const int64_t* rows, const int64_t* cand, int64_t n, int kk, double* out,
                                       meld_stream_t stream) {
  MELD_CHECK_ARG(X && rows && cand && out && d > 0 && n >= 0 && kk > 0, "meld_knn_pair_distances: bad arguments");
  if (n == 0) return MELD_OK;
  hipLaunchKernelGGL(pair_distances_kernel, dim3((unsigned)ceil_div(n * kk, 256)), dim3(256), 0, S(stream), X, d, rows, cand, n, kk, out);
  MELD_LAUNCH_CHECK("pair_distances_kernel");
  return MELD_OK;
}
