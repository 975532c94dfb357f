// vk_kernel_like.h: chi-square / log-likelihood kernel - part of libvictor_hip.so (see victor_hip.hip for the overview and DESIGN.md section 5).
#pragma once
#include "vk_common.h"

namespace vk {

// --------------------------------------------------------------------------------------------------
// shared pieces of the likelihood stage
// --------------------------------------------------------------------------------------------------
// likelihood form (ccf_fit.py:455-473); `factor` = -1/2 log det of the covariance when it depends on beta, else 0
__device__ __forceinline__ double like_form(const LikeArgs& a, double chisq, double factor) {
  const double nm = a.nmocks;
  if (a.like_form == VK_LIKE_SELLENTIN) return -nm * log(1.0 + chisq / (nm - 1.0)) / 2.0 + factor;
  if (a.like_form == VK_LIKE_HARTLAP) return -0.5 * chisq * ((nm - a.N - 2.0) / (nm - 1.0)) + factor;
  if (a.like_form == VK_LIKE_PERCIVAL) {
    const double nd = (double)a.N;
    const double B = (nm - nd - 2.0) / ((nm - nd - 1.0) * (nm - nd - 4.0));
    const double m = a.nparams + 2.0 + (nm - 1.0 + B * (nd - a.nparams)) / (1.0 + B * (nd - a.nparams));
    return -m * log(1.0 + chisq / (nm - 1.0)) / 2.0 + factor;
  }
  return -0.5 * chisq + factor;
}

// precision / covariance bracket, ccf_fit.py:213-228,245-260: below / above the grid -> first / last slice, exact grid
// value -> that slice, else blend of slice `lo` and the LAST slice with weight t (upper bracket = last grid entry)
__device__ __forceinline__ void cov_bracket(const LikeArgs& a, double beta, int* lo_out, double* t_out) {
  int lo = 0;
  double t = 0.0;
  const int last = a.n_beta_c - 1;
  if (beta < a.beta_c[0]) {
    lo = 0;
  } else if (beta > a.beta_c[last]) {
    lo = last;
  } else {
    int exact = -1, below = 0;
    for (int i = 0; i <= last; ++i) {
      const double g = a.beta_c[i];
      if (g == beta && exact < 0) exact = i;
      if (g < beta) below = i;
    }
    if (exact >= 0) {
      lo = exact;
    } else {
      lo = below;
      t = (beta - a.beta_c[lo]) / (a.beta_c[last] - a.beta_c[lo]);
    }
  }
  *lo_out = lo;
  *t_out = t;
}

// One factor 1 - t + t lambda_i of det((1-t) C_lo + t C_last) = det(C_lo) prod_i (1 - t + t lambda_i).  The reference
// tests the SIGN of the blended determinant (np.linalg.slogdet(...)[0] != 1, ccf_fit.py:447-450): a zero factor or an
// odd number of negative ones fails, an even number of negative factors passes with log|det|.
__device__ __forceinline__ double logdet_term(double fct, int* n_neg, int* n_bad) {
  *n_neg += (fct < 0.0) ? 1 : 0;
  *n_bad += (fct == 0.0 || fct != fct) ? 1 : 0;
  return log(fabs(fct));
}

// sum over the workgroup's kBlock threads, deterministic; `red` = kWaves doubles of LDS; every thread gets the result
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();                       // `red` may still be read from a previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) s += red[w];
  return s;
}

// number of threads of the workgroup whose predicate holds, to every thread; `red` as for block_sum.  (Not __syncthreads_count:
// HIP implements that through a static __shared__ word, which moves the base of dynamic LDS off zero for the whole kernel -
// every LDS address of the theory loops then carries an addend the ds_read offset field could have held.)
__device__ __forceinline__ int block_count(bool pred, double* red) {
  const int c = __popcll(__ballot(pred));
  __syncthreads();                       // `red` may still be read from a previous call
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (double)c;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int w = 1; w < kWaves; ++w) s += red[w];
  return (int)s;
}

// sum_b (sum_{r in [r0, r1)} th[r] P[r][b]) th[b] over this lane's column pairs b = 2 lane, 2 lane + 128, ... (N even), with
// P = (1-t) P0 + t P1 when BLEND.  The rows of a column pair are streamed through a rolling window of D 16-byte loads:
// D in flight at any time, so the whole column costs one exposed round trip to L2 plus issue time, in 4 D registers.
template <bool BLEND>
__device__ __forceinline__ double like_quadratic(int N, const double* P0, const double* P1, double t, const double* th, int r0,
                                                 int r1, int lane) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int D = BLEND ? 4 : 8;
  const double omt = 1.0 - t;
  const int rows = r1 - r0;
  const unsigned stride = (unsigned)N * 8u;             // bytes; a slice is < 4 GB, so 32-bit byte offsets from its (uniform) base
  const char* s0 = reinterpret_cast<const char*>(P0);
  const char* s1 = reinterpret_cast<const char*>(P1);
  double part = 0.0;
  for (int b = 2 * lane; b < N; b += 128) {
    unsigned next = ((unsigned)r0 * (unsigned)N + (unsigned)b) * 8u;   // byte offset of the next row to fetch
    d2 p[D], q[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      p[j] = d2{0.0, 0.0};
      q[j] = d2{0.0, 0.0};
      if (j < rows) {
        p[j] = *reinterpret_cast<const d2*>(s0 + next);
        if (BLEND) q[j] = *reinterpret_cast<const d2*>(s1 + next);
        next += stride;
      }
    }
    double y0 = 0.0, y1 = 0.0;
    for (int base = 0; base < rows; base += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int i = base + j;
        const d2 cp = p[j], cq = q[j];
        if (i + D < rows) {
          p[j] = *reinterpret_cast<const d2*>(s0 + next);
          if (BLEND) q[j] = *reinterpret_cast<const d2*>(s1 + next);
          next += stride;
        }
        if (i < rows) {
          const double w = th[r0 + i];
          y0 = fma(w, BLEND ? omt * cp.x + t * cq.x : cp.x, y0);
          y1 = fma(w, BLEND ? omt * cp.y + t * cq.y : cp.y, y1);
        }
      }
    }
    part = fma(y0, th[b], part);
    part = fma(y1, th[b + 1], part);
  }
  return part;
}

// chi2 / lnL of ONE point by a whole workgroup (kBlock threads).  On entry `th` (LDS, N doubles) holds the point's theory
// vector and the workgroup is synchronised; `th` is overwritten with the residual.  `red`: kWaves + 2 doubles of LDS.
// Threads = (row slice, column pair): rows of the precision matrix split over the four waves, lanes over PAIRS of adjacent
// columns (one 16-byte load serves two), rows streamed through a rolling window of loads (like_quadratic): one point's N^2
// products cost about one round trip to L2, not the N / 64 * N of the wave-per-point kernel - this is what a batch of one
// (the reference's calling convention, CCFLikelihood.py:32-39) needs.
// ccf_fit.py:349-354 (chi2), :166-193 (data vector), :195-260 (bracket), :444-481 (log det, forms, guards).
__device__ __forceinline__ void like_point_workgroup(const LikeArgs& a, long long point, double beta, double* th, double* red) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int tid = late_tid();
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  // Interval searches on the (increasing) beta grids as counts over the threads - one grid value per thread, one
  // barrier each - instead of loops whose loads the compiler keeps in program order (31 dependent round trips each for
  // BOSS).  Grids longer than the workgroup fall back to the loops.
  const bool by_count = a.n_beta_d <= kBlock && a.n_beta_c <= kBlock;
  int k = 0;                              // PCHIP piece of the data vector: last i in [1, n-2] with beta >= beta_d[i], else 0
  int lo = 0;
  double t = 0.0;
  if (by_count) {
    if (a.n_beta_d > 0) k = block_count(tid >= 1 && tid < a.n_beta_d - 1 && beta >= a.beta_d[tid < a.n_beta_d ? tid : 0], red);
    if (a.n_beta_c > 0) {
      const double g = tid < a.n_beta_c ? a.beta_c[tid] : inf;
      const int n_lt = block_count(tid < a.n_beta_c && g < beta, red);
      const int n_eq = block_count(tid < a.n_beta_c && g == beta, red);
      const int last = a.n_beta_c - 1;
      if (beta != beta) {
        t = beta;                         // NaN beta: the blend weight of cov_bracket, i.e. the row reports (-inf, inf) in every K2 variant
      } else if (n_lt == 0 && !n_eq) {
        lo = 0;                           // below the grid: first slice
      } else if (n_lt == a.n_beta_c) {
        lo = last;                        // above the grid: last slice
      } else if (n_eq) {
        lo = n_lt;                        // exact grid value (ccf_fit.py:221-222)
      } else {
        lo = n_lt - 1;
        t = (beta - a.beta_c[lo]) / (a.beta_c[last] - a.beta_c[lo]);
      }
    }
  } else {
    for (int i = 1; i < a.n_beta_d - 1; ++i) k = (beta >= a.beta_d[i]) ? i : k;
    if (a.n_beta_c > 0) cov_bracket(a, beta, &lo, &t);
  }
  if (a.n_beta_d > 0) {
    const double db = beta - a.beta_d[k];
    const double* piece = a.data + (size_t)k * a.N * 4;
    for (int e = tid; e < a.N; e += kBlock) {
      const double* c = piece + (size_t)e * 4;
      th[e] -= fma(fma(fma(c[3], db, c[2]), db, c[1]), db, c[0]);
    }
  } else {
    for (int e = tid; e < a.N; e += kBlock) th[e] -= a.data[e];
  }
  const double* P0 = a.prec;
  const double* P1 = a.prec;
  if (a.n_beta_c > 0) {
    P0 = a.prec + (size_t)lo * a.N * a.N;
    P1 = a.prec + (size_t)(a.n_beta_c - 1) * a.N * a.N;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  const int rows = (a.N + kWaves - 1) / kWaves;
  const int r0 = wave * rows, r1 = min(a.N, r0 + rows);
  const double omt = 1.0 - t;
  double part = 0.0;
  if ((a.N & 1) == 0) {
    part = t != 0.0 ? like_quadratic<true>(a.N, P0, P1, t, th, r0, r1, lane) : like_quadratic<false>(a.N, P0, P1, t, th, r0, r1, lane);
  } else {
    for (int b = lane; b < a.N; b += 64) {
      double y = 0.0;
      for (int r = r0; r < r1; ++r) {
        const double p = t != 0.0 ? omt * P0[(size_t)r * a.N + b] + t * P1[(size_t)r * a.N + b] : P0[(size_t)r * a.N + b];
        y = fma(th[r], p, y);
      }
      part = fma(y, th[b], part);
    }
  }
  const double chisq = block_sum(part, red);
  double factor = 0.0;
  bool singular = false;
  if (a.n_beta_c > 0) {
    double ld = 0.0;
    int n_neg = 0, n_bad = 0;
    if (t != 0.0) {
      const double* ev = a.eig + (size_t)lo * a.N;
      for (int e = tid; e < a.N; e += kBlock) ld += logdet_term(fma(t, ev[e], omt), &n_neg, &n_bad);
    }
    ld = block_sum(ld, red);
    const int neg = block_count(n_neg & 1, red);       // parity of the number of negative factors
    const int bad = block_count(n_bad != 0, red);
    singular = (neg & 1) || bad || !(fabs(a.logdet[lo]) < inf);
    factor = -0.5 * (a.logdet[lo] + ld);
  }
  if (tid == 0) {
    double lnl = like_form(a, chisq, factor);
    double chi_out = chisq;
    if (singular || lnl != lnl) {  // ccf_fit.py:448-450, 477-481
      lnl = -inf;
      chi_out = inf;
    }
    if (a.lnl) a.lnl[point] = lnl;
    if (a.chi2) a.chi2[point] = chi_out;
  }
}

// K2 "wide": one workgroup per point (small batches, and the theory kernels that do not carry the fused tail)
__global__ __launch_bounds__(kBlock, 4) void vk_like_wide_kernel(LikeArgs a) {
  extern __shared__ double lds[];
  double* th = lds;
  double* red = lds + ((a.N + 1) & ~1);
  for (long long point = blockIdx.x; point < a.n; point += gridDim.x) {
    __syncthreads();
    for (int e = threadIdx.x; e < a.N; e += kBlock) th[e] = a.theory[point * a.N + e];
    __syncthreads();
    like_point_workgroup(a, point, a.params[point * VK_NPAR + VK_P_BETA], th, red);
  }
}

// --------------------------------------------------------------------------------------------------
// K2: chi-square and log-likelihood, one wave per parameter point
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void vk_like_kernel(LikeArgs a) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  double* res = lds + (size_t)wave * a.N;
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  for (long long point = (long long)blockIdx.x * kWaves + wave; point < a.n;
       point += (long long)gridDim.x * kWaves) {
    const double beta = a.params[point * VK_NPAR + VK_P_BETA];
    const double* th = a.theory + point * a.N;
    // residual against the (beta-interpolated) data vector, ccf_fit.py:166-193,323
    if (a.n_beta_d > 0) {
      int k = 0;
      for (int i = 1; i < a.n_beta_d - 1; ++i) k = (beta >= a.beta_d[i]) ? i : k;
      const double db = beta - a.beta_d[k];
      const double* piece = a.data + (size_t)k * a.N * 4;
      for (int e = lane; e < a.N; e += 64) {
        const double* c = piece + (size_t)e * 4;
        res[e] = th[e] - fma(fma(fma(c[3], db, c[2]), db, c[1]), db, c[0]);
      }
    } else {
      for (int e = lane; e < a.N; e += 64) res[e] = th[e] - a.data[e];
    }
    // precision / covariance bracket, ccf_fit.py:213-228,245-260 (upper bracket = LAST grid entry)
    int lo = 0;
    double t = 0.0;
    const double* P0 = a.prec;
    const double* P1 = a.prec;
    if (a.n_beta_c > 0) {
      cov_bracket(a, beta, &lo, &t);
      P0 = a.prec + (size_t)lo * a.N * a.N;
      P1 = a.prec + (size_t)(a.n_beta_c - 1) * a.N * a.N;
    }
    __builtin_amdgcn_wave_barrier();
    // chi2 = sum_b (sum_a r_a P_ab) r_b with lanes over b (coalesced rows of P), ccf_fit.py:354
    double part = 0.0;
    const double omt = 1.0 - t;
    for (int b = lane; b < a.N; b += 64) {
      double y = 0.0;
      if (t != 0.0) {
        for (int r = 0; r < a.N; ++r) {
          const double p = omt * P0[(size_t)r * a.N + b] + t * P1[(size_t)r * a.N + b];
          y = fma(res[r], p, y);
        }
      } else {
        for (int r = 0; r < a.N; ++r) y = fma(res[r], P0[(size_t)r * a.N + b], y);
      }
      part = fma(y, res[b], part);
    }
    const double chisq = wave_sum(part);
    // -1/2 log det of the blended covariance, ccf_fit.py:445-451 (sign test as np.linalg.slogdet: see logdet_term)
    double factor = 0.0;
    bool singular = false;
    if (a.n_beta_c > 0) {
      double ld = 0.0;
      int n_neg = 0, n_bad = 0;
      if (t != 0.0) {
        const double* ev = a.eig + (size_t)lo * a.N;
        for (int e = lane; e < a.N; e += 64) ld += logdet_term(fma(t, ev[e], omt), &n_neg, &n_bad);
        ld = wave_sum(ld);
      }
      const int neg = __popcll(__ballot(n_neg & 1));
      singular = (neg & 1) || __any(n_bad) || !(fabs(a.logdet[lo]) < inf);
      factor = -0.5 * (a.logdet[lo] + ld);
    }
    double lnl = like_form(a, chisq, factor);
    double chi_out = chisq;
    if (singular || lnl != lnl) {  // ccf_fit.py:448-450, 477-481
      lnl = -inf;
      chi_out = inf;
    }
    if (lane == 0) {
      if (a.lnl) a.lnl[point] = lnl;
      if (a.chi2) a.chi2[point] = chi_out;
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// --------------------------------------------------------------------------------------------------
// K2 for a fixed covariance: T points per wave share every load of the precision matrix.
// The per-point kernel above reads all N^2 elements of P from L2 for every point (115 KB at N = 120: 7.5 GB per 65536
// batch, which is what bounds it); here a wave keeps the residuals of T points in LDS as res[a][p], streams each row of
// P once (lanes over columns, coalesced) and feeds T accumulators from it.
// --------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(kBlock) void vk_like_tiled_kernel(LikeArgs a) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  double* res = lds + (size_t)wave * a.N * T;          // res[a * T + p]
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  const long long tiles = (a.n + T - 1) / T;
  for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < tiles; tile += (long long)gridDim.x * kWaves) {
    const long long p0 = tile * T;
    // residuals of the tile's points (data vector possibly PCHIP-interpolated in beta, ccf_fit.py:166-193)
    for (int p = 0; p < T; ++p) {
      const long long point = (p0 + p < a.n) ? p0 + p : a.n - 1;
      const double* th = a.theory + point * a.N;
      if (a.n_beta_d > 0) {
        const double beta = a.params[point * VK_NPAR + VK_P_BETA];
        int k = 0;
        for (int i = 1; i < a.n_beta_d - 1; ++i) k = (beta >= a.beta_d[i]) ? i : k;
        const double db = beta - a.beta_d[k];
        const double* piece = a.data + (size_t)k * a.N * 4;
        for (int e = lane; e < a.N; e += 64) {
          const double* c = piece + (size_t)e * 4;
          res[e * T + p] = th[e] - fma(fma(fma(c[3], db, c[2]), db, c[1]), db, c[0]);
        }
      } else {
        for (int e = lane; e < a.N; e += 64) res[e * T + p] = th[e] - a.data[e];
      }
    }
    __builtin_amdgcn_wave_barrier();
    double part[T];
#pragma unroll
    for (int p = 0; p < T; ++p) part[p] = 0.0;
    for (int b = lane; b < a.N; b += 64) {
      double y[T];
#pragma unroll
      for (int p = 0; p < T; ++p) y[p] = 0.0;
      for (int r = 0; r < a.N; ++r) {
        const double pab = a.prec[(size_t)r * a.N + b];
        const double* rr = res + r * T;
#pragma unroll
        for (int p = 0; p < T; ++p) y[p] = fma(rr[p], pab, y[p]);
      }
      const double* rb = res + b * T;
#pragma unroll
      for (int p = 0; p < T; ++p) part[p] = fma(y[p], rb[p], part[p]);
    }
    double mine = 0.0;                                 // lane p keeps the chi-square of point p0 + p
#pragma unroll
    for (int p = 0; p < T; ++p) {
      const double c = wave_sum(part[p]);
      mine = (lane == p) ? c : mine;
    }
    if (lane < T && p0 + lane < a.n) {
      const double chisq = mine;
      double lnl = like_form(a, chisq, 0.0);
      double chi_out = chisq;
      if (lnl != lnl) {  // ccf_fit.py:477-481
        lnl = -inf;
        chi_out = inf;
      }
      if (a.lnl) a.lnl[p0 + lane] = lnl;
      if (a.chi2) a.chi2[p0 + lane] = chi_out;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace vk
