// vk_kernel_like.h: chi-square / log-likelihood kernel - part of libvictor_hip.so (see victor_hip.hip for the overview and DESIGN.md section 5).
#pragma once
#include "vk_common.h"

namespace vk {

// --------------------------------------------------------------------------------------------------
// shared pieces of the likelihood stage
// --------------------------------------------------------------------------------------------------
// likelihood form (ccf_fit.py:455-473); `factor` = -1/2 log det of the covariance when it depends on beta, else 0
__device__ __forceinline__ double like_form(const LikeArgs& a, double chisq, double factor) {
  double nm = a.nmocks, np = a.nparams, nd = (double)a.N;
  // Everything below but chisq is the same for every point of a launch, and the compiler would evaluate it (four divisions) at
  // the top of the kernel and carry the results through the theory loops - in registers those loops need.  One thread
  // evaluates this once per point: keep it here.
  asm volatile("" : "+v"(nm), "+v"(np), "+v"(nd));
  if (a.like_form == VK_LIKE_SELLENTIN) return -nm * log(1.0 + chisq / (nm - 1.0)) / 2.0 + factor;
  if (a.like_form == VK_LIKE_HARTLAP) return -0.5 * chisq * ((nm - nd - 2.0) / (nm - 1.0)) + factor;
  if (a.like_form == VK_LIKE_PERCIVAL) {
    const double B = (nm - nd - 2.0) / ((nm - nd - 1.0) * (nm - nd - 4.0));
    const double m = np + 2.0 + (nm - 1.0 + B * (nd - np)) / (1.0 + B * (nd - np));
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

// (No __syncthreads_count anywhere in these kernels: HIP implements it through a static __shared__ word, which moves the
// base of dynamic LDS off zero for the whole kernel - every LDS address of the theory loops then carries an addend the ds_read
// offset field could have held.  The interval searches of the tail count per wave with ballots, see LikePrefetch::issue.)

// four sums over the workgroup in one pass (two barriers instead of eight); `red`: kLikeRed doubles
__device__ __forceinline__ void block_sum4(double (&v)[4], double* red) {
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = wave_sum(v[q]);
  __syncthreads();                       // `red` may still be read from a previous call
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) red[q * kWaves + (threadIdx.x >> 6)] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double s = red[q * kWaves];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) s += red[q * kWaves + w];
    v[q] = s;
  }
}

// --------------------------------------------------------------------------------------------------
// chi2 / lnL of ONE point by a whole workgroup (kBlock threads): the fused tail of the point-major and cells kernels and
// the "wide" K2.  ccf_fit.py:349-354 (chi2), :166-193 (data vector), :195-260 (bracket), :444-481 (log det, forms, guards).
//
// A launch starts with cold L2s (each XCD's is invalidated at the kernel boundary), so every dependent load of the tail is
// a trip to the memory side - ~0.9 us each on MI355X (tools/gpu_phases.py) - and a single-point launch, the reference's
// calling convention (CCFLikelihood.py:32-39), is little else: partial sums, beta grids, data vector, precision slices and
// log-det factors in sequence were 5-6 us of its 14.  The tail is therefore split in two:
//   LikePrefetch::issue   everything that depends on beta alone - the interval searches, this thread's entry of the data
//                         vector, the log-det factors of the blended covariance, the first rows of the quadratic form -
//                         is requested BEFORE the theory vector is gathered, and travels with it;
//   like_point_workgroup  residual, quadratic form, reductions, likelihood form, once the theory vector sits in LDS.
//
// The quadratic form on the folded DIAGONALS.  With M = N rounded up to even (an odd N gets a zero row and column; vk_create) the
// host folds every precision slice onto its upper triangle, T_ii = P_ii, T_ij = P_ij + P_ji, and stores it by circular
// diagonals: row k (k = 0 .. M/2) holds D_k[i] = T[i][(i + k) mod M] for i = 0 .. M - 1 (row M/2: i < M/2 only, zeros behind -
// every unordered pair exactly once), M + 2 doubles per row with the padding.  Then
//     chi2 = sum_i r_i t_i,   t_i = sum_k D_k[i] r_((i + k) mod M),
// half the bytes of the slice, M/2 + 1 rows of EQUAL length, and the factor of entry i of row k is the residual k places on:
// with the residual stored twice in LDS (r2 = r | r) a lane's two factors are ONE ds_read2_b64 at an immediate offset and
// its two entries two fmas - three instructions per row where the triangle's two-rows-in-one layout (round 3) cost 22 (a
// select between "row c" and "row M - 1 - c" per entry, four LDS reads with computed addresses): the tail of a single
// point's launch spent 1.2 us of its 2.2 issuing them from one wave per SIMD.  Waves take the rows round-robin, lanes the entry
// pairs (16-byte loads; more than 64 pairs - N > 126 - in chunks); RB rows per lane are in flight (4 RB registers; blended
// slices: RB / 2 of each), the first RB from the prefetch: N = 120 (61 rows, 16 for the first wave) in ONE batch.
// --------------------------------------------------------------------------------------------------
// LDS of a fused tail / the wide K2: residual twice over (2 M doubles) and kLikeSlack zeros behind it - the rows a wave's last
// batch holds beyond the last diagonal are zeros whose factors are read up to kWaves (rows in flight - 1) + 1 places further
// on, and those must be finite -, kLikeRed of reduction scratch, 4 more (completion flag)
constexpr int kLikeSlack = 4 * 16 + 4;       // kWaves * kLikeRows + 4 (static_assert below)
__host__ __device__ constexpr int like_red_off(int N) { return 2 * ((N + 1) & ~1) + kLikeSlack; }
__host__ __device__ constexpr int like_lds_doubles(int N) { return like_red_off(N) + kLikeRed + 4; }
__host__ __device__ constexpr size_t like_slice_doubles(int N) { return (size_t)((((N + 1) & ~1) >> 1) + 1) * (((N + 1) & ~1) + 2); }

typedef double like_d2 __attribute__((ext_vector_type(2)));

template <int RB>
struct LikePrefetch {
  int k, lo;              // PCHIP piece of the data vector; lower slice of the covariance bracket
  double t;               // blend weight of the bracket (0: slice `lo` alone; NaN beta: NaN)
  double db;              // beta - beta_d[k]
  bool data_in_regs;      // N <= kBlock: this thread's data-vector entry is in c[]
  double c[4];            // data-vector entry of element `tid`: PCHIP coefficients in db (fixed data vector: c[0] alone)
  double ev_mine;         // eigenvalue `tid` of C_lo^-1 C_last (blended covariance), else 1
  const double* T0;
  const double* T1;
  like_d2 rows[RB];       // t == 0: rows wave, wave + 4, ... of T0; else RB / 2 rows of T0, then the same rows of T1

  // rows (diagonals) c0, c0 + kWaves, ... of this wave, entry pair e0 / 2 of this lane; M = N rounded up to even
  template <bool BLEND>
  __device__ __forceinline__ void load_rows(int M, int c0, int e0) {
    constexpr int R = BLEND ? RB / 2 : RB;
    const int n_rows = (M >> 1) + 1, W = M + 2;
    const char* b0 = reinterpret_cast<const char*>(T0);   // a slice is far below 4 GB: 32-bit byte offsets from the uniform bases
    const char* b1 = reinterpret_cast<const char*>(T1);
    unsigned off = ((unsigned)c0 * (unsigned)W + (unsigned)e0) * 8u;
    const unsigned stride = (unsigned)(kWaves * W) * 8u;
#pragma unroll
    for (int s = 0; s < R; ++s) {
      const bool live = c0 + kWaves * s < n_rows && e0 < W;
      rows[s] = live ? *reinterpret_cast<const like_d2*>(b0 + off) : like_d2{0.0, 0.0};
      if (BLEND) rows[R + s] = live ? *reinterpret_cast<const like_d2*>(b1 + off) : like_d2{0.0, 0.0};
      off += stride;
    }
  }

  // no barrier, no LDS writes: may be called by every thread of the workgroup at any point before like_point_workgroup.
  // `lds_beta_r`: the kernel's LDS copy of the real-space tables' beta grid, or null (see LikeArgs::grids_in_lds)
  __device__ __forceinline__ void issue(const LikeArgs& a, double beta, int tid, const double* lds_beta_r = nullptr) {
    const double* grid_d = (lds_beta_r && (a.grids_in_lds & 1)) ? lds_beta_r : a.beta_d;
    const double* grid_c = (lds_beta_r && (a.grids_in_lds & 2)) ? lds_beta_r : a.beta_c;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    // Interval searches on the (increasing) beta grids as COUNTS: every wave counts the whole grid by itself, lanes over the
    // grid values, a ballot per 64 of them - no barrier, the loads of both grids in flight together (loops over the grids
    // cost 31 dependent round trips each for BOSS, counts over the workgroup six barriers).
    const int lane = tid & 63;
    k = 0;                                  // last i in [1, n-2] with beta >= beta_d[i], else 0
    lo = 0;
    t = 0.0;
    int n_lt = 0, n_eq = 0;
    const int n_max = max(a.n_beta_d, a.n_beta_c);
    for (int base = 0; base < n_max; base += 64) {          // wave-uniform trip count (one trip for BOSS)
      const int i = base + lane;
      const double gd = (i < a.n_beta_d) ? grid_d[i] : inf;
      const double gc = (i < a.n_beta_c) ? grid_c[i] : inf;
      k += __popcll(__ballot(i >= 1 && i < a.n_beta_d - 1 && beta >= gd));
      n_lt += __popcll(__ballot(i < a.n_beta_c && gc < beta));
      n_eq += __popcll(__ballot(i < a.n_beta_c && gc == beta));
    }
    const int last = a.n_beta_c - 1;
    if (a.n_beta_c > 0) {
      if (beta != beta) {
        t = beta;                           // NaN beta: the blend weight of cov_bracket, i.e. the row reports (-inf, inf) in every K2 variant
      } else if (n_lt == 0 && !n_eq) {
        lo = 0;                             // below the grid: first slice
      } else if (n_lt == a.n_beta_c) {
        lo = last;                          // above the grid: last slice
      } else if (n_eq) {
        lo = n_lt;                          // exact grid value (ccf_fit.py:221-222)
      } else {
        lo = n_lt - 1;
        t = (beta - grid_c[lo]) / (grid_c[last] - grid_c[lo]);
      }
    }
    // log det of the blended covariance, det((1-t) C_lo + t C_last) = det(C_lo) prod_i (1 - t + t lambda_i) (ccf_fit.py:445-451):
    // this thread's eigenvalue is requested here; the logarithm is taken in like_point_workgroup, after everything has arrived
    ev_mine = (a.n_beta_c > 0 && t != 0.0 && tid < a.N) ? a.eig[(size_t)lo * a.N + tid] : 1.0;
    // the first rows of the quadratic form
    {
      const int M = (a.N + 1) & ~1;
      const size_t slice = like_slice_doubles(a.N);
      T0 = a.tri + (a.n_beta_c > 0 ? (size_t)lo * slice : 0);
      T1 = a.tri + (a.n_beta_c > 0 ? (size_t)last * slice : 0);
      if (t != 0.0) load_rows<true>(M, tid >> 6, 2 * lane); else load_rows<false>(M, tid >> 6, 2 * lane);
    }
    // this thread's entry of the data vector (ccf_fit.py:166-193)
    data_in_regs = a.N <= kBlock;
    db = a.n_beta_d > 0 ? beta - grid_d[k] : 0.0;
    c[0] = c[1] = c[2] = c[3] = 0.0;
    if (data_in_regs && tid < a.N) {
      if (a.n_beta_d > 0) {
        const double* p = a.data + ((size_t)k * a.N + tid) * 4;
        c[0] = p[0]; c[1] = p[1]; c[2] = p[2]; c[3] = p[3];
      } else {
        c[0] = a.data[tid];
      }
    }
  }

  // sum_{i <= j} T_ij r_i r_j, this thread's share; `r2` = the residual twice over in LDS (2 M doubles, r2[N] = r2[M + N] = 0
  // when N is odd).  The first batch of rows (of the first chunk of entry pairs) is in rows[] already (issue); further batches
  // are loaded here, one exposed round trip each.  A row beyond the last one holds zeros (load_rows) and reads residuals
  // that exist, so there is no branch per row; lanes beyond the last entry pair read the pair at M (zeros in the rows).
  template <bool BLEND>
  __device__ __forceinline__ double quadratic(const LikeArgs& a, const double* r2, int tid) {
    constexpr int R = BLEND ? RB / 2 : RB;
    const int M = (a.N + 1) & ~1, n_rows = (M >> 1) + 1, W = M + 2;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double omt = 1.0 - t;
    double part = 0.0;
    for (int e0 = 2 * (tid & 63); e0 < W; e0 += 128) {     // one chunk of 64 entry pairs up to N = 126
      const int ec = min(e0, M);                           // (entries M, M + 1 are the padding: zeros)
      double t0 = 0.0, t1 = 0.0;
      for (int base = wave; base < n_rows; base += kWaves * R) {
        if (e0 >= 128 || base != wave) load_rows<BLEND>(M, base, e0);
        // row k = base + kWaves s: this lane's factors are r2[ec + k], r2[ec + k + 1] - immediate offsets from one address
        // (k < M/2 + 1 + kWaves (R - 1): inside the doubled residual and its kLikeSlack zeros)
        const double* rl = r2 + ec + base;
#pragma unroll
        for (int s = 0; s < R; ++s) {
          const double x = BLEND ? omt * rows[s].x + t * rows[R + s].x : rows[s].x;
          const double y = BLEND ? omt * rows[s].y + t * rows[R + s].y : rows[s].y;
          t0 = fma(x, rl[kWaves * s], t0);
          t1 = fma(y, rl[kWaves * s + 1], t1);
        }
      }
      part = fma(t0, r2[ec], part);
      part = fma(t1, r2[ec + 1], part);
    }
    return part;
  }
};

constexpr int kLikeRows = 16;
static_assert(kLikeSlack >= kWaves * kLikeRows + 2, "kLikeSlack: zeros behind the doubled residual (vk_kernel_like.h)");
constexpr int kLikeRows_doc = 0;  // rows in flight in the point-major kernel and the wide K2, where the tail's latency is the launch's: N = 120
                               // (15 rows per wave) in ONE batch.  64 registers: affordable since those kernels lost their
                               // grid-stride loops (128 in all).  The cells kernel (96 registers, tail never latency-critical:
                               // it fuses batches of 24-512 points) keeps 4 in flight.
constexpr int kLikeRowsCells = 4;
typedef LikePrefetch<kLikeRows> LikePre;

#ifdef VK_PHASES
#define VK_LIKE_STAMP(a, k) do { if ((a).stamps && threadIdx.x == 0 && blockIdx.x < 4096) (a).stamps[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define VK_LIKE_STAMP(a, k) do { } while (0)
#endif

// On entry `th` (LDS, N doubles) holds the point's theory vector, the workgroup is synchronised and `pf` was issued for
// this point; `th` is overwritten with the residual.  `red`: kLikeRed doubles of LDS.
template <int RB>
__device__ __forceinline__ void like_point_workgroup(const LikeArgs& a, long long point, double beta, double* th, double* red,
                                                     LikePrefetch<RB>& pf) {
  const int tid = late_tid();
  VK_LIKE_STAMP(a, 8);
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  const double t = pf.t;
  const int lo = pf.lo;
  // residual against the (beta-interpolated) data vector, ccf_fit.py:166-193, 323 - stored twice, M = N rounded up to even
  // places apart (LikePrefetch::quadratic reads r[(i + k) mod M] as r2[i + k]), with zeros behind (kLikeSlack)
  const int M = (a.N + 1) & ~1;
  if (pf.data_in_regs) {
    if (tid < a.N) {
      const double v = th[tid] - fma(fma(fma(pf.c[3], pf.db, pf.c[2]), pf.db, pf.c[1]), pf.db, pf.c[0]);
      th[tid] = v;
      th[M + tid] = v;
    }
  } else if (a.n_beta_d > 0) {
    const double* piece = a.data + (size_t)pf.k * a.N * 4;
    for (int e = tid; e < a.N; e += kBlock) {
      const double* c = piece + (size_t)e * 4;
      const double v = th[e] - fma(fma(fma(c[3], pf.db, c[2]), pf.db, c[1]), pf.db, c[0]);
      th[e] = v;
      th[M + e] = v;
    }
  } else {
    for (int e = tid; e < a.N; e += kBlock) {
      const double v = th[e] - a.data[e];
      th[e] = v;
      th[M + e] = v;
    }
  }
  if ((a.N & 1) && tid == 0) th[a.N] = th[M + a.N] = 0.0;    // the zero row / column that makes an odd N even
  if (tid >= kBlock - kLikeSlack) th[2 * M + (kBlock - 1 - tid)] = 0.0;     // (threads from the far end: the first N are busy above)
  __syncthreads();
  VK_LIKE_STAMP(a, 10);
  double sums[4] = {0.0, 0.0, 0.0, 0.0};               // chi2 share, log |factors|, negative factors, zero / NaN factors
  if (a.n_beta_c > 0 && t != 0.0) {
    int neg = 0, bad = 0;
    if (tid < a.N) sums[1] = logdet_term(fma(t, pf.ev_mine, 1.0 - t), &neg, &bad);
    for (int e = tid + kBlock; e < a.N; e += kBlock) sums[1] += logdet_term(fma(t, a.eig[(size_t)lo * a.N + e], 1.0 - t), &neg, &bad);
    sums[2] = (double)neg;
    sums[3] = (double)bad;
  }
  sums[0] = t != 0.0 ? pf.template quadratic<true>(a, th, tid) : pf.template quadratic<false>(a, th, tid);
  VK_LIKE_STAMP(a, 11);
  double chisq, factor = 0.0;
  bool singular = false;
  if (a.n_beta_c > 0) {
    block_sum4(sums, red);
    chisq = sums[0];
    const int neg = (int)sums[2], bad = (int)sums[3];   // np.linalg.slogdet's sign: parity of the negative factors (logdet_term)
    singular = (neg & 1) || bad || !(fabs(a.logdet[lo]) < inf);
    factor = -0.5 * (a.logdet[lo] + sums[1]);
  } else {
    chisq = block_sum(sums[0], red);
  }
  VK_LIKE_STAMP(a, 12);
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

#ifndef VK_KERNEL_TEMPLATES_ONLY   // (the kernels that are not templates are defined in one translation unit only)
// K2 "wide": one workgroup per point (small batches, and the theory kernels that do not carry the fused tail)
__global__ __launch_bounds__(kBlock, 2) void vk_like_wide_kernel(LikeArgs a) {
  extern __shared__ double lds[];
  double* th = lds;
  double* red = lds + like_red_off(a.N);
  {
    const long long point = blockIdx.x;               // one point per workgroup (the host launches n of them)
    if (point >= a.n) return;
    const double beta = a.params[point * VK_NPAR + VK_P_BETA];
    LikePre pf;
    pf.issue(a, beta, late_tid());
    __syncthreads();
    for (int e = threadIdx.x; e < a.N; e += kBlock) th[e] = a.theory[point * a.N + e];
    __syncthreads();
    like_point_workgroup(a, point, beta, th, red, pf);
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


#endif  // VK_KERNEL_TEMPLATES_ONLY

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
