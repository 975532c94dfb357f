// vk_kernel_cells.h: cells theory kernel (per-point tables, velocity loop innermost) - part of libvictor_hip.so (see victor_hip.hip for the overview and DESIGN.md section 5).
#pragma once
#include "vk_kernel_fast.h"

namespace vk {

// --------------------------------------------------------------------------------------------------
// K1 "cells" variant: the lanes kernel's inner loop for per-point tables (reconstruction beta, BOSS).
// One workgroup owns one parameter point - or one of `parts` contiguous slices of its s bins - with the xi^r records
// rebuilt in LDS as in the point-major kernel.  The (s bin, mu) cells of the slice are flattened (s bin major) and dealt to
// the four waves in trips of 64 (wave w takes trips w, w + 4, ...: every wave gets the same number of trips to within one,
// whatever n_s is - with whole s bins per wave, 30 bins meant 8 + 8 + 7 + 7), lanes over the cells, the 50 velocity nodes
// as the inner, wave-uniform loop.  Like the lanes kernel this forms s_perp and s_par once per cell, reads x_k, w_k through
// the scalar cache and closes the v sum before the projection, so the integrand costs the same instructions; the
// projection sum over mu is a two-segment wave reduction per trip (a trip's 64 cells straddle at most two s bins when
// n_mu >= 64), accumulated by lane 0 in wave-private LDS and combined over the waves at the end.
// With one workgroup per point the finished theory vector sits in LDS and the chi-square is taken there (`fuse`).
// --------------------------------------------------------------------------------------------------
struct CellsPlan {
  int mu, w, s, betar, da, image_end, acc, like, total;
};

__host__ __device__ inline int cells_slice_bins(int n_s, int parts) { return (n_s + parts - 1) / parts; }

__host__ __device__ inline CellsPlan make_cells_plan(int n_mu, int n_x, int n_s, int uni_n, int nlr, int n_beta_r,
                                                     int lut_n, int with_da, int parts, int n_like) {
  CellsPlan p;
  int o = fast_fixed_doubles(uni_n, nlr, lut_n);          // exp table + records first (fixed offsets)
  o = (o + 1) & ~1;
  p.mu = o;    o += 2 * n_mu;                      // {mu_i, sqrt(1 - mu_i^2)}
  p.w = o;     o += kMaxEll * n_mu;                // W_l[i]
  p.s = o;     o += (n_s + 1) & ~1;
  p.betar = o; o += (n_beta_r + 1) & ~1;
  p.da = o;    o += with_da ? uni_n * 4 : 0;      // Da table of the dispersion model
  o = (o + 1) & ~1;
  p.image_end = o;                                 // batch-constant up to here (LDS image, see vk_kernel_fast.h)
  p.acc = o;   o += kMaxEll * (cells_slice_bins(n_s, parts) + 1) * kWaves;   // [l][local bin][wave] (+ one spill bin)
  o = (o + 1) & ~1;
  p.like = o;  o += n_like > 0 ? like_lds_doubles(n_like) : 0;
  p.total = o;
  return p;
}

template <int NLR>
__device__ __forceinline__ void stage_cells(const TheoryArgs& a, const CellsPlan& pl, double* lds, bool with_da) {
  const int tid = threadIdx.x;
  for (int i = tid; i < a.n_mu; i += kBlock) {
    const double m = a.mu[i];
    lds[pl.mu + 2 * i] = m;
    lds[pl.mu + 2 * i + 1] = sqrt(1.0 - m * m);
#pragma unroll
    for (int l = 0; l < kMaxEll; ++l) lds[pl.w + l * a.n_mu + i] = (l < a.n_ell) ? a.w_ell[l * a.n_mu + i] : 0.0;
  }
  for (int j = tid; j < a.n_s; j += kBlock) lds[pl.s + j] = a.s[j];
  stage_uni_records<NLR>(a, lds);
  if (with_da) stage_da<NLR>(a, lds + pl.da);
  if (a.n_beta_r > 0)
    for (int i = tid; i < a.n_beta_r; i += kBlock) lds[pl.betar + i] = a.beta_r[i];
}

// 5 workgroups per CU for the streaming mode (<= 96 VGPRs); the from_data and dispersion modes need more registers and
// run 4 per CU without spills
template <int NLR, int NL, int GRID, int MODE>
__global__ __launch_bounds__(kBlock, MODE == kModeStreaming ? 5 : 4) void vk_theory_cells_kernel(TheoryArgs a) {
  extern __shared__ double lds[];
  warm_kernarg_lines<sizeof(TheoryArgs)>();
  const int N = a.n_ell * a.n_s;
  const int S = a.parts;
  const bool tail = a.fuse || S > 1;
  const CellsPlan pl = make_cells_plan(a.n_mu, a.n_x, a.n_s, a.uni_n, NLR, a.n_beta_r, a.uni_lut_n, mode_is_dispersion(MODE), S,
                                       tail ? N : 0);
  const int tid = threadIdx.x;
  if (a.image) copy_image(lds, a.image, pl.image_end);
  else stage_cells<NLR>(a, pl, lds, mode_is_dispersion(MODE));
  const FastConsts fc = make_fast_consts<NLR>(a);
  __syncthreads();

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const double* l_mu = lds + pl.mu;
  const double* l_w = lds + pl.w;
  typedef const vk_d2 __attribute__((address_space(4))) * cvec_ptr;
  const cvec_ptr cxw = (cvec_ptr)(unsigned long long)a.xw_scaled;
  const double* l_s = lds + pl.s;
  const int slots = cells_slice_bins(a.n_s, S) + 1;          // local bins of a slice (+ one that only ever receives zeros)
  double* l_acc = lds + pl.acc;                              // [l][local bin][wave]: each entry touched by one wave only
  const unsigned items = (unsigned)a.n * (unsigned)S;                // the host keeps n * parts below 2^31

  for (unsigned item = blockIdx.x; item < items; item += gridDim.x) {
    const long long point = item / (unsigned)S;
    const int q = (int)(item - (unsigned)point * (unsigned)S);
    const int j0 = (int)((long long)a.n_s * q / S), j1 = (int)((long long)a.n_s * (q + 1) / S);   // this slice's s bins
    const int cells = (j1 - j0) * a.n_mu;
    const double* row = a.params + point * VK_NPAR;
    const PointScalars ps = point_scalars(a, row);
    __syncthreads();      // every wave is done with the previous item's records and accumulators
    if (a.n_beta_r > 0 || a.empirical) {
      if (a.n_beta_r > 0) rebuild_uni_xi<NLR>(a, lds + kRecsOff, lds + pl.betar, row[VK_P_BETA]);
      if (a.empirical) rebuild_uni_v_emp<NLR>(a, lds + kRecsOff, ps.av);
      if (mode_is_dispersion(MODE) && a.empirical) rebuild_da_emp(a, lds + pl.da, ps.av);
    }
    for (int e = lane; e < kMaxEll * slots; e += 64) l_acc[e * kWaves + wave] = 0.0;
    __syncthreads();
    const FastPoint fp = make_fast_point(ps, fc);
    for (int base = 64 * wave; base < cells; base += 64 * kWaves) {
      const int e = base + lane;
      const bool live = e < cells;
      const unsigned ec = (unsigned)(live ? e : cells - 1);
      const int jj = (int)__umulhi(ec, a.nmu_magic);           // ec / n_mu
      const int i = (int)ec - jj * a.n_mu;
      const double sj = l_s[j0 + jj];
      const vk_d2 mm = *reinterpret_cast<const vk_d2*>(l_mu + 2 * i);
      const double s_perp = sj * fp.k_perp * mm.y;
      const double sperp2 = s_perp * s_perp;
      const double sperp2x = sperp2 * fp.fp2;      // from_data only
      const double s_par = sj * fp.k_par * mm.x;
      double g = 0.0;
      for (int k = 0; k < a.n_x; ++k) {
        const vk_d2 xw = cxw[k];             // scalar-cache read (wave-uniform), see vk_kernel_lanes.h
        const double xk = xw.x;
        const double num = fma(-xk, fp.Bk, s_par);
        g = fma(xw.y,
                mode_is_dispersion(MODE)
                    ? disp_value<NLR, GRID, MODE == kModeDispersionFromData>(lds, lds + pl.da, fc, fp, a.niter, num, s_par, sperp2,
                                                                           xk)
                    : uni_value<NLR, GRID, MODE == kModeFromData>(lds, fc, fp.AVk, num, sperp2, xk, fp.fa, sperp2x),
                g);
      }
      if (!live) g = 0.0;
      // projection: this trip's cells belong to local bin jj0 or jj0 + 1 (the latter may be the spill bin)
      const int jj0 = __builtin_amdgcn_readfirstlane(jj);
      const bool first = (jj == jj0);
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const double v = l_w[l * a.n_mu + i] * g;
        const double s0 = wave_sum(first ? v : 0.0);
        const double s1 = wave_sum(first ? 0.0 : v);
        if (lane == 0) {
          l_acc[(l * slots + jj0) * kWaves + wave] += s0;
          l_acc[(l * slots + jj0 + 1) * kWaves + wave] += s1;
        }
      }
    }
    // the slice's part of the theory vector is complete in LDS once every wave has finished its trips
    __syncthreads();
    const int nb = j1 - j0;
    double* th = lds + pl.like;
    for (int e = tid; e < NL * nb; e += kBlock) {
      const int l = e / nb, jl = e - l * nb;
      const double* pa = l_acc + (l * slots + jl) * kWaves;
      const double ws = l == 0 ? a.wsum[0] : (l == 1 ? a.wsum[1] : a.wsum[2]);
      const double v = ((pa[0] + pa[1]) + (pa[2] + pa[3])) - ws + ps.poison;
      double* dst = a.out + point * (long long)N + l * a.n_s + j0 + jl;
      if (S > 1) store_shared(dst, v); else *dst = v;
      if (tail && S == 1) th[l * a.n_s + jl] = v;
    }
    if (tail) {
      // fused / split launches run one item per workgroup and leave from here (see vk_kernel_fast.h)
      if (S == 1) {
        __syncthreads();
        like_point_workgroup(a.like, point, row[VK_P_BETA], th, th + ((N + 1) & ~1));
      } else {
        int* flag = reinterpret_cast<int*>(th + ((N + 1) & ~1) + kWaves + 2);
        if (point_completed(a.counters, point, (unsigned)S, flag)) {
          finish_point<NL>(a, point, row[VK_P_BETA], ps.poison, th, false);
        }
      }
      return;
    }
  }
}


}  // namespace vk
