// vk_kernel_cells.h: cells theory kernel (per-point tables, velocity loop innermost) - part of libvictor_hip.so (see victor_hip.hip for the overview and DESIGN.md section 5).
#pragma once
#include "vk_kernel_fast.h"

namespace vk {

// --------------------------------------------------------------------------------------------------
// K1 "cells" variant: the lanes kernel's inner loop for per-point tables (reconstruction beta, BOSS).
// One workgroup owns one parameter point (its xi^r records are rebuilt in LDS as in the point-major kernel); each
// of its four waves owns the s bins j = wave, wave+4, ... and spreads the (s bin, mu) cells of those bins over its
// lanes, with the 50 velocity nodes as the inner, wave-uniform loop.  Like the lanes kernel this forms s_perp and
// s_par once per cell, reads x_k, w_k through the scalar cache and closes the v sum before the projection, so the
// integrand costs the same ~56 instructions; the projection sum over mu is a two-segment wave reduction per trip
// (a wave's 64 cells straddle at most two s bins when n_mu >= 64), accumulated by lane 0 in wave-private LDS.
// --------------------------------------------------------------------------------------------------
struct CellsPlan {
  int mu, w, s, betar, acc, da, total;
};

__host__ __device__ inline CellsPlan make_cells_plan(int n_mu, int n_x, int n_s, int uni_n, int nlr, int n_beta_r,
                                                     int lut_n, int with_da) {
  CellsPlan p;
  int o = fast_fixed_doubles(uni_n, nlr, lut_n);          // exp table + records first (fixed offsets)
  o = (o + 1) & ~1;
  p.mu = o;    o += 2 * n_mu;                      // {mu_i, sqrt(1 - mu_i^2)}
  p.w = o;     o += kMaxEll * n_mu;                // W_l[i]
  p.s = o;     o += (n_s + 1) & ~1;
  p.betar = o; o += (n_beta_r + 1) & ~1;
  p.acc = o;   o += kMaxEll * ((n_s + kWaves - 1) / kWaves) * kWaves;   // [l][slot][wave]
  o = (o + 1) & ~1;
  p.da = o;    o += with_da ? uni_n * 4 : 0;      // Da table of the dispersion model
  p.total = o;
  return p;
}

// 5 workgroups per CU for the streaming mode (<= 96 VGPRs, a few dwords of scratch outside the hot loop); the from_data
// and dispersion modes need more registers and run 4 per CU without spills
template <int NLR, int NL, int GRID, int MODE>
__global__ __launch_bounds__(kBlock, MODE == kModeStreaming ? 5 : 4) void vk_theory_cells_kernel(TheoryArgs a) {
  extern __shared__ double lds[];
  const CellsPlan pl = make_cells_plan(a.n_mu, a.n_x, a.n_s, a.uni_n, NLR, a.n_beta_r, a.uni_lut_n, mode_is_dispersion(MODE));
  const int tid = threadIdx.x;
  for (int i = tid; i < a.n_mu; i += kBlock) {
    const double m = a.mu[i];
    lds[pl.mu + 2 * i] = m;
    lds[pl.mu + 2 * i + 1] = sqrt(1.0 - m * m);
#pragma unroll
    for (int l = 0; l < kMaxEll; ++l) lds[pl.w + l * a.n_mu + i] = (l < NL) ? a.w_ell[l * a.n_mu + i] : 0.0;
  }
  for (int j = tid; j < a.n_s; j += kBlock) lds[pl.s + j] = a.s[j];
  stage_uni_records<NLR>(a, lds);
  if (mode_is_dispersion(MODE)) stage_da<NLR>(a, lds + pl.da);
  if (a.n_beta_r > 0)
    for (int i = tid; i < a.n_beta_r; i += kBlock) lds[pl.betar + i] = a.beta_r[i];
  const FastConsts fc = make_fast_consts<NLR>(a);
  __syncthreads();

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const double* l_mu = lds + pl.mu;
  const double* l_w = lds + pl.w;
  typedef const vk_d2 __attribute__((address_space(4))) * cvec_ptr;
  const cvec_ptr cxw = (cvec_ptr)(unsigned long long)a.xw_scaled;
  const double* l_s = lds + pl.s;
  const int slots = (a.n_s + kWaves - 1) / kWaves;        // s bins per wave (upper bound)
  double* l_acc = lds + pl.acc;                            // [l][slot][wave]: each entry touched by one wave only
  const int my_bins = (a.n_s - wave + kWaves - 1) / kWaves;  // bins j = wave + 4*jj, jj < my_bins
  const int cells = my_bins * a.n_mu;

  double wsum[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    double t = 0.0;
    for (int i = lane; i < a.n_mu; i += 64) t += l_w[l * a.n_mu + i];
    wsum[l] = wave_sum(t);
  }

  for (long long point = blockIdx.x; point < a.n; point += gridDim.x) {
    const double* row = a.params + point * VK_NPAR;
    const PointScalars ps = point_scalars(a, row);
    if (a.n_beta_r > 0 || a.empirical) {
      __syncthreads();  // every wave is done with the previous point's records
      if (a.n_beta_r > 0) rebuild_uni_xi<NLR>(a, lds + kRecsOff, lds + pl.betar, row[VK_P_BETA]);
      if (a.empirical) rebuild_uni_v_emp<NLR>(a, lds + kRecsOff, ps.av);
      if (mode_is_dispersion(MODE) && a.empirical) rebuild_da_emp(a, lds + pl.da, ps.av);
      __syncthreads();
    }
    for (int e = lane; e < kMaxEll * slots; e += 64) l_acc[e * kWaves + wave] = 0.0;
    const FastPoint fp = make_fast_point(ps, fc);
    for (int base = 0; base < cells; base += 64) {
      const int e = base + lane;
      const bool live = e < cells;
      const int ec = live ? e : cells - 1;
      const int jj = ec / a.n_mu;
      const int i = ec - jj * a.n_mu;
      const double sj = l_s[wave + kWaves * jj];
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
      // projection: this trip's cells belong to s bin jj0 or jj0 + 1
      const int jj0 = __builtin_amdgcn_readfirstlane(jj);
      const bool first = (jj == jj0);
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const double v = l_w[l * a.n_mu + i] * g;
        const double s0 = wave_sum(first ? v : 0.0);
        const double s1 = wave_sum(first ? 0.0 : v);
        if (lane == 0) {
          l_acc[(l * slots + jj0) * kWaves + wave] += s0;
          if (jj0 + 1 < my_bins) l_acc[(l * slots + jj0 + 1) * kWaves + wave] += s1;
        }
      }
    }
    // the point's theory vector is complete in LDS once every wave has finished its s bins: write it as one
    // contiguous run (coalesced; 8-byte stores strided by the s-bin ownership of the waves cost 1.5x the bytes in HBM)
    __syncthreads();
    for (int e = tid; e < NL * a.n_s; e += kBlock) {
      const int l = e / a.n_s, j = e - l * a.n_s;
      double ws = wsum[0];
#pragma unroll
      for (int q = 1; q < NL; ++q) ws = (l == q) ? wsum[q] : ws;
      a.out[point * (long long)(a.n_ell * a.n_s) + e] =
          l_acc[(l * slots + j / kWaves) * kWaves + (j & (kWaves - 1))] - ws + ps.poison;
    }
    __syncthreads();      // l_acc is zeroed (and the records may be rebuilt) at the top of the next point
  }
}


}  // namespace vk
