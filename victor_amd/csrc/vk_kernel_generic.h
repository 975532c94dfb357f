// vk_kernel_generic.h: generic theory kernel (every RSD model / option) and the xi(s,mu) kernel - part of libvictor_hip.so (see victor_hip.hip for the overview and DESIGN.md section 5).
#pragma once
#include "vk_common.h"

namespace vk {

struct LdsPlan {
  int mu, smu, w, x, wx, svk, svc, vrk, vrc, xik, xic, betar, red, etab, total;
};

__host__ __device__ inline LdsPlan make_plan(int n_mu, int n_x, int n_ell, int sv_int, int vr_int, int xi_int,
                                            int n_ell_r, int n_beta_r) {
  LdsPlan p;
  int o = 0;
  p.mu = o;   o += n_mu;
  p.smu = o;  o += n_mu;
  p.w = o;    o += n_ell * n_mu;
  p.x = o;    o += n_x;
  p.wx = o;   o += n_x;
  p.svk = o;  o += sv_int + 1;
  o = (o + 1) & ~1;
  p.svc = o;  o += sv_int * 4;
  p.vrk = o;  o += vr_int + 1;
  o = (o + 1) & ~1;
  p.vrc = o;  o += kVrVars * vr_int * 4;
  p.xik = o;  o += xi_int + 1;
  o = (o + 1) & ~1;
  p.xic = o;  o += n_ell_r * xi_int * 4;
  p.betar = o; o += n_beta_r;
  p.red = o;  o += kWaves * kMaxEll;
  o = (o + 1) & ~1;
  p.etab = o; o += vkm::kExpNonposTab;
  p.total = o;
  return p;
}

// xi^r(r, mu_r) summed over the first NLR real-space multipoles (ccf_model.py:681-687).  With
// realspace_ccf_from_data the point is first mapped back to fiducial coordinates and the table abscissae are
// not rescaled (ccf_model.py:618-619, 675-679).
template <int NLR>
__device__ __forceinline__ double xi_real(const PPLds& xi, const PointScalars& ps, const TheoryArgs& a, double u,
                                          double mu_r, double r_par, double s_perp) {
  if (a.from_data) {
    const double rp = r_par * ps.inv_apar;
    const double rt = s_perp * ps.inv_aperp;
    u = sqrt(fma(rp, rp, rt * rt));
    mu_r = rp / u;
  }
  const double ux = clampd(u, xi.lo, xi.hi);
  const int ix = pp_interval(xi, ux);
  double xir = pp_eval_at(xi, 0, ix, ux);
  if (NLR > 1) {
    const double m2 = mu_r * mu_r;
    xir = fma(pp_eval_at(xi, 1, ix, ux), fma(1.5, m2, -0.5), xir);
    if (NLR > 2) xir = fma(pp_eval_at(xi, 2, ix, ux), fma(fma(35.0, m2, -30.0), m2, 3.0) * 0.125, xir);
  }
  return xir;
}

// V(u) = V1 + av V2 : the velocity profile shape, v_r(r) = -gb V(r/c) / (3 aH)
__device__ __forceinline__ double vel_shape(const PPLds& vr, const PointScalars& ps, const TheoryArgs& a, int iv,
                                            double uv) {
  double V = pp_eval_at(vr, 0, iv, uv);
  if (a.empirical) V = fma(ps.av, pp_eval_at(vr, 2, iv, uv), V);
  return V;
}

// Normalised dispersion sigma_v(r/c, mu_r)/sigma_v: the 1-D table, or the bicubic patches of the anisotropic
// template with both arguments clamped to the table box (FITPACK bispeu; a negative mu_r therefore reads mu = 0).
__device__ __forceinline__ double sv_shape(const PPLds& sv, const TheoryArgs& a, double u, double mu_r) {
  const double usv = clampd(u, sv.lo, sv.hi);
  const int i = pp_interval(sv, usv);
  if (a.sv_n_mu == 0) return pp_eval_at(sv, 0, i, usv);
  const int nm = a.sv_n_mu - 1;
  const double m = clampd(mu_r, a.sv_mu[0], a.sv_mu[nm]);
  int j;
  if (a.sv_mu_inv_h > 0.0) {
    j = min(max((int)((m - a.sv_mu[0]) * a.sv_mu_inv_h), 0), nm - 1);
  } else {
    int lo = 0, hi = nm;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (m >= a.sv_mu[mid]) lo = mid; else hi = mid;
    }
    j = lo;
  }
  const double du = usv - sv.knots[i], dm = m - a.sv_mu[j];
  const double* c = a.sv2d + ((size_t)i * nm + j) * 16;
  double acc = 0.0;
#pragma unroll
  for (int p = 3; p >= 0; --p) {
    const double cp = fma(fma(fma(c[4 * p + 3], dm, c[4 * p + 2]), dm, c[4 * p + 1]), dm, c[4 * p]);
    acc = fma(acc, du, cp);
  }
  return acc;
}

// One integrand point of the streaming model (ccf_model.py:648-657, 681-690), already multiplied by the
// Simpson weight.  NLR = number of real-space multipoles summed (1 = assume_isotropic).
template <int NLR>
__device__ __forceinline__ double streaming_integrand(const PPLds& sv, const PPLds& vr, const PPLds& xi,
                                                      const PointScalars& ps, const TheoryArgs& a,
                                                      const double* __restrict__ etab, double s_perp, double s_par,
                                                      double xk, double wk) {
  const double r_par = fma(-xk, ps.B, s_par);
  double r, inv_r;
  vkm::sqrt_rsqrt(fma(s_perp, s_perp, r_par * r_par), r, inv_r);
  const double mu_r = r_par * inv_r;
  const double u = r * ps.inv_c;

  const double SV = sv_shape(sv, a, u, mu_r);
  const double uv = clampd(u, vr.lo, vr.hi);
  const double V = vel_shape(vr, ps, a, pp_interval(vr, uv), uv);
  const double xir = xi_real<NLR>(xi, ps, a, u, mu_r, r_par, s_perp);
  const double inv_sv = vkm::recip(SV);
  const double z = fma(ps.A * V, mu_r, xk) * inv_sv;
  const double e = vkm::exp_nonpos((-0.5 * z) * z, etab);
  return wk * inv_sv * fma(e, xir, e);
}

// The other RSD mappings of the reference on the same tables:
//   dispersion     ccf_model.py:658-671   zero-mean Gaussian pdf, iterated real-space coordinate, Jacobian
//   kaiser         ccf_model.py:692-741   no velocity integral; nuisance M, Q; optional linearised Jacobian
//   euclid_special ccf_model.py:743-784   as kaiser with factors 3 and 2 and the linear form
// With q(r) = aH^-1 v_r(r)/r = -G V(r/c)/r and dq(r) = aH^-1 v_r'(r) = -gD D(r/c) the reference's expressions read
//   r_par <- (s_par - v/aH) / (1 + M q(r)),  J = a M q + b M Q mu_r^2 (dq - q).
// Returns f such that xi^s = sum_v f - 1 (for kaiser/euclid the "plane" has the single node x = 0, weight 1).
template <int RSD, int NLR>
__device__ __forceinline__ double rsd_integrand(const PPLds& sv, const PPLds& vr, const PPLds& xi, const PointScalars& ps,
                                                const TheoryArgs& a, const double* __restrict__ etab, double s_perp,
                                                double s_par, double xk, double wk) {
  if (RSD == VK_RSD_STREAMING) return streaming_integrand<NLR>(sv, vr, xi, ps, a, etab, s_perp, s_par, xk, wk);
  const double mfac = (RSD == VK_RSD_DISPERSION) ? 1.0 : ps.M;
  const double num = (RSD == VK_RSD_DISPERSION) ? fma(-xk, ps.B, s_par) : s_par;
  const double sp2 = s_perp * s_perp;
  // square roots, 1/r and the divisions by 1 + q, the Jacobian and sigma_v through the refined v_rsq / v_rcp forms
  // of vk_devmath.h (<= 2 ulp), as in the streaming branch: the fixed-point iteration alone has 6 of each
  auto q_of = [&](double r2) {     // aH^-1 v_r(r) / r at r = sqrt(r2)
    double r, inv_r;
    vkm::sqrt_rsqrt(r2, r, inv_r);
    const double uv = clampd(r * ps.inv_c, vr.lo, vr.hi);
    return -ps.G * vel_shape(vr, ps, a, pp_interval(vr, uv), uv) * inv_r;
  };
  double r_par = s_par;
  if (RSD == VK_RSD_DISPERSION || a.coord_shift) {
    r_par = num * vkm::recip(fma(mfac, q_of(fma(s_par, s_par, sp2)), 1.0));
    for (int it = 0; it < a.niter; ++it) r_par = num * vkm::recip(fma(mfac, q_of(fma(r_par, r_par, sp2)), 1.0));
  }
  double r, inv_r;
  vkm::sqrt_rsqrt(fma(r_par, r_par, sp2), r, inv_r);
  const double mu_r = r_par * inv_r;
  const double u = r * ps.inv_c;
  const double uv = clampd(u, vr.lo, vr.hi);
  const int iv = pp_interval(vr, uv);
  const double q = -ps.G * vel_shape(vr, ps, a, iv, uv) * inv_r;
  // derivative table: analytic delta - 2 Delta/3, or the numerical-gradient tables of the empirical branch
  const double Dq = a.empirical ? fma(ps.av, pp_eval_at(vr, 4, iv, uv), pp_eval_at(vr, 3, iv, uv)) : pp_eval_at(vr, 1, iv, uv);
  const double dq = -ps.gD * Dq;
  const double m2 = mu_r * mu_r;
  const double xir = xi_real<NLR>(xi, ps, a, u, mu_r, r_par, s_perp);
  if (RSD == VK_RSD_DISPERSION) {
    const double SV = sv_shape(sv, a, u, mu_r);
    const double inv_sv = vkm::recip(SV);
    const double z = xk * inv_sv;
    const double jac = vkm::recip(1.0 + q + m2 * (dq - q));
    return wk * (1.0 + xir) * jac * vkm::exp_nonpos((-0.5 * z) * z, etab) * inv_sv;
  }
  if (RSD == VK_RSD_KAISER) {
    const double J = ps.M * q + ps.M * ps.Q * m2 * (dq - q);
    if (a.kaiser_approx) return 1.0 + (ps.M * xir - J);
    return (1.0 + ps.M * xir) * vkm::recip(1.0 + J);
  }
  const double J = 3.0 * ps.M * q + 2.0 * ps.M * ps.Q * m2 * (dq - q);
  return 1.0 + (ps.M * xir - J);
}

// stage the batch-constant tables into LDS and fill the PPLds views
__device__ void stage_tables(const TheoryArgs& a, const LdsPlan& pl, double* lds, int n_ell_r_used, PPLds& sv,
                             PPLds& vr, PPLds& xi) {
  const int tid = threadIdx.x;
  for (int i = tid; i < a.n_mu; i += kBlock) {
    const double m = a.mu[i];
    lds[pl.mu + i] = m;
    lds[pl.smu + i] = sqrt(1.0 - m * m);
  }
  for (int i = tid; i < a.n_ell * a.n_mu; i += kBlock) lds[pl.w + i] = a.w_ell[i];
  for (int i = tid; i < a.n_x; i += kBlock) {
    lds[pl.x + i] = a.x[i];
    lds[pl.wx + i] = a.w_x[i];
  }
  for (int j = tid; j < vkm::kExpNonposTab; j += kBlock) lds[pl.etab + j] = vkm::exp2_frac(j);
  for (int i = tid; i <= a.sv.n_int; i += kBlock) lds[pl.svk + i] = a.sv.knots[i];
  for (int i = tid; i < a.sv.n_int * 4; i += kBlock) lds[pl.svc + i] = a.sv_n_mu ? 0.0 : a.sv.coef[i];
  for (int i = tid; i <= a.vr.n_int; i += kBlock) lds[pl.vrk + i] = a.vr.knots[i];
  for (int i = tid; i < kVrVars * a.vr.n_int * 4; i += kBlock)
    lds[pl.vrc + i] = a.vr_beta_dep ? 0.0 : a.vr.coef[i];
  for (int i = tid; i <= a.xi.n_int; i += kBlock) lds[pl.xik + i] = a.xi.knots[i];
  if (a.n_beta_r == 0) {
    for (int i = tid; i < n_ell_r_used * a.xi.n_int * 4; i += kBlock) lds[pl.xic + i] = a.xi.coef[i];
  } else {
    for (int i = tid; i < a.n_beta_r; i += kBlock) lds[pl.betar + i] = a.beta_r[i];
  }
  auto fill = [&](PPLds& t, const PPView& v, int k, int c) {
    t.knots = lds + k;
    t.coef = lds + c;
    t.n_int = v.n_int;
    t.lead = v.lead;
    t.inv_h = v.inv_h;
    t.lo = v.knots[0];
    t.hi = v.knots[v.n_int];
    t.x_u0 = v.knots[v.lead];
  };
  fill(sv, a.sv, pl.svk, pl.svc);
  fill(vr, a.vr, pl.vrk, pl.vrc);
  fill(xi, a.xi, pl.xik, pl.xic);
}

// per-point xi^r tables when the real-space input depends on the reconstruction beta:
// coef[l][i][q] = sum_p T[l][k][i][q][p] (beta - beta_k)^p   (PCHIP piece k; extrapolates with end pieces)
__device__ void build_beta_tables(const TheoryArgs& a, const LdsPlan& pl, double* lds, int n_ell_r_used,
                                  double beta) {
  const double* bg = lds + pl.betar;
  int k = 0;
  for (int i = 1; i < a.n_beta_r - 1; ++i) k = (beta >= bg[i]) ? i : k;
  const double db = beta - bg[k];
  const int per_l = a.xi.n_int * 4;
  const size_t stride_l = (size_t)(a.n_beta_r - 1) * per_l * 4;
  for (int e = threadIdx.x; e < n_ell_r_used * per_l; e += kBlock) {
    const int l = e / per_l;
    const int iq = e - l * per_l;
    const double* c = a.xi.coef + l * stride_l + ((size_t)k * per_l + iq) * 4;
    lds[pl.xic + e] = fma(fma(fma(c[3], db, c[2]), db, c[1]), db, c[0]);
  }
  if (a.vr_beta_dep) {   // V1 and Da follow xi^r_0(beta) (linear_bias with reconstruction)
    const int per_v = a.vr.n_int * 4;
    const size_t stride_v = (size_t)(a.n_beta_r - 1) * per_v * 4;
    for (int e = threadIdx.x; e < 2 * per_v; e += kBlock) {
      const int var = e / per_v;
      const int iq = e - var * per_v;
      const double* c = a.vr.coef + var * stride_v + ((size_t)k * per_v + iq) * 4;
      lds[pl.vrc + e] = fma(fma(fma(c[3], db, c[2]), db, c[1]), db, c[0]);
    }
    if (a.empirical) {   // V2 = r Delta delta, Ge1, Ge2: products of PCHIP cubics -> degree 6 in beta (ccf_model.py:451-459)
      const size_t stride_e = (size_t)(a.n_beta_r - 1) * per_v * 7;
      for (int e = threadIdx.x; e < 3 * per_v; e += kBlock) {
        const int var = e / per_v;
        const int iq = e - var * per_v;
        const double* c = a.vr_emp + var * stride_e + ((size_t)k * per_v + iq) * 7;
        double v = c[6];
#pragma unroll
        for (int p = 5; p >= 0; --p) v = fma(v, db, c[p]);
        lds[pl.vrc + 2 * per_v + e] = v;
      }
    }
  }
}

// --------------------------------------------------------------------------------------------------
// K1: theory multipoles
// --------------------------------------------------------------------------------------------------
template <int RSD, int NLR, int NL>
__global__ __launch_bounds__(kBlock) void vk_theory_kernel(TheoryArgs a) {
  extern __shared__ double lds[];
  const LdsPlan pl = make_plan(a.n_mu, a.n_x, a.n_ell, a.sv.n_int, a.vr.n_int, a.xi.n_int, NLR, a.n_beta_r);
  PPLds sv, vr, xi;
  stage_tables(a, pl, lds, NLR, sv, vr, xi);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int team = a.team;
  const int nteams = kWaves / team;
  const int my_team = wave / team;
  const int my_rank = wave - my_team * team;
  const int groups = (a.n_s + a.sbins_per_item - 1) / a.sbins_per_item;
  const long long items = a.n * groups;
  const int plane = a.n_mu * a.n_x;
  const int step = 64 * team;
  const int rounds = (a.sbins_per_item + nteams - 1) / nteams;
  const double* l_mu = lds + pl.mu;
  const double* l_smu = lds + pl.smu;
  const double* l_w = lds + pl.w;
  const double* l_x = lds + pl.x;
  const double* l_wx = lds + pl.wx;
  double* l_red = lds + pl.red;
  const double* l_etab = lds + pl.etab;

  double wsum[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    double t = 0.0;
    for (int i = lane; i < a.n_mu; i += 64) t += l_w[l * a.n_mu + i];
    wsum[l] = wave_sum(t);
  }

  for (long long item = blockIdx.x; item < items; item += gridDim.x) {
    const long long point = item / groups;
    const int g = (int)(item - point * groups);
    const double* row = param_row(a, point);
    const PointScalars ps = point_scalars(a, row);
    if (a.n_beta_r > 0) {
      __syncthreads();  // previous item's readers are done with the per-point table
      build_beta_tables(a, pl, lds, NLR, row[VK_P_BETA]);
      __syncthreads();
    }
    for (int rd = 0; rd < rounds; ++rd) {
      const int jl = rd * nteams + my_team;
      const int j = g * a.sbins_per_item + jl;
      const bool valid = (jl < a.sbins_per_item) && (j < a.n_s);
      double acc[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[l] = 0.0;
      if (valid) {
        const double sj = a.s[j];
        const double s_aperp = sj * ps.aperp;
        const double s_apar = sj * ps.apar;
        int idx = lane + 64 * my_rank;
        int i = idx / a.n_x;
        int k = idx - i * a.n_x;
        for (; idx < plane; idx += step) {
          const double f = rsd_integrand<RSD, NLR>(sv, vr, xi, ps, a, l_etab, s_aperp * l_smu[i], s_apar * l_mu[i],
                                                   l_x[k], l_wx[k]);
#pragma unroll
          for (int l = 0; l < NL; ++l) acc[l] = fma(l_w[l * a.n_mu + i], f, acc[l]);
          k += step;
          while (k >= a.n_x) { k -= a.n_x; ++i; }
        }
      }
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[l] = wave_sum(acc[l]);
      if (team == 1) {
        if (valid && lane < NL) {
          double v = acc[0] - wsum[0];
#pragma unroll
          for (int l = 1; l < NL; ++l) v = (lane == l) ? acc[l] - wsum[l] : v;
          a.out[point * (long long)(a.n_ell * a.n_s) + (long long)lane * a.n_s + j] = v + ps.poison;
        }
      } else {
        __syncthreads();
        if (lane == 0) {
#pragma unroll
          for (int l = 0; l < NL; ++l) l_red[wave * kMaxEll + l] = acc[l];
        }
        __syncthreads();
        if (valid && my_rank == 0 && lane < NL) {
          double v = 0.0;
          for (int q = 0; q < team; ++q) v += l_red[(wave + q) * kMaxEll + lane];
          double ws = wsum[0];
#pragma unroll
          for (int l = 1; l < NL; ++l) ws = (lane == l) ? wsum[l] : ws;
          a.out[point * (long long)(a.n_ell * a.n_s) + (long long)lane * a.n_s + j] = v - ws + ps.poison;
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------------------
// K1x: xi^s(mu_i, s_j), one wave per (point, mu, s) cell, lanes over the velocity nodes
// --------------------------------------------------------------------------------------------------
template <int RSD, int NLR>
__global__ __launch_bounds__(kBlock) void vk_xi_smu_kernel(TheoryArgs a) {
  extern __shared__ double lds[];
  const LdsPlan pl = make_plan(a.n_mu, a.n_x, a.n_ell, a.sv.n_int, a.vr.n_int, a.xi.n_int, NLR, a.n_beta_r);
  PPLds sv, vr, xi;
  stage_tables(a, pl, lds, NLR, sv, vr, xi);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const double* l_mu = lds + pl.mu;
  const double* l_smu = lds + pl.smu;
  const double* l_x = lds + pl.x;
  const double* l_wx = lds + pl.wx;
  const double* l_etab = lds + pl.etab;
  const int cells = a.n_mu * a.n_s;
  const int rounds = (cells + kWaves - 1) / kWaves;
  for (long long point = blockIdx.x; point < a.n; point += gridDim.x) {
    const double* row = param_row(a, point);
    const PointScalars ps = point_scalars(a, row);
    if (a.n_beta_r > 0) {
      __syncthreads();
      build_beta_tables(a, pl, lds, NLR, row[VK_P_BETA]);
      __syncthreads();
    }
    for (int rd = 0; rd < rounds; ++rd) {
      const int cell = rd * kWaves + wave;
      if (cell >= cells) break;
      const int i = cell / a.n_s;
      const int j = cell - i * a.n_s;
      const double sj = a.s[j];
      const double s_perp = sj * l_smu[i] * ps.aperp;
      const double s_par = sj * l_mu[i] * ps.apar;
      double acc = 0.0;
      for (int k = lane; k < a.n_x; k += 64)
        acc += rsd_integrand<RSD, NLR>(sv, vr, xi, ps, a, l_etab, s_perp, s_par, l_x[k], l_wx[k]);
      acc = wave_sum(acc);
      if (lane == 0) a.out[(point * a.n_mu + i) * (long long)a.n_s + j] = acc - 1.0 + ps.poison;
    }
  }
}


}  // namespace vk
