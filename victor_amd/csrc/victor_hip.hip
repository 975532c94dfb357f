// victor_hip.hip - MI355X (gfx950, CDNA4) kernels and C ABI for victor's likelihood hot path.
//
// K1  vk_theory_kernel     fused Gaussian-streaming quadrature + Legendre projection
//                          (reference: CCFModel.theory_xi streaming branch, victor/ccf_model.py:589-690,
//                           theory_multipoles :816-825, utils.multipoles_from_fn victor/utils.py:45-56)
// K1x vk_xi_smu_kernel     the un-projected xi^s(mu, s) for CCFModel.theory_xi callers
// K2  vk_like_kernel       residual . precision . residual, log det, likelihood form, NaN guard
//                          (reference: CCFFit.chi_squared victor/ccf_fit.py:349-354,
//                           get_interpolated_{covariance,precision} :195-260, log_likelihood :444-481)
//
// Design: all arithmetic is IEEE binary64 on the vector ALU; the work per evaluation is
// n_s*n_mu*n_x (= 200 000) integrand points of ~130 FP64 operations each against ~64 bytes of HBM traffic,
// so the kernels are laid out for VALU issue and LDS gather bandwidth, not for HBM:
//   * every spline of the reference is an explicit piecewise-cubic table staged ONCE per workgroup in LDS
//     (<= 14 KB of the CU's 160 KB), per-point tables (reconstruction beta) are rebuilt in LDS per point;
//   * one 64-lane wavefront owns one (parameter point, s bin): its lanes sweep the flattened (mu, v) plane
//     (5000 nodes -> 79 trips at 98.9 % lane use), accumulate W_l[mu]*w[v]*integrand for l = 0,2,4 in
//     registers and finish with one cross-lane reduction - no atomics, no second pass;
//   * per-point scalars (AP factors, rescaling integral, velocity amplitude) are wave-uniform;
//   * small batches (MCMC with one proposal per step) split a single s bin over the four waves of a
//     workgroup so that batch = 1 still spreads over 40 workgroups.

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "victor_hip.h"
#include "vk_devmath.h"

namespace {

constexpr int kBlock = 256;               // 4 wavefronts
constexpr int kWaves = kBlock / 64;
constexpr int kMaxEll = 3;
constexpr int kVrVars = 5;               // V1, Da, V2, Ge1, Ge2 (see vk_tables.vr)

// --------------------------------------------------------------------------------------------------
// device-side views
// --------------------------------------------------------------------------------------------------
struct PPView {           // a vk_pp living in global memory (device pointers)
  int n_int;
  int lead;
  double inv_h;
  const double* knots;
  const double* coef;
};

struct TheoryArgs {
  const double* params;   // [n][VK_NPAR]
  long long n;
  int n_s, n_mu, n_x, n_ell;
  const double* s;        // [n_s]
  const double* mu;       // [n_mu]
  const double* w_ell;    // [n_ell][n_mu]
  const double* x;        // [n_x]
  const double* w_x;      // [n_x]
  int n_beta_r;           // 0 = fixed xi tables
  const double* beta_r;
  PPView xi, vr, sv;
  double iaH;
  double inv_sigma8;
  int rescale_from_ap;
  int matter_vt;          // velocity-template mean model: amplitude = -3 iaH vt_amp fsigma8
  double vt_amp;
  int sv_n_mu;            // > 0: anisotropic sigma_v(r, mu) template, bicubic patches in global memory
  double sv_mu_inv_h;
  const double* sv_mu;
  const double* sv2d;
  int matter_lb;          // linear_bias matter model: amplitudes carry 1/bias (ccf_model.py:358-370,426-435)
  int vr_beta_dep;        // velocity tables are PCHIP-in-beta polynomials (rebuilt per point)
  int from_data;          // ccf_model.py:618-619,675-679
  int empirical;          // ccf_model.py:451-459
  int rsd;                // VK_RSD_*
  int niter;              // fixed-point iterations of the dispersion / Kaiser coordinate shift
  int kaiser_approx;      // ccf_model.py:730-738
  int coord_shift;        // ccf_model.py:698-707
  int sbins_per_item;     // s bins handled by one workgroup visit
  int team;               // waves cooperating on one s bin (1, 2 or 4)
  double* out;            // theory: [n][n_ell*n_s];  xi_smu: [n][n_mu][n_s]
};

struct LikeArgs {
  const double* params;
  const double* theory;   // [n][N]
  long long n;
  int N;
  int n_beta_d;
  const double* beta_d;
  const double* data;
  int n_beta_c;
  const double* beta_c;
  const double* prec;
  const double* logdet;
  const double* eig;
  int like_form;
  double nmocks, nparams;
  double* lnl;
  double* chi2;
};

// piecewise-cubic table resident in LDS
struct PPLds {
  const double* knots;
  const double* coef;
  int n_int;
  int lead;
  double inv_h;
  double lo, hi, x_u0;
};

__device__ __forceinline__ int pp_interval(const PPLds& t, double u) {
  int i;
  if (t.inv_h > 0.0) {
    const int n_uniform = t.n_int - t.lead;
    double tt = (u - t.x_u0) * t.inv_h;
    i = (int)tt;
    i = min(max(i, 0), n_uniform - 1) + t.lead;
    if (t.lead && u < t.x_u0) i = 0;
  } else {
    // general knots: largest i with knots[i] <= u
    int lo = 0, hi = t.n_int;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (u >= t.knots[mid]) lo = mid; else hi = mid;
    }
    i = lo;
  }
  return i;
}

__device__ __forceinline__ double pp_eval_at(const PPLds& t, int var, int i, double u) {
  const double dx = u - t.knots[i];
  const double* c = t.coef + ((size_t)var * t.n_int + i) * 4;
  return fma(fma(fma(c[3], dx, c[2]), dx, c[1]), dx, c[0]);
}

__device__ __forceinline__ double clampd(double u, double lo, double hi) { return fmin(fmax(u, lo), hi); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// per-point, wave-uniform quantities (ccf_model.py:589-613, 432-450, 638)
struct PointScalars {
  double aperp, apar, inv_c, A, B;
  double G;       // fsigma8/(3 sigma8_tmpl):  aH^-1 v_r(r)/r = -G V(r/c)/r
  double gD;      // fsigma8/(sigma8_tmpl c):   aH^-1 v_r'(r)  = -gD D(r/c)
  double M, Q;    // Kaiser nuisance parameters (ccf_model.py:695-696)
  double av;      // Av (divided by bias for linear_bias) when empirical_corr is on, else 0
  double inv_aperp, inv_apar;
  double poison;  // 0, or NaN when any input of the point is NaN/inf: added to every output so that a bad
                  // parameter can never be masked by a clamp (the reference propagates NaN, ccf_fit.py:477)
};

// layout of the dynamic LDS block (doubles)
struct LdsPlan {
  int mu, smu, w, x, wx, svk, svc, vrk, vrc, xik, xic, betar, red, total;
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
                                                      const PointScalars& ps, const TheoryArgs& a, double s_perp,
                                                      double s_par, double xk, double wk) {
  const double r_par = fma(-xk, ps.B, s_par);
  const double r = sqrt(fma(s_perp, s_perp, r_par * r_par));
  const double mu_r = r_par / r;
  const double u = r * ps.inv_c;

  const double SV = sv_shape(sv, a, u, mu_r);
  const double uv = clampd(u, vr.lo, vr.hi);
  const double V = vel_shape(vr, ps, a, pp_interval(vr, uv), uv);
  const double xir = xi_real<NLR>(xi, ps, a, u, mu_r, r_par, s_perp);
  const double inv_sv = 1.0 / SV;
  const double z = fma(ps.A * V, mu_r, xk) * inv_sv;
  const double e = exp(-0.5 * z * z);
  return wk * (1.0 + xir) * e * inv_sv;
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
                                                const TheoryArgs& a, double s_perp, double s_par, double xk, double wk) {
  if (RSD == VK_RSD_STREAMING) return streaming_integrand<NLR>(sv, vr, xi, ps, a, s_perp, s_par, xk, wk);
  const double mfac = (RSD == VK_RSD_DISPERSION) ? 1.0 : ps.M;
  const double num = (RSD == VK_RSD_DISPERSION) ? fma(-xk, ps.B, s_par) : s_par;
  const double sp2 = s_perp * s_perp;
  auto q_of = [&](double r) {
    const double uv = clampd(r * ps.inv_c, vr.lo, vr.hi);
    return -ps.G * vel_shape(vr, ps, a, pp_interval(vr, uv), uv) / r;
  };
  double r_par = s_par;
  if (RSD == VK_RSD_DISPERSION || a.coord_shift) {
    const double s_true = sqrt(fma(s_par, s_par, sp2));
    r_par = num / (1.0 + mfac * q_of(s_true));
    for (int it = 0; it < a.niter; ++it) {
      const double r = sqrt(fma(r_par, r_par, sp2));
      r_par = num / (1.0 + mfac * q_of(r));
    }
  }
  const double r = sqrt(fma(r_par, r_par, sp2));
  const double mu_r = r_par / r;
  const double u = r * ps.inv_c;
  const double uv = clampd(u, vr.lo, vr.hi);
  const int iv = pp_interval(vr, uv);
  const double q = -ps.G * vel_shape(vr, ps, a, iv, uv) / r;
  // derivative table: analytic delta - 2 Delta/3, or the numerical-gradient tables of the empirical branch
  const double Dq = a.empirical ? fma(ps.av, pp_eval_at(vr, 4, iv, uv), pp_eval_at(vr, 3, iv, uv)) : pp_eval_at(vr, 1, iv, uv);
  const double dq = -ps.gD * Dq;
  const double m2 = mu_r * mu_r;
  const double xir = xi_real<NLR>(xi, ps, a, u, mu_r, r_par, s_perp);
  if (RSD == VK_RSD_DISPERSION) {
    const double SV = sv_shape(sv, a, u, mu_r);
    const double inv_sv = 1.0 / SV;
    const double z = xk * inv_sv;
    const double jac = 1.0 / (1.0 + q + m2 * (dq - q));
    return wk * (1.0 + xir) * jac * exp(-0.5 * z * z) * inv_sv;
  }
  if (RSD == VK_RSD_KAISER) {
    const double J = ps.M * q + ps.M * ps.Q * m2 * (dq - q);
    if (a.kaiser_approx) return 1.0 + (ps.M * xir - J);
    return (1.0 + ps.M * xir) / (1.0 + J);
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
  }
}

__device__ __forceinline__ PointScalars point_scalars(const TheoryArgs& a, const double* row) {
  PointScalars ps;
  const double fs8 = row[VK_P_FSIGMA8];
  const double sigv = row[VK_P_SIGMAV];
  ps.aperp = row[VK_P_APERP];
  ps.apar = row[VK_P_APAR];
  const double eps = row[VK_P_EPSILON];
  double c;
  if (a.rescale_from_ap) {
    // ccf_model.py:609-611: trapz over mu = linspace(1e-10, 1, 50) of apar*sqrt(1+(1-mu^2)(eps^2-1))
    const int lane = threadIdx.x & 63;
    const double e2 = eps * eps - 1.0;
    const double h = (1.0 - 1e-10) / 49.0;
    double v = 0.0;
    if (lane < 50) {
      const double m = (lane == 49) ? 1.0 : fma((double)lane, h, 1e-10);
      v = ps.apar * sqrt(fma(1.0 - m * m, e2, 1.0));
      if (lane == 0 || lane == 49) v *= 0.5;
    }
    c = wave_sum(v) * h;
  } else {
    c = row[VK_P_ASTAR];
  }
  ps.inv_c = 1.0 / c;
  const double iaH_true = a.iaH * ps.apar;
  // growth term and powers of the bias (ccf_model.py:426-435, 358-370): v_r = -gb [V1 + av V2](r/c) / (3 aH)
  double growth = fs8 * a.inv_sigma8;
  double binv = 1.0, extra = 0.0;
  if (a.matter_lb) {
    const double bias = row[VK_P_BIAS];
    if (a.from_data) growth = row[VK_P_BETA] * bias;
    binv = 1.0 / bias;
    extra += bias;
  }
  // velocity template: v_r = growth_t V_t(r/c), growth_t = fsigma8 vt_amp / apar  ==  -gb V_t / (3 aH_true)
  if (a.matter_vt) growth = -3.0 * a.iaH * a.vt_amp * fs8;
  const double gb = growth * binv;
  ps.av = 0.0;
  if (a.empirical && !a.matter_vt) {
    ps.av = row[VK_P_AV] * binv;
    extra += ps.av;
  }
  ps.B = sigv * iaH_true;
  ps.A = gb / (3.0 * iaH_true * sigv);
  ps.G = gb / 3.0;
  ps.gD = gb * ps.inv_c;
  ps.M = row[VK_P_M];
  ps.Q = row[VK_P_Q];
  ps.inv_aperp = 1.0 / ps.aperp;
  ps.inv_apar = 1.0 / ps.apar;
  ps.poison = 0.0 * (gb + sigv + ps.aperp + ps.apar + eps + c + ps.A + extra + (a.n_beta_r > 0 ? row[VK_P_BETA] : 0.0));
  return ps;
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
    const double* row = a.params + point * VK_NPAR;
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
          const double f = rsd_integrand<RSD, NLR>(sv, vr, xi, ps, a, s_aperp * l_smu[i], s_apar * l_mu[i],
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
// K1 fast path: all three tables on uniform grids, and the velocity table shares the xi^r knots behind its
// extra leading node at 0.01 (always true for tables built by the reference's own recipe, ccf_model.py:625).
//   * coefficients are re-expressed in interval units (tau = (u - knot_i)/h in [0,1)) when they are staged,
//     so one fma + clamp + v_cvt + v_fract yields interval and local coordinate, with no knot read;
//   * V, xi_0, xi_2, xi_4 of one r interval sit in one LDS record (one index for four cubics), records are
//     padded to 4*(1+NLR)+2 doubles so that the ds_read_b128 of 16 consecutive intervals hit distinct banks;
//   * sqrt and 1/r come from one refined v_rsq_f64, 1/sigma_v from a refined v_rcp_f64, exp from a 32-entry
//     2^(j/32) table and a degree-6 polynomial (vk_devmath.h; all within 2 ulp).
// --------------------------------------------------------------------------------------------------
typedef double vk_d2 __attribute__((ext_vector_type(2)));
constexpr int kMuRec = 6;   // {mu, sqrt(1-mu^2), W_0, W_1, W_2, pad}
constexpr int kSvRec = 6;   // {c0..c3, pad, pad}

struct FastPlan {
  int murec, xrec, svrec, vxrec, lead, etab, betar, red, node, total, vx_stride;
};

__host__ __device__ inline FastPlan make_fast_plan(int n_mu, int n_x, int sv_int, int xi_int, int nlr, int n_beta_r) {
  FastPlan p;
  int o = 0;
  p.vx_stride = 4 * (1 + nlr) + 2;
  p.murec = o; o += n_mu * kMuRec;
  p.xrec = o;  o += n_x * 2;
  p.svrec = o; o += sv_int * kSvRec;
  p.vxrec = o; o += xi_int * p.vx_stride;
  p.lead = o;  o += 4;
  p.etab = o;  o += 32;
  p.betar = o; o += n_beta_r;
  o = (o + 1) & ~1;
  p.red = o;   o += kWaves * kMaxEll;
  p.node = o;  o += (n_mu * n_x + 1) / 2;   // one packed u32 per (mu, v) node
  p.total = o;
  return p;
}

struct FastConsts {
  double inv_hs, off_s, ns_eps;   // sigma_v table
  double inv_hx, off_x, nx_eps;   // xi / V table (uniform part)
  double inv_hl, off_l;           // V leading interval [0.01, r_0]
};

// v_min_f64 without the canonicalising v_max hipcc puts in front of fmin() for a bound it cannot prove quiet
// (the bound is a finite table size; the other operand comes out of an fma/max and is canonical already)
__device__ __forceinline__ double vmin_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ const double* lds_at(const double* base, int byte_off) {
  return reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
}

__device__ __forceinline__ double cubic_b128(const double* rec, double t) {
  const vk_d2 lo = *reinterpret_cast<const vk_d2*>(rec);
  const vk_d2 hi = *reinterpret_cast<const vk_d2*>(rec + 2);
  return fma(fma(fma(hi.y, t, hi.x), t, lo.y), t, lo.x);
}

template <int NLR>
__device__ __forceinline__ double fast_integrand(const double* __restrict__ svrec, const double* __restrict__ vxrec,
                                                 const double* __restrict__ leadrec, const double* __restrict__ etab,
                                                 const FastConsts& fc, const PointScalars& ps,
                                                 double s_perp, double s_par, double xk, double wk) {
  constexpr int vx_stride = 4 * (1 + NLR) + 2;
  const double r_par = fma(-xk, ps.B, s_par);
  const double r2 = fma(s_perp, s_perp, r_par * r_par);
  double r, inv_r;
  vkm::sqrt_rsqrt(r2, r, inv_r);
  const double mu_r = r_par * inv_r;
  const double u = r * ps.inv_c;

  const double ts = vmin_f64(fmax(fma(u, fc.inv_hs, fc.off_s), 0.0), fc.ns_eps);
  const double SV = cubic_b128(lds_at(svrec, __mul24((int)ts, kSvRec * 8)), __builtin_amdgcn_fract(ts));

  const double tr = fma(u, fc.inv_hx, fc.off_x);
  const double tx = vmin_f64(fmax(tr, 0.0), fc.nx_eps);
  const double tq = __builtin_amdgcn_fract(tx);
  const double* rec = lds_at(vxrec, __mul24((int)tx, vx_stride * 8));
  double V = cubic_b128(rec, tq);
  if (tr < 0.0) V = cubic_b128(leadrec, fmax(fma(u, fc.inv_hl, fc.off_l), 0.0));
  double xir = cubic_b128(rec + 4, tq);
  if (NLR > 1) {
    const double m2 = mu_r * mu_r;
    xir = fma(cubic_b128(rec + 8, tq), fma(1.5, m2, -0.5), xir);
    if (NLR > 2) xir = fma(cubic_b128(rec + 12, tq), vkm::fma3(vkm::fma3(m2, 4.375, -3.75), m2, 0.375), xir);
  }
  const double inv_sv = vkm::recip(SV);
  const double z = fma(ps.A * V, mu_r, xk) * inv_sv;
  const double e = vkm::exp_nonpos((-0.5 * z) * z, etab);
  const double t1 = wk * inv_sv;
  return fma(t1, xir, t1) * e;
}

__device__ __forceinline__ double hpow(double h, int q) {
  return q == 0 ? 1.0 : (q == 1 ? h : (q == 2 ? h * h : h * h * h));
}

template <int NLR, int NL>
__global__ __launch_bounds__(kBlock) void vk_theory_fast_kernel(TheoryArgs a) {
  extern __shared__ double lds[];
  const FastPlan pl = make_fast_plan(a.n_mu, a.n_x, a.sv.n_int, a.xi.n_int, NLR, a.n_beta_r);
  const int tid = threadIdx.x;
  const double hs = 1.0 / a.sv.inv_h, hx = 1.0 / a.xi.inv_h;
  const double hl = a.vr.knots[1] - a.vr.knots[0];
  // ---- stage batch-constant tables -------------------------------------------------------------
  for (int i = tid; i < a.n_mu; i += kBlock) {
    const double m = a.mu[i];
    double* rec = lds + pl.murec + i * kMuRec;
    rec[0] = m;
    rec[1] = sqrt(1.0 - m * m);
#pragma unroll
    for (int l = 0; l < kMaxEll; ++l) rec[2 + l] = (l < NL) ? a.w_ell[l * a.n_mu + i] : 0.0;
    rec[5] = 0.0;
  }
  for (int i = tid; i < a.n_x; i += kBlock) {
    lds[pl.xrec + 2 * i] = a.x[i];
    lds[pl.xrec + 2 * i + 1] = a.w_x[i];
  }
  for (int e = tid; e < a.sv.n_int * 4; e += kBlock)
    lds[pl.svrec + (e >> 2) * kSvRec + (e & 3)] = a.sv.coef[e] * hpow(hs, e & 3);
  for (int e = tid; e < a.xi.n_int * 4; e += kBlock)   // V lives one interval further in its own table
    lds[pl.vxrec + (e >> 2) * pl.vx_stride + (e & 3)] = a.vr.coef[4 + e] * hpow(hx, e & 3);
  if (a.n_beta_r == 0) {
    const int per_l = a.xi.n_int * 4;
    for (int e = tid; e < NLR * per_l; e += kBlock) {
      const int l = e / per_l, iq = e - l * per_l;
      lds[pl.vxrec + (iq >> 2) * pl.vx_stride + 4 * (1 + l) + (iq & 3)] = a.xi.coef[e] * hpow(hx, iq & 3);
    }
  } else {
    for (int i = tid; i < a.n_beta_r; i += kBlock) lds[pl.betar + i] = a.beta_r[i];
  }
  if (tid < 4) lds[pl.lead + tid] = a.vr.coef[tid] * hpow(hl, tid);
  if (tid < 32) lds[pl.etab + tid] = vkm::exp2_frac32(tid);
  // byte offsets of the mu record (low 16 bits) and the (x, w) record (high 16 bits) of every plane node, so the
  // hot loop needs no index arithmetic: one ds_read_b32 per trip
  unsigned* node = reinterpret_cast<unsigned*>(lds + pl.node);
  for (int idx = tid; idx < a.n_mu * a.n_x; idx += kBlock) {
    const int i = idx / a.n_x, k = idx - i * a.n_x;
    node[idx] = (unsigned)(i * kMuRec * 8) | ((unsigned)(k * 16) << 16);
  }
  FastConsts fc;
  fc.inv_hs = a.sv.inv_h;
  fc.off_s = -a.sv.knots[0] * a.sv.inv_h;
  fc.ns_eps = __builtin_canonicalize((double)a.sv.n_int * (1.0 - 0x1p-52));
  fc.inv_hx = a.xi.inv_h;
  fc.off_x = -a.xi.knots[0] * a.xi.inv_h;
  fc.nx_eps = __builtin_canonicalize((double)a.xi.n_int * (1.0 - 0x1p-52));
  fc.inv_hl = 1.0 / hl;
  fc.off_l = -a.vr.knots[0] * fc.inv_hl;
  __syncthreads();

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int team = a.team;
  const int nteams = kWaves / team;
  const int my_team = wave / team;
  const int my_rank = wave - my_team * team;
  const int groups = (a.n_s + a.sbins_per_item - 1) / a.sbins_per_item;
  const long long items = a.n * groups;
  const int plane = a.n_mu * a.n_x;
  const int step = 64 * team;
  const int rounds = (a.sbins_per_item + nteams - 1) / nteams;
  const double* murec = lds + pl.murec;
  const double* xrec = lds + pl.xrec;
  const double* svrec = lds + pl.svrec;
  const double* vxrec = lds + pl.vxrec;
  const double* leadrec = lds + pl.lead;
  const double* etab = lds + pl.etab;
  double* l_red = lds + pl.red;

  double wsum[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    double t = 0.0;
    for (int i = lane; i < a.n_mu; i += 64) t += murec[i * kMuRec + 2 + l];
    wsum[l] = wave_sum(t);
  }

  for (long long item = blockIdx.x; item < items; item += gridDim.x) {
    const long long point = item / groups;
    const int g = (int)(item - point * groups);
    const double* row = a.params + point * VK_NPAR;
    const PointScalars ps = point_scalars(a, row);
    if (a.n_beta_r > 0) {
      __syncthreads();
      const double* bg = lds + pl.betar;
      const double beta = row[VK_P_BETA];
      int kb = 0;
      for (int i = 1; i < a.n_beta_r - 1; ++i) kb = (beta >= bg[i]) ? i : kb;
      const double db = beta - bg[kb];
      const int per_l = a.xi.n_int * 4;
      const size_t stride_l = (size_t)(a.n_beta_r - 1) * per_l * 4;
      for (int e = tid; e < NLR * per_l; e += kBlock) {
        const int l = e / per_l, iq = e - l * per_l;
        const double* c = a.xi.coef + l * stride_l + ((size_t)kb * per_l + iq) * 4;
        lds[pl.vxrec + (iq >> 2) * pl.vx_stride + 4 * (1 + l) + (iq & 3)] =
            fma(fma(fma(c[3], db, c[2]), db, c[1]), db, c[0]) * hpow(hx, iq & 3);
      }
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
        const char* mu_bytes = reinterpret_cast<const char*>(murec);
        const char* x_bytes = reinterpret_cast<const char*>(xrec);
        for (int idx = lane + 64 * my_rank; idx < plane; idx += step) {
          const unsigned pk = node[idx];
          const double* mr = reinterpret_cast<const double*>(mu_bytes + (pk & 0xffffu));
          const vk_d2 m01 = *reinterpret_cast<const vk_d2*>(mr);
          const vk_d2 xw = *reinterpret_cast<const vk_d2*>(x_bytes + (pk >> 16));
          const double f = fast_integrand<NLR>(svrec, vxrec, leadrec, etab, fc, ps, s_aperp * m01.y,
                                               s_apar * m01.x, xw.x, xw.y);
          const vk_d2 w01 = *reinterpret_cast<const vk_d2*>(mr + 2);
          acc[0] = fma(w01.x, f, acc[0]);
          if (NL > 1) acc[1] = fma(w01.y, f, acc[1]);
          if (NL > 2) acc[2] = fma(mr[4], f, acc[2]);
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
// K1 "lanes over the batch" variant (the mapping BASELINE.json's north star sketches): one wavefront owns one s bin
// of 64 consecutive parameter points, lane = point.  Everything that depends only on the (mu, v) node - mu_i,
// sqrt(1-mu_i^2), x_k, w_k, W_l[i], the loop counters - is wave-uniform and lives in SGPRs / scalar loads, the
// per-point factors live in VGPRs, no cross-lane reduction is needed and all 64 lanes are busy on every trip.
// Per integrand point this saves the node-table read, two multiplies (s_perp, s_par are formed once per mu row),
// two of the three projection FMAs (the v sum is closed per mu row first) and five LDS reads.
// Needs batch-constant tables (no reconstruction beta) and a batch large enough to fill the chip with
// n_s * n/64 wavefronts; the point-major kernel above serves every other case.
// --------------------------------------------------------------------------------------------------
struct LanesPlan {
  int smu, xw, svrec, vxrec, lead, etab, total, vx_stride;
};

__host__ __device__ inline LanesPlan make_lanes_plan(int n_mu, int n_x, int sv_int, int xi_int, int nlr) {
  LanesPlan p;
  int o = 0;
  p.vx_stride = 4 * (1 + nlr) + 2;
  p.smu = o;   o += 2 * n_mu;           // {mu_i, sqrt(1 - mu_i^2)}
  p.xw = o;    o += 2 * n_x;            // {x_k, w_k}: read with a wave-uniform address (LDS broadcast)
  p.svrec = o; o += sv_int * kSvRec;
  p.vxrec = o; o += xi_int * p.vx_stride;
  p.lead = o;  o += 4;
  p.etab = o;  o += 32;
  p.total = o;
  return p;
}

// per-lane version of point_scalars (each lane integrates its own AP rescaling factor, ccf_model.py:609-611)
__device__ __forceinline__ PointScalars point_scalars_lane(const TheoryArgs& a, const double* row) {
  PointScalars ps;
  const double fs8 = row[VK_P_FSIGMA8];
  const double sigv = row[VK_P_SIGMAV];
  ps.aperp = row[VK_P_APERP];
  ps.apar = row[VK_P_APAR];
  const double eps = row[VK_P_EPSILON];
  double c;
  if (a.rescale_from_ap) {
    const double e2 = eps * eps - 1.0;
    const double h = (1.0 - 1e-10) / 49.0;
    double acc = 0.0;
    for (int m = 0; m < 50; ++m) {
      const double mm = (m == 49) ? 1.0 : fma((double)m, h, 1e-10);
      const double v = sqrt(fma(1.0 - mm * mm, e2, 1.0));
      acc += (m == 0 || m == 49) ? 0.5 * v : v;
    }
    c = ps.apar * acc * h;
  } else {
    c = row[VK_P_ASTAR];
  }
  ps.inv_c = 1.0 / c;
  const double iaH_true = a.iaH * ps.apar;
  double growth = fs8 * a.inv_sigma8;
  double binv = 1.0, extra = 0.0;
  if (a.matter_lb) {
    const double bias = row[VK_P_BIAS];
    binv = 1.0 / bias;
    extra = bias;
  }
  const double gb = growth * binv;
  ps.av = 0.0;
  ps.B = sigv * iaH_true;
  ps.A = gb / (3.0 * iaH_true * sigv);
  ps.G = gb / 3.0;
  ps.gD = gb * ps.inv_c;
  ps.M = row[VK_P_M];
  ps.Q = row[VK_P_Q];
  ps.inv_aperp = 1.0 / ps.aperp;
  ps.inv_apar = 1.0 / ps.apar;
  ps.poison = 0.0 * (gb + sigv + ps.aperp + ps.apar + eps + c + ps.A + extra);
  return ps;
}

template <int NLR, int NL>
__global__ __launch_bounds__(kBlock) void vk_theory_lanes_kernel(TheoryArgs a) {
  extern __shared__ double lds[];
  constexpr int vx_stride = 4 * (1 + NLR) + 2;
  const LanesPlan pl = make_lanes_plan(a.n_mu, a.n_x, a.sv.n_int, a.xi.n_int, NLR);
  const int tid = threadIdx.x;
  const double hs = 1.0 / a.sv.inv_h, hx = 1.0 / a.xi.inv_h;
  const double hl = a.vr.knots[1] - a.vr.knots[0];
  for (int i = tid; i < a.n_mu; i += kBlock) {
    const double m = a.mu[i];
    lds[pl.smu + 2 * i] = m;
    lds[pl.smu + 2 * i + 1] = sqrt(1.0 - m * m);
  }
  for (int k = tid; k < a.n_x; k += kBlock) {
    lds[pl.xw + 2 * k] = a.x[k];
    lds[pl.xw + 2 * k + 1] = a.w_x[k];
  }
  for (int e = tid; e < a.sv.n_int * 4; e += kBlock)
    lds[pl.svrec + (e >> 2) * kSvRec + (e & 3)] = a.sv.coef[e] * hpow(hs, e & 3);
  for (int e = tid; e < a.xi.n_int * 4; e += kBlock)
    lds[pl.vxrec + (e >> 2) * vx_stride + (e & 3)] = a.vr.coef[4 + e] * hpow(hx, e & 3);
  {
    const int per_l = a.xi.n_int * 4;
    for (int e = tid; e < NLR * per_l; e += kBlock) {
      const int l = e / per_l, iq = e - l * per_l;
      lds[pl.vxrec + (iq >> 2) * vx_stride + 4 * (1 + l) + (iq & 3)] = a.xi.coef[e] * hpow(hx, iq & 3);
    }
  }
  if (tid < 4) lds[pl.lead + tid] = a.vr.coef[tid] * hpow(hl, tid);
  if (tid < 32) lds[pl.etab + tid] = vkm::exp2_frac32(tid);
  FastConsts fc;
  fc.inv_hs = a.sv.inv_h;
  fc.off_s = -a.sv.knots[0] * a.sv.inv_h;
  fc.ns_eps = (double)a.sv.n_int * (1.0 - 0x1p-52);
  fc.inv_hx = a.xi.inv_h;
  fc.off_x = -a.xi.knots[0] * a.xi.inv_h;
  fc.nx_eps = (double)a.xi.n_int * (1.0 - 0x1p-52);
  fc.inv_hl = 1.0 / hl;
  fc.off_l = -a.vr.knots[0] * fc.inv_hl;
  __syncthreads();

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const double* l_smu = lds + pl.smu;
  const double* l_xw = lds + pl.xw;
  const double* svrec = lds + pl.svrec;
  const double* vxrec = lds + pl.vxrec;
  const double* leadrec = lds + pl.lead;
  const double* etab = lds + pl.etab;
  const long long chunks = (a.n + 63) >> 6;
  const long long items = chunks * a.n_s;
  for (long long item = (long long)blockIdx.x * kWaves + wave; item < items; item += (long long)gridDim.x * kWaves) {
    const long long chunk = item / a.n_s;
    const int j = (int)(item - chunk * a.n_s);
    long long point = chunk * 64 + lane;
    const bool valid = point < a.n;
    if (!valid) point = a.n - 1;
    const PointScalars ps = point_scalars_lane(a, a.params + point * VK_NPAR);
    const double sj = a.s[j];
    const double sa = sj * ps.aperp, sp = sj * ps.apar;
    const double AV = ps.A;
    double acc[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) acc[l] = 0.0;
    for (int i = 0; i < a.n_mu; ++i) {
      const vk_d2 mm = *reinterpret_cast<const vk_d2*>(l_smu + 2 * i);
      const double s_perp = sa * mm.y;
      const double sperp2 = s_perp * s_perp;
      const double s_par = sp * mm.x;
      double g = 0.0;
      for (int k = 0; k < a.n_x; ++k) {
        const vk_d2 xw = *reinterpret_cast<const vk_d2*>(l_xw + 2 * k);
        const double xk = xw.x;
        const double r_par = fma(-xk, ps.B, s_par);
        const double r2 = fma(r_par, r_par, sperp2);
        double r, inv_r;
        vkm::sqrt_rsqrt(r2, r, inv_r);
        const double mu_r = r_par * inv_r;
        const double u = r * ps.inv_c;
        const double ts = vmin_f64(fmax(fma(u, fc.inv_hs, fc.off_s), 0.0), fc.ns_eps);
        const double SV = cubic_b128(lds_at(svrec, __mul24((int)ts, kSvRec * 8)), __builtin_amdgcn_fract(ts));
        const double tr = fma(u, fc.inv_hx, fc.off_x);
        const double tx = vmin_f64(fmax(tr, 0.0), fc.nx_eps);
        const double tq = __builtin_amdgcn_fract(tx);
        const double* rec = lds_at(vxrec, __mul24((int)tx, vx_stride * 8));
        double V = cubic_b128(rec, tq);
        if (tr < 0.0) V = cubic_b128(leadrec, fmax(fma(u, fc.inv_hl, fc.off_l), 0.0));
        double xir = cubic_b128(rec + 4, tq);
        if (NLR > 1) {
          const double m2 = mu_r * mu_r;
          xir = fma(cubic_b128(rec + 8, tq), fma(1.5, m2, -0.5), xir);
          if (NLR > 2) xir = fma(cubic_b128(rec + 12, tq), vkm::fma3(vkm::fma3(m2, 4.375, -3.75), m2, 0.375), xir);
        }
        const double inv_sv = vkm::recip(SV);
        const double z = fma(AV * V, mu_r, xk) * inv_sv;
        const double e = vkm::exp_nonpos((-0.5 * z) * z, etab);
        g = fma(xw.y * inv_sv, fma(e, xir, e), g);
      }
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[l] = fma(a.w_ell[l * a.n_mu + i], g, acc[l]);
    }
    if (valid) {
      double* o = a.out + point * (long long)(a.n_ell * a.n_s) + j;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        double ws = 0.0;
        for (int i = 0; i < a.n_mu; ++i) ws += a.w_ell[l * a.n_mu + i];
        o[(long long)l * a.n_s] = acc[l] - ws + ps.poison;
      }
    }
  }
}

// --------------------------------------------------------------------------------------------------
// K1 "cells" variant: the lanes kernel's inner loop for per-point tables (reconstruction beta, BOSS).
// One workgroup owns one parameter point (its xi^r records are rebuilt in LDS as in the point-major kernel); each
// of its four waves owns the s bins j = wave, wave+4, ... and spreads the (s bin, mu) cells of those bins over its
// lanes, with the 50 velocity nodes as the inner, wave-uniform loop.  Like the lanes kernel this forms s_perp and
// s_par once per cell, reads x_k, w_k as LDS broadcasts and closes the v sum before the projection, so the
// integrand costs the same ~80 instructions; the projection sum over mu is a two-segment wave reduction per trip
// (a wave's 64 cells straddle at most two s bins when n_mu >= 64), accumulated by lane 0 in wave-private LDS.
// --------------------------------------------------------------------------------------------------
struct CellsPlan {
  int mu, w, xw, s, svrec, vxrec, lead, etab, betar, acc, total, vx_stride;
};

__host__ __device__ inline CellsPlan make_cells_plan(int n_mu, int n_x, int n_s, int n_ell, int sv_int, int xi_int,
                                                    int nlr, int n_beta_r) {
  CellsPlan p;
  int o = 0;
  p.vx_stride = 4 * (1 + nlr) + 2;
  p.mu = o;    o += 2 * n_mu;                      // {mu_i, sqrt(1 - mu_i^2)}
  p.w = o;     o += kMaxEll * n_mu;                // W_l[i]
  p.xw = o;    o += 2 * n_x;                       // {x_k, w_k}
  p.s = o;     o += (n_s + 1) & ~1;
  p.svrec = o; o += sv_int * kSvRec;
  p.vxrec = o; o += xi_int * p.vx_stride;
  p.lead = o;  o += 4;
  p.etab = o;  o += 32;
  p.betar = o; o += (n_beta_r + 1) & ~1;
  p.acc = o;   o += kMaxEll * ((n_s + kWaves - 1) / kWaves) * kWaves;   // [l][slot][wave]
  p.total = o;
  return p;
}

// (130 VGPRs -> 3 waves per SIMD; forcing 4 with __launch_bounds__(256, 4) spills and measured 1.5 % slower)
template <int NLR, int NL>
__global__ __launch_bounds__(kBlock) void vk_theory_cells_kernel(TheoryArgs a) {
  extern __shared__ double lds[];
  constexpr int vx_stride = 4 * (1 + NLR) + 2;
  const CellsPlan pl = make_cells_plan(a.n_mu, a.n_x, a.n_s, a.n_ell, a.sv.n_int, a.xi.n_int, NLR, a.n_beta_r);
  const int tid = threadIdx.x;
  const double hs = 1.0 / a.sv.inv_h, hx = 1.0 / a.xi.inv_h;
  const double hl = a.vr.knots[1] - a.vr.knots[0];
  for (int i = tid; i < a.n_mu; i += kBlock) {
    const double m = a.mu[i];
    lds[pl.mu + 2 * i] = m;
    lds[pl.mu + 2 * i + 1] = sqrt(1.0 - m * m);
#pragma unroll
    for (int l = 0; l < kMaxEll; ++l) lds[pl.w + l * a.n_mu + i] = (l < NL) ? a.w_ell[l * a.n_mu + i] : 0.0;
  }
  for (int k = tid; k < a.n_x; k += kBlock) {
    lds[pl.xw + 2 * k] = a.x[k];
    lds[pl.xw + 2 * k + 1] = a.w_x[k];
  }
  for (int j = tid; j < a.n_s; j += kBlock) lds[pl.s + j] = a.s[j];
  for (int e = tid; e < a.sv.n_int * 4; e += kBlock)
    lds[pl.svrec + (e >> 2) * kSvRec + (e & 3)] = a.sv.coef[e] * hpow(hs, e & 3);
  for (int e = tid; e < a.xi.n_int * 4; e += kBlock)
    lds[pl.vxrec + (e >> 2) * vx_stride + (e & 3)] = a.vr.coef[4 + e] * hpow(hx, e & 3);
  if (a.n_beta_r == 0) {
    const int per_l = a.xi.n_int * 4;
    for (int e = tid; e < NLR * per_l; e += kBlock) {
      const int l = e / per_l, iq = e - l * per_l;
      lds[pl.vxrec + (iq >> 2) * vx_stride + 4 * (1 + l) + (iq & 3)] = a.xi.coef[e] * hpow(hx, iq & 3);
    }
  } else {
    for (int i = tid; i < a.n_beta_r; i += kBlock) lds[pl.betar + i] = a.beta_r[i];
  }
  if (tid < 4) lds[pl.lead + tid] = a.vr.coef[tid] * hpow(hl, tid);
  if (tid < 32) lds[pl.etab + tid] = vkm::exp2_frac32(tid);
  FastConsts fc;
  fc.inv_hs = a.sv.inv_h;
  fc.off_s = -a.sv.knots[0] * a.sv.inv_h;
  fc.ns_eps = (double)a.sv.n_int * (1.0 - 0x1p-52);
  fc.inv_hx = a.xi.inv_h;
  fc.off_x = -a.xi.knots[0] * a.xi.inv_h;
  fc.nx_eps = (double)a.xi.n_int * (1.0 - 0x1p-52);
  fc.inv_hl = 1.0 / hl;
  fc.off_l = -a.vr.knots[0] * fc.inv_hl;
  __syncthreads();

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const double* l_mu = lds + pl.mu;
  const double* l_w = lds + pl.w;
  const double* l_xw = lds + pl.xw;
  const double* l_s = lds + pl.s;
  const double* svrec = lds + pl.svrec;
  const double* vxrec = lds + pl.vxrec;
  const double* leadrec = lds + pl.lead;
  const double* etab = lds + pl.etab;
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
    if (a.n_beta_r > 0) {
      __syncthreads();
      const double* bg = lds + pl.betar;
      const double beta = row[VK_P_BETA];
      int kb = 0;
      for (int i = 1; i < a.n_beta_r - 1; ++i) kb = (beta >= bg[i]) ? i : kb;
      const double db = beta - bg[kb];
      const int per_l = a.xi.n_int * 4;
      const size_t stride_l = (size_t)(a.n_beta_r - 1) * per_l * 4;
      for (int e = tid; e < NLR * per_l; e += kBlock) {
        const int l = e / per_l, iq = e - l * per_l;
        const double* c = a.xi.coef + l * stride_l + ((size_t)kb * per_l + iq) * 4;
        lds[pl.vxrec + (iq >> 2) * vx_stride + 4 * (1 + l) + (iq & 3)] =
            fma(fma(fma(c[3], db, c[2]), db, c[1]), db, c[0]) * hpow(hx, iq & 3);
      }
      __syncthreads();
    }
    for (int e = lane; e < kMaxEll * slots; e += 64) l_acc[e * kWaves + wave] = 0.0;
    const double AV = ps.A;
    for (int base = 0; base < cells; base += 64) {
      const int e = base + lane;
      const bool live = e < cells;
      const int ec = live ? e : cells - 1;
      const int jj = ec / a.n_mu;
      const int i = ec - jj * a.n_mu;
      const double sj = l_s[wave + kWaves * jj];
      const vk_d2 mm = *reinterpret_cast<const vk_d2*>(l_mu + 2 * i);
      const double s_perp = sj * ps.aperp * mm.y;
      const double sperp2 = s_perp * s_perp;
      const double s_par = sj * ps.apar * mm.x;
      double g = 0.0;
      for (int k = 0; k < a.n_x; ++k) {
        const vk_d2 xw = *reinterpret_cast<const vk_d2*>(l_xw + 2 * k);
        const double xk = xw.x;
        const double r_par = fma(-xk, ps.B, s_par);
        const double r2 = fma(r_par, r_par, sperp2);
        double r, inv_r;
        vkm::sqrt_rsqrt(r2, r, inv_r);
        const double mu_r = r_par * inv_r;
        const double u = r * ps.inv_c;
        const double ts = vmin_f64(fmax(fma(u, fc.inv_hs, fc.off_s), 0.0), fc.ns_eps);
        const double SV = cubic_b128(lds_at(svrec, __mul24((int)ts, kSvRec * 8)), __builtin_amdgcn_fract(ts));
        const double tr = fma(u, fc.inv_hx, fc.off_x);
        const double tx = vmin_f64(fmax(tr, 0.0), fc.nx_eps);
        const double tq = __builtin_amdgcn_fract(tx);
        const double* rec = lds_at(vxrec, __mul24((int)tx, vx_stride * 8));
        double V = cubic_b128(rec, tq);
        if (tr < 0.0) V = cubic_b128(leadrec, fmax(fma(u, fc.inv_hl, fc.off_l), 0.0));
        double xir = cubic_b128(rec + 4, tq);
        if (NLR > 1) {
          const double m2 = mu_r * mu_r;
          xir = fma(cubic_b128(rec + 8, tq), fma(1.5, m2, -0.5), xir);
          if (NLR > 2) xir = fma(cubic_b128(rec + 12, tq), vkm::fma3(vkm::fma3(m2, 4.375, -3.75), m2, 0.375), xir);
        }
        const double inv_sv = vkm::recip(SV);
        const double z = fma(AV * V, mu_r, xk) * inv_sv;
        const double ex = vkm::exp_nonpos((-0.5 * z) * z, etab);
        g = fma(xw.y * inv_sv, fma(ex, xir, ex), g);
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
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < NL * my_bins; e += 64) {
      const int l = e / my_bins, jj = e - l * my_bins;
      double ws = wsum[0];
#pragma unroll
      for (int q = 1; q < NL; ++q) ws = (l == q) ? wsum[q] : ws;
      a.out[point * (long long)(a.n_ell * a.n_s) + (long long)l * a.n_s + wave + kWaves * jj] =
          l_acc[(l * slots + jj) * kWaves + wave] - ws + ps.poison;
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
  const int cells = a.n_mu * a.n_s;
  const int rounds = (cells + kWaves - 1) / kWaves;
  for (long long point = blockIdx.x; point < a.n; point += gridDim.x) {
    const double* row = a.params + point * VK_NPAR;
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
        acc += rsd_integrand<RSD, NLR>(sv, vr, xi, ps, a, s_perp, s_par, l_x[k], l_wx[k]);
      acc = wave_sum(acc);
      if (lane == 0) a.out[(point * a.n_mu + i) * (long long)a.n_s + j] = acc - 1.0 + ps.poison;
    }
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
      P0 = a.prec + (size_t)lo * a.N * a.N;
      P1 = a.prec + (size_t)last * a.N * a.N;
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
    // -1/2 log det of the blended covariance, ccf_fit.py:445-451
    double factor = 0.0;
    bool singular = false;
    if (a.n_beta_c > 0) {
      double ld = 0.0;
      int bad = 0;
      if (t != 0.0) {
        const double* ev = a.eig + (size_t)lo * a.N;
        for (int e = lane; e < a.N; e += 64) {
          const double fct = fma(t, ev[e], omt);
          bad |= !(fct > 0.0);
          ld += log(fct);
        }
        ld = wave_sum(ld);
      }
      singular = __any(bad) || !(fabs(a.logdet[lo]) < inf);
      factor = -0.5 * (a.logdet[lo] + ld);
    }
    double lnl;
    const double nm = a.nmocks;
    if (a.like_form == VK_LIKE_SELLENTIN) {
      lnl = -nm * log(1.0 + chisq / (nm - 1.0)) / 2.0 + factor;
    } else if (a.like_form == VK_LIKE_HARTLAP) {
      lnl = -0.5 * chisq * ((nm - a.N - 2.0) / (nm - 1.0)) + factor;
    } else if (a.like_form == VK_LIKE_PERCIVAL) {
      const double nd = (double)a.N;
      const double B = (nm - nd - 2.0) / ((nm - nd - 1.0) * (nm - nd - 4.0));
      const double m = a.nparams + 2.0 + (nm - 1.0 + B * (nd - a.nparams)) / (1.0 + B * (nd - a.nparams));
      lnl = -m * log(1.0 + chisq / (nm - 1.0)) / 2.0 + factor;
    } else {
      lnl = -0.5 * chisq + factor;
    }
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

}  // namespace

// ==================================================================================================
// host side
// ==================================================================================================
struct vk_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  std::string err;
  int n_cu = 256;
  // host copy of sizes
  int n_s = 0, n_mu = 0, n_x = 0, n_ell = 0, n_ell_r = 0, n_beta_r = 0, n_beta_d = 0, n_beta_c = 0, N = 0;
  double iaH = 0, template_sigma8 = 0;
  double* d_tables = nullptr;  // one allocation holding every table
  // device pointers into d_tables
  const double *d_x1 = nullptr, *d_w1 = nullptr;  // single velocity node for the Kaiser-type models
  const double *d_s = nullptr, *d_mu = nullptr, *d_w = nullptr, *d_x = nullptr, *d_wx = nullptr, *d_beta_r = nullptr,
               *d_beta_d = nullptr, *d_data = nullptr, *d_beta_c = nullptr, *d_prec = nullptr, *d_logdet = nullptr,
               *d_eig = nullptr;
  PPView xi{}, vr{}, sv{};
  bool fast_ok = false;      // tables qualify for vk_theory_fast_kernel
  int matter_lb = 0, vr_beta_dep = 0, matter_vt = 0, sv_n_mu = 0;
  double vt_amp = 0, sv_mu_inv_h = 0;
  const double *d_sv_mu = nullptr, *d_sv2d = nullptr;
  const char* last_kernel = "none";  // theory kernel variant of the most recent launch
  // scratch for the host-buffer entry points
  double* d_scratch = nullptr;
  size_t scratch_bytes = 0;
  // timing
  bool timing = false;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  double theory_ms = 0, like_ms = 0;
  long long launches = 0;
  bool pending = false;
  // RCCL (loaded lazily)
  void* rccl_lib = nullptr;
  void* comm = nullptr;
};

namespace {

thread_local std::string g_create_err;

int fail(vk_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define VK_HIP(ctx, call)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) return fail((ctx), VK_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

struct Uploader {
  std::vector<double> host;
  size_t add(const double* p, size_t n) {
    size_t off = host.size();
    off = (off + 1) & ~size_t(1);  // 16-byte alignment for b128 reads
    host.resize(off + n);
    if (n) memcpy(host.data() + off, p, n * sizeof(double));
    return off;
  }
};

int ensure_scratch(vk_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return VK_OK;
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  ctx->d_scratch = nullptr;
  ctx->scratch_bytes = 0;
  VK_HIP(ctx, hipMalloc((void**)&ctx->d_scratch, bytes));
  ctx->scratch_bytes = bytes;
  return VK_OK;
}

int check_opts(vk_ctx* ctx, const vk_eval_opts* o) {
  if (!o) return fail(ctx, VK_E_ARG, "opts is NULL");
  if (o->rsd_model < VK_RSD_STREAMING || o->rsd_model > VK_RSD_EUCLID)
    return fail(ctx, VK_E_ARG, "unknown rsd_model %d", o->rsd_model);
  if (o->niter < 0 || o->niter > 64) return fail(ctx, VK_E_ARG, "niter must be in 0..64");
  if (o->like_form < VK_LIKE_GAUSSIAN || o->like_form > VK_LIKE_PERCIVAL)
    return fail(ctx, VK_E_ARG, "unknown likelihood form %d", o->like_form);
  return VK_OK;
}

void choose_split(const vk_ctx* ctx, long long n, int n_s, int* spi, int* team) {
  // enough workgroups to fill 256 CUs several times over, otherwise split finer
  const long long want = 4LL * ctx->n_cu;
  if (n >= want) { *spi = n_s; *team = 1; return; }
  if (n * ((n_s + 3) / 4) >= want) { *spi = 4; *team = 1; return; }
  if (n * n_s >= want) { *spi = 1; *team = 2; return; }
  *spi = 1; *team = 4;
}

template <int RSD, int NLR>
int launch_generic_nl(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  switch (a.n_ell) {
    case 1: hipLaunchKernelGGL((vk_theory_kernel<RSD, NLR, 1>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((vk_theory_kernel<RSD, NLR, 2>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL((vk_theory_kernel<RSD, NLR, 3>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    default: return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
  }
  VK_HIP(ctx, hipGetLastError());
  return VK_OK;
}

template <int RSD>
int launch_generic(vk_ctx* ctx, const TheoryArgs& a, int nlr, int grid, size_t lds) {
  switch (nlr) {
    case 1: return launch_generic_nl<RSD, 1>(ctx, a, grid, lds);
    case 2: return launch_generic_nl<RSD, 2>(ctx, a, grid, lds);
    case 3: return launch_generic_nl<RSD, 3>(ctx, a, grid, lds);
  }
  return fail(ctx, VK_E_ARG, "bad number of real-space multipoles %d", nlr);
}

template <int NLR>
int launch_fast_nl(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  switch (a.n_ell) {
    case 1: hipLaunchKernelGGL((vk_theory_fast_kernel<NLR, 1>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((vk_theory_fast_kernel<NLR, 2>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL((vk_theory_fast_kernel<NLR, 3>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    default: return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
  }
  VK_HIP(ctx, hipGetLastError());
  return VK_OK;
}

template <int NLR>
int launch_lanes_nl(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  switch (a.n_ell) {
    case 1: hipLaunchKernelGGL((vk_theory_lanes_kernel<NLR, 1>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((vk_theory_lanes_kernel<NLR, 2>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL((vk_theory_lanes_kernel<NLR, 3>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    default: return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
  }
  VK_HIP(ctx, hipGetLastError());
  return VK_OK;
}

template <int NLR>
int launch_cells_nl(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  switch (a.n_ell) {
    case 1: hipLaunchKernelGGL((vk_theory_cells_kernel<NLR, 1>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((vk_theory_cells_kernel<NLR, 2>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL((vk_theory_cells_kernel<NLR, 3>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    default: return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
  }
  VK_HIP(ctx, hipGetLastError());
  return VK_OK;
}

template <int RSD>
int launch_xi_smu(vk_ctx* ctx, const TheoryArgs& a, int nlr, int grid, size_t lds) {
  switch (nlr) {
    case 1: hipLaunchKernelGGL((vk_xi_smu_kernel<RSD, 1>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    case 2: hipLaunchKernelGGL((vk_xi_smu_kernel<RSD, 2>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    case 3: hipLaunchKernelGGL((vk_xi_smu_kernel<RSD, 3>), dim3(grid), dim3(kBlock), lds, ctx->stream, a); break;
    default: return fail(ctx, VK_E_ARG, "bad number of real-space multipoles %d", nlr);
  }
  VK_HIP(ctx, hipGetLastError());
  return VK_OK;
}

// fills the grid-independent part of TheoryArgs
int theory_args(vk_ctx* ctx, const vk_eval_opts* o, TheoryArgs* a, int* nlr) {
  a->n_beta_r = ctx->n_beta_r;
  a->beta_r = ctx->d_beta_r;
  a->xi = ctx->xi;
  a->vr = ctx->vr;
  a->sv = ctx->sv;
  a->iaH = ctx->iaH;
  a->inv_sigma8 = 1.0 / ctx->template_sigma8;
  a->rescale_from_ap = o->rescale_from_ap;
  a->matter_lb = ctx->matter_lb;
  a->matter_vt = ctx->matter_vt;
  a->vt_amp = ctx->vt_amp;
  a->sv_n_mu = ctx->sv_n_mu;
  a->sv_mu_inv_h = ctx->sv_mu_inv_h;
  a->sv_mu = ctx->d_sv_mu;
  a->sv2d = ctx->d_sv2d;
  a->vr_beta_dep = ctx->vr_beta_dep;
  a->from_data = o->from_data ? 1 : 0;
  a->empirical = (o->empirical_corr && !ctx->matter_vt) ? 1 : 0;   // the template-mean branch ignores Av (ccf_model.py:483-490)
  if (a->empirical && a->vr_beta_dep)
    return fail(ctx, VK_E_ARG, "empirical_corr with a beta-dependent linear_bias velocity profile is not implemented");
  a->rsd = o->rsd_model;
  a->niter = o->niter;
  a->kaiser_approx = o->kaiser_approx;
  a->coord_shift = o->kaiser_coord_shift;
  if (o->rsd_model == VK_RSD_KAISER || o->rsd_model == VK_RSD_EUCLID) {
    a->n_x = 1;                 // no velocity integral: a single node x = 0 with unit weight
    a->x = ctx->d_x1;
    a->w_x = ctx->d_w1;
  } else {
    a->n_x = ctx->n_x;
    a->x = ctx->d_x;
    a->w_x = ctx->d_wx;
  }
  *nlr = o->assume_isotropic ? 1 : ctx->n_ell_r;
  return VK_OK;
}

int launch_theory(vk_ctx* ctx, TheoryArgs a, int nlr) {
  if (a.n <= 0) return VK_OK;
  choose_split(ctx, a.n, a.n_s, &a.sbins_per_item, &a.team);
  // the fast kernel (streaming only) packs LDS byte offsets of the mu and (x, w) records into 16 bits each
  const bool fast = a.rsd == VK_RSD_STREAMING && ctx->fast_ok && !a.from_data && !a.empirical && !a.vr_beta_dep &&
                    a.n_mu <= 1024 && a.n_x <= 2048 && !getenv("VICTOR_HIP_FORCE_GENERIC");
  size_t lds;
  if (fast) {
    lds = (size_t)make_fast_plan(a.n_mu, a.n_x, a.sv.n_int, a.xi.n_int, nlr, a.n_beta_r).total * sizeof(double);
  } else {
    lds = (size_t)make_plan(a.n_mu, a.n_x, a.n_ell, a.sv.n_int, a.vr.n_int, a.xi.n_int, nlr, a.n_beta_r).total *
          sizeof(double);
  }
  if (lds > 160 * 1024) return fail(ctx, VK_E_ARG, "tables need %zu bytes of LDS (> 160 KiB)", lds);
  const long long groups = (a.n_s + a.sbins_per_item - 1) / a.sbins_per_item;
  const long long items = a.n * groups;
  // over-subscribe: 27.9 / 25.9 / 24.8 / 24.0 ms at 4 / 8 / 16 / 64 workgroups per CU on BOSS x 65536 (4 are resident)
  const char* pcap_env = getenv("VICTOR_HIP_POINT_CAP");            // tuning knob: workgroups per CU in the launch
  const long long cap = (pcap_env ? atoll(pcap_env) : 64LL) * ctx->n_cu;
  const int grid = (int)(items < cap ? items : cap);
  // lanes-over-batch variant: batch-constant tables and enough points to fill the chip with n_s * n/64 waves
  const char* mapping = getenv("VICTOR_HIP_MAPPING");
  const bool lanes_ok = fast && a.n_beta_r == 0 && !a.matter_lb;
  // One wave per (s bin, 64-point chunk).  The kernel is register-limited to 4 waves per SIMD, i.e. 16 per CU, so the
  // chip holds n_cu*16 waves at a time; the last round of waves is only partly filled.  The lanes kernel is ~1.2x
  // faster per integrand than the point-major one (81 vs 92 VALU instructions), so it wins once that fill
  // efficiency exceeds ~0.85.
  const long long waves = ((a.n + 63) >> 6) * (long long)a.n_s;
  const long long slots = 16LL * ctx->n_cu;
  const long long rounds = (waves + slots - 1) / slots;
  const double fill = (double)waves / (double)(rounds * slots);
  const bool lanes = lanes_ok && (mapping ? !strcmp(mapping, "lanes") : fill >= 0.85);
  if (lanes) {
    ctx->last_kernel = "vk_theory_lanes_kernel";
    const size_t lds_l = (size_t)make_lanes_plan(a.n_mu, a.n_x, a.sv.n_int, a.xi.n_int, nlr).total * sizeof(double);
    const long long blocks = (waves + kWaves - 1) / kWaves;
    // Many more workgroups than fit at once: letting the dispatcher refill CUs as workgroups retire measured
    // 38.1 / 36.0 / 34.6 / 33.8 ms at 4 / 8 / 16 / 64 workgroups per CU on the bench workload (4 are resident)
    const char* cap_env = getenv("VICTOR_HIP_LANES_CAP");          // tuning knob: workgroups per CU in the launch
    const long long capl = (cap_env ? atoll(cap_env) : 64LL) * ctx->n_cu;
    const int grid_l = (int)(blocks < capl ? blocks : capl);
    switch (nlr) {
      case 1: return launch_lanes_nl<1>(ctx, a, grid_l, lds_l);
      case 2: return launch_lanes_nl<2>(ctx, a, grid_l, lds_l);
      case 3: return launch_lanes_nl<3>(ctx, a, grid_l, lds_l);
    }
  }
  // cells variant: one workgroup per point with the velocity loop innermost; needs enough points to fill the chip
  // and n_mu >= 64 (a wave's 64 cells must not straddle more than two s bins)
  const bool cells_ok = fast && a.n_mu >= 64 && a.n_mu <= 4096 && a.n_x <= 2048;
  const bool cells = cells_ok && (mapping ? !strcmp(mapping, "cells") : a.n >= 4LL * ctx->n_cu);
  if (cells) {
    ctx->last_kernel = "vk_theory_cells_kernel";
    const size_t lds_c =
        (size_t)make_cells_plan(a.n_mu, a.n_x, a.n_s, a.n_ell, a.sv.n_int, a.xi.n_int, nlr, a.n_beta_r).total * sizeof(double);
    if (lds_c > 160 * 1024) return fail(ctx, VK_E_ARG, "tables need %zu bytes of LDS (> 160 KiB)", lds_c);
    const long long capc = (pcap_env ? atoll(pcap_env) : 64LL) * ctx->n_cu;
    const int grid_c = (int)(a.n < capc ? a.n : capc);
    switch (nlr) {
      case 1: return launch_cells_nl<1>(ctx, a, grid_c, lds_c);
      case 2: return launch_cells_nl<2>(ctx, a, grid_c, lds_c);
      case 3: return launch_cells_nl<3>(ctx, a, grid_c, lds_c);
    }
  }
  ctx->last_kernel = fast ? "vk_theory_fast_kernel" : "vk_theory_kernel";
  if (fast) {
    switch (nlr) {
      case 1: return launch_fast_nl<1>(ctx, a, grid, lds);
      case 2: return launch_fast_nl<2>(ctx, a, grid, lds);
      case 3: return launch_fast_nl<3>(ctx, a, grid, lds);
    }
    return fail(ctx, VK_E_ARG, "bad number of real-space multipoles %d", nlr);
  }
  switch (a.rsd) {
    case VK_RSD_STREAMING: return launch_generic<VK_RSD_STREAMING>(ctx, a, nlr, grid, lds);
    case VK_RSD_DISPERSION: return launch_generic<VK_RSD_DISPERSION>(ctx, a, nlr, grid, lds);
    case VK_RSD_KAISER: return launch_generic<VK_RSD_KAISER>(ctx, a, nlr, grid, lds);
    case VK_RSD_EUCLID: return launch_generic<VK_RSD_EUCLID>(ctx, a, nlr, grid, lds);
  }
  return fail(ctx, VK_E_ARG, "unknown rsd_model %d", a.rsd);
}

int launch_like(vk_ctx* ctx, const vk_eval_opts* o, const double* d_params, const double* d_theory, long long n,
                double* d_lnl, double* d_chi2) {
  if (n <= 0) return VK_OK;
  LikeArgs a{};
  a.params = d_params;
  a.theory = d_theory;
  a.n = n;
  a.N = ctx->N;
  a.n_beta_d = ctx->n_beta_d;
  a.beta_d = ctx->d_beta_d;
  a.data = ctx->d_data;
  a.n_beta_c = ctx->n_beta_c;
  a.beta_c = ctx->d_beta_c;
  a.prec = ctx->d_prec;
  a.logdet = ctx->d_logdet;
  a.eig = ctx->d_eig;
  a.like_form = o->like_form;
  a.nmocks = o->nmocks;
  a.nparams = o->nparams;
  a.lnl = d_lnl;
  a.chi2 = d_chi2;
  const long long blocks = (n + kWaves - 1) / kWaves;
  const long long cap = 16LL * ctx->n_cu;
  const int grid = (int)(blocks < cap ? blocks : cap);
  const size_t lds = (size_t)kWaves * ctx->N * sizeof(double);
  hipLaunchKernelGGL(vk_like_kernel, dim3(grid), dim3(kBlock), lds, ctx->stream, a);
  VK_HIP(ctx, hipGetLastError());
  return VK_OK;
}

void harvest_timing(vk_ctx* ctx) {
  if (!ctx->pending) return;
  float t0 = 0, t1 = 0;
  if (hipEventSynchronize(ctx->ev[2]) == hipSuccess && hipEventElapsedTime(&t0, ctx->ev[0], ctx->ev[1]) == hipSuccess &&
      hipEventElapsedTime(&t1, ctx->ev[1], ctx->ev[2]) == hipSuccess) {
    ctx->theory_ms += t0;
    ctx->like_ms += t1;
    ctx->launches += 1;
  }
  ctx->pending = false;
}

// ---- RCCL via dlopen -------------------------------------------------------------------------------
typedef struct { char internal[VK_COMM_ID_BYTES]; } rccl_id_t;
typedef int (*fn_get_id)(rccl_id_t*);
typedef int (*fn_init_rank)(void**, int, rccl_id_t, int);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);

void* open_rccl() {
  static void* lib = nullptr;
  if (lib) return lib;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names) {
    lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  return lib;
}

}  // namespace

extern "C" {

int vk_abi_version(void) { return VK_ABI_VERSION; }

int vk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void vk_default_opts(vk_eval_opts* o) {
  if (!o) return;
  memset(o, 0, sizeof *o);
  o->rsd_model = VK_RSD_STREAMING;
  o->assume_isotropic = 1;
  o->rescale_from_ap = 0;
  o->like_form = VK_LIKE_GAUSSIAN;
  o->nmocks = 1;
  o->nparams = 0;
  o->kaiser_approx = 0;
  o->kaiser_coord_shift = 1;
  o->niter = 5;
  o->from_data = 0;
  o->empirical_corr = 0;
}

const char* vk_last_error(const vk_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

const char* vk_last_kernel(const vk_ctx* ctx) { return ctx ? ctx->last_kernel : "none"; }

static int check_pp(const vk_pp* p, const char* name, std::string* err) {
  char buf[256];
  if (p->n_int < 1 || !p->knots || !p->coef || p->lead < 0 || p->lead > 1 || p->lead >= p->n_int + (p->inv_h > 0 ? 0 : 1)) {
    snprintf(buf, sizeof buf, "table '%s' is malformed", name);
    *err = buf;
    return VK_E_ARG;
  }
  for (int i = 0; i < p->n_int; ++i)
    if (!(p->knots[i + 1] > p->knots[i])) {
      snprintf(buf, sizeof buf, "table '%s': knots must be strictly increasing", name);
      *err = buf;
      return VK_E_ARG;
    }
  return VK_OK;
}

vk_ctx* vk_create(const vk_tables* t, int device, char* err, size_t errlen) {
  auto bail = [&](const std::string& msg) -> vk_ctx* {
    g_create_err = msg;
    if (err && errlen) {
      strncpy(err, msg.c_str(), errlen - 1);
      err[errlen - 1] = 0;
    }
    return nullptr;
  };
  if (!t) return bail("tables is NULL");
  if (t->n_s < 1 || t->n_mu < 2 || t->n_x < 3 || t->n_ell < 1 || t->n_ell > kMaxEll || t->n_ell_r < 1 ||
      t->n_ell_r > kMaxEll)
    return bail("bad grid sizes (need n_s>=1, n_mu>=2, n_x>=3, 1<=n_ell<=3, 1<=n_ell_r<=3)");
  if (!t->s || !t->mu || !t->w_ell || !t->x || !t->w_x) return bail("grid arrays missing");
  if (!(t->template_sigma8 > 0) || !(t->iaH > 0)) return bail("iaH and template_sigma8 must be positive");
  std::string e;
  if (check_pp(&t->xi, "xi", &e) || check_pp(&t->vr, "vr", &e) || check_pp(&t->sv, "sv", &e)) return bail(e);
  if (t->n_beta_r == 1 || t->n_beta_d == 1 || t->n_beta_c == 1) return bail("beta grids need at least 2 nodes");
  if (t->n_beta_r > 0 && !t->beta_r) return bail("beta_r missing");
  if (t->vr_beta_dep && t->n_beta_r < 2) return bail("beta-dependent velocity tables need the beta_r grid");
  if (t->matter_model < VK_MATTER_TEMPLATE || t->matter_model > VK_MATTER_VELOCITY_TEMPLATE) return bail("unknown matter_model");
  if (t->sv_n_mu != 0 && (t->sv_n_mu < 4 || !t->sv_mu || !t->sv2d)) return bail("anisotropic dispersion template needs >= 4 mu nodes and its patches");
  const int N = t->n_ell * t->n_s;
  if (t->data || t->prec) {
    if (!t->data || !t->prec) return bail("data and prec must be given together");
    if (t->n_beta_d > 0 && !t->beta_d) return bail("beta_d missing");
    if (t->n_beta_c > 0 && (!t->beta_c || !t->logdet || !t->eig)) return bail("beta_c/logdet/eig missing");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bail("no HIP device available");
  if (device < 0 || device >= ndev) return bail("device index out of range");

  vk_ctx* ctx = new (std::nothrow) vk_ctx();
  if (!ctx) return bail("out of memory");
  ctx->device = device;
  auto hip_bail = [&](hipError_t code, const char* what) -> vk_ctx* {
    std::string msg = std::string(what) + ": " + hipGetErrorString(code);
    vk_destroy(ctx);
    return bail(msg);
  };
  hipError_t rc;
  if ((rc = hipSetDevice(device)) != hipSuccess) return hip_bail(rc, "hipSetDevice");
  hipDeviceProp_t prop;
  if ((rc = hipGetDeviceProperties(&prop, device)) != hipSuccess) return hip_bail(rc, "hipGetDeviceProperties");
  ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if ((rc = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
    return hip_bail(rc, "hipStreamCreate");
  for (auto& evt : ctx->ev)
    if ((rc = hipEventCreate(&evt)) != hipSuccess) return hip_bail(rc, "hipEventCreate");

  ctx->n_s = t->n_s; ctx->n_mu = t->n_mu; ctx->n_x = t->n_x; ctx->n_ell = t->n_ell; ctx->n_ell_r = t->n_ell_r;
  ctx->n_beta_r = t->n_beta_r; ctx->N = N; ctx->iaH = t->iaH; ctx->template_sigma8 = t->template_sigma8;
  ctx->n_beta_d = t->data ? t->n_beta_d : 0;
  ctx->n_beta_c = t->data ? t->n_beta_c : 0;

  {
    bool ok = t->xi.inv_h > 0 && t->xi.lead == 0 && t->sv.inv_h > 0 && t->sv.lead == 0 && t->vr.inv_h > 0 &&
              t->vr.lead == 1 && t->vr.n_int == t->xi.n_int + 1;
    if (ok) {
      const double tol = 1e-12 * fabs(t->xi.knots[t->xi.n_int]);
      for (int i = 0; i <= t->xi.n_int && ok; ++i) ok = fabs(t->vr.knots[i + 1] - t->xi.knots[i]) <= tol;
      ok = ok && fabs(t->vr.inv_h - t->xi.inv_h) <= 1e-12 * t->xi.inv_h;
    }
    ctx->fast_ok = ok && !t->vr_beta_dep && t->sv_n_mu == 0;
    ctx->matter_vt = t->matter_model == VK_MATTER_VELOCITY_TEMPLATE;
    ctx->vt_amp = t->vt_amp;
    ctx->matter_lb = t->matter_model == VK_MATTER_LINEAR_BIAS;
    ctx->vr_beta_dep = t->vr_beta_dep ? 1 : 0;
  }
  Uploader up;
  const size_t o_s = up.add(t->s, t->n_s), o_mu = up.add(t->mu, t->n_mu),
               o_w = up.add(t->w_ell, (size_t)t->n_ell * t->n_mu), o_x = up.add(t->x, t->n_x),
               o_wx = up.add(t->w_x, t->n_x);
  const double zero_one[2] = {0.0, 1.0};
  const size_t o_x1 = up.add(zero_one, 2);
  const size_t o_br = t->n_beta_r > 0 ? up.add(t->beta_r, t->n_beta_r) : 0;
  const size_t xi_coef_n = t->n_beta_r > 0 ? (size_t)t->n_ell_r * (t->n_beta_r - 1) * t->xi.n_int * 16
                                           : (size_t)t->n_ell_r * t->xi.n_int * 4;
  const size_t o_xik = up.add(t->xi.knots, t->xi.n_int + 1), o_xic = up.add(t->xi.coef, xi_coef_n);
  const size_t vr_coef_n = t->vr_beta_dep ? (size_t)2 * (t->n_beta_r - 1) * t->vr.n_int * 16 : (size_t)kVrVars * t->vr.n_int * 4;
  const size_t o_vrk = up.add(t->vr.knots, t->vr.n_int + 1), o_vrc = up.add(t->vr.coef, vr_coef_n);
  const size_t o_svk = up.add(t->sv.knots, t->sv.n_int + 1),
               o_svc = up.add(t->sv.coef, t->sv_n_mu ? 4 : (size_t)t->sv.n_int * 4);   // 1-D coefficients unused with sv2d
  size_t o_svmu = 0, o_sv2d = 0;
  if (t->sv_n_mu) {
    o_svmu = up.add(t->sv_mu, t->sv_n_mu);
    o_sv2d = up.add(t->sv2d, (size_t)t->sv.n_int * (t->sv_n_mu - 1) * 16);
  }
  size_t o_bd = 0, o_data = 0, o_bc = 0, o_prec = 0, o_ld = 0, o_eig = 0;
  if (t->data) {
    if (t->n_beta_d > 0) {
      o_bd = up.add(t->beta_d, t->n_beta_d);
      o_data = up.add(t->data, (size_t)(t->n_beta_d - 1) * N * 4);
    } else {
      o_data = up.add(t->data, N);
    }
    if (t->n_beta_c > 0) {
      o_bc = up.add(t->beta_c, t->n_beta_c);
      o_prec = up.add(t->prec, (size_t)t->n_beta_c * N * N);
      o_ld = up.add(t->logdet, t->n_beta_c);
      o_eig = up.add(t->eig, (size_t)t->n_beta_c * N);
    } else {
      o_prec = up.add(t->prec, (size_t)N * N);
    }
  }
  const size_t bytes = up.host.size() * sizeof(double);
  if ((rc = hipMalloc((void**)&ctx->d_tables, bytes)) != hipSuccess) return hip_bail(rc, "hipMalloc(tables)");
  if ((rc = hipMemcpy(ctx->d_tables, up.host.data(), bytes, hipMemcpyHostToDevice)) != hipSuccess)
    return hip_bail(rc, "hipMemcpy(tables)");
  const double* base = ctx->d_tables;
  ctx->d_x1 = base + o_x1; ctx->d_w1 = base + o_x1 + 1;
  ctx->d_s = base + o_s; ctx->d_mu = base + o_mu; ctx->d_w = base + o_w; ctx->d_x = base + o_x; ctx->d_wx = base + o_wx;
  ctx->d_beta_r = t->n_beta_r > 0 ? base + o_br : nullptr;
  auto view = [&](const vk_pp& p, size_t ok, size_t oc) {
    PPView v;
    v.n_int = p.n_int; v.lead = p.lead; v.inv_h = p.inv_h; v.knots = base + ok; v.coef = base + oc;
    return v;
  };
  ctx->xi = view(t->xi, o_xik, o_xic);
  ctx->vr = view(t->vr, o_vrk, o_vrc);
  ctx->sv = view(t->sv, o_svk, o_svc);
  ctx->sv_n_mu = t->sv_n_mu;
  ctx->sv_mu_inv_h = t->sv_mu_inv_h;
  if (t->sv_n_mu) {
    ctx->d_sv_mu = base + o_svmu;
    ctx->d_sv2d = base + o_sv2d;
  }
  if (t->data) {
    ctx->d_beta_d = t->n_beta_d > 0 ? base + o_bd : nullptr;
    ctx->d_data = base + o_data;
    ctx->d_beta_c = t->n_beta_c > 0 ? base + o_bc : nullptr;
    ctx->d_prec = base + o_prec;
    ctx->d_logdet = t->n_beta_c > 0 ? base + o_ld : nullptr;
    ctx->d_eig = t->n_beta_c > 0 ? base + o_eig : nullptr;
  }
  return ctx;
}

void vk_destroy(vk_ctx* ctx) {
  if (!ctx) return;
  if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
  if (ctx->comm) vk_comm_destroy(ctx);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_tables) (void)hipFree(ctx->d_tables);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  for (auto& evt : ctx->ev)
    if (evt) (void)hipEventDestroy(evt);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

void* vk_device_alloc(vk_ctx* ctx, size_t bytes) {
  if (!ctx) return nullptr;
  void* p = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) {
    ctx->err = "hipMalloc failed";
    return nullptr;
  }
  return p;
}

void vk_device_free(vk_ctx* ctx, void* ptr) {
  if (ctx && ptr) {
    (void)hipSetDevice(ctx->device);
    (void)hipFree(ptr);
  }
}

int vk_memcpy_h2d(vk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return VK_E_ARG;
  VK_HIP(ctx, hipSetDevice(ctx->device));
  VK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VK_OK;
}

int vk_memcpy_d2h(vk_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return VK_E_ARG;
  VK_HIP(ctx, hipSetDevice(ctx->device));
  VK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VK_OK;
}

int vk_sync(vk_ctx* ctx) {
  if (!ctx) return VK_E_ARG;
  VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  harvest_timing(ctx);
  return VK_OK;
}

int vk_timing_enable(vk_ctx* ctx, int on) {
  if (!ctx) return VK_E_ARG;
  harvest_timing(ctx);
  ctx->timing = on != 0;
  return VK_OK;
}

int vk_timing_read(vk_ctx* ctx, double* theory_ms, double* like_ms, int64_t* launches, int reset) {
  if (!ctx) return VK_E_ARG;
  harvest_timing(ctx);
  if (theory_ms) *theory_ms = ctx->theory_ms;
  if (like_ms) *like_ms = ctx->like_ms;
  if (launches) *launches = ctx->launches;
  if (reset) {
    ctx->theory_ms = ctx->like_ms = 0;
    ctx->launches = 0;
  }
  return VK_OK;
}

int vk_eval_batch_device_async(vk_ctx* ctx, const vk_eval_opts* opts, const double* d_params, int64_t n,
                               double* d_lnl, double* d_chi2, double* d_theory_ws) {
  if (!ctx) return VK_E_ARG;
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (n < 0 || (n > 0 && (!d_params || !d_theory_ws))) return fail(ctx, VK_E_ARG, "bad device buffers");
  const bool want_like = d_lnl || d_chi2;
  if (want_like && !ctx->d_data) return fail(ctx, VK_E_ARG, "context was created without a data vector");
  if (n == 0) return VK_OK;
  VK_HIP(ctx, hipSetDevice(ctx->device));
  TheoryArgs a{};
  int nlr = 1;
  rc = theory_args(ctx, opts, &a, &nlr);
  if (rc) return rc;
  a.params = d_params;
  a.n = n;
  a.n_s = ctx->n_s; a.n_mu = ctx->n_mu; a.n_ell = ctx->n_ell;
  a.s = ctx->d_s; a.mu = ctx->d_mu; a.w_ell = ctx->d_w;
  a.out = d_theory_ws;
  const bool timed = ctx->timing && want_like;
  if (timed) {
    harvest_timing(ctx);
    VK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  }
  rc = launch_theory(ctx, a, nlr);
  if (rc) return rc;
  if (timed) VK_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
  if (want_like) {
    rc = launch_like(ctx, opts, d_params, d_theory_ws, n, d_lnl, d_chi2);
    if (rc) return rc;
  }
  if (timed) {
    VK_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    ctx->pending = true;
  }
  return VK_OK;
}

int vk_eval_batch(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n, double* lnl, double* chi2,
                  double* theory) {
  if (!ctx) return VK_E_ARG;
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (n < 0 || (n > 0 && !params)) return fail(ctx, VK_E_ARG, "params is NULL");
  if (n == 0) return VK_OK;
  if ((lnl || chi2) && !ctx->d_data) return fail(ctx, VK_E_ARG, "context was created without a data vector");
  VK_HIP(ctx, hipSetDevice(ctx->device));
  const size_t nb_par = (size_t)n * VK_NPAR * sizeof(double);
  const size_t nb_th = (size_t)n * ctx->N * sizeof(double);
  const size_t nb_out = (size_t)n * sizeof(double);
  rc = ensure_scratch(ctx, nb_par + nb_th + 2 * nb_out);
  if (rc) return rc;
  double* d_par = ctx->d_scratch;
  double* d_th = d_par + (size_t)n * VK_NPAR;
  double* d_lnl = d_th + (size_t)n * ctx->N;
  double* d_chi = d_lnl + n;
  VK_HIP(ctx, hipMemcpyAsync(d_par, params, nb_par, hipMemcpyHostToDevice, ctx->stream));
  rc = vk_eval_batch_device_async(ctx, opts, d_par, n, lnl ? d_lnl : nullptr, chi2 ? d_chi : nullptr, d_th);
  if (rc) return rc;
  if (lnl) VK_HIP(ctx, hipMemcpyAsync(lnl, d_lnl, nb_out, hipMemcpyDeviceToHost, ctx->stream));
  if (chi2) VK_HIP(ctx, hipMemcpyAsync(chi2, d_chi, nb_out, hipMemcpyDeviceToHost, ctx->stream));
  if (theory) VK_HIP(ctx, hipMemcpyAsync(theory, d_th, nb_th, hipMemcpyDeviceToHost, ctx->stream));
  return vk_sync(ctx);
}

static int general_grid(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n, const double* s,
                        int32_t n_s, const double* mu, int32_t n_mu, const double* w_ell, int32_t n_ell, double* out,
                        bool project) {
  if (!ctx) return VK_E_ARG;
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (n < 0 || n_s < 1 || n_mu < 2 || !params || !s || !mu || !out) return fail(ctx, VK_E_ARG, "bad arguments");
  if (project && (n_ell < 1 || n_ell > kMaxEll || !w_ell)) return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
  if (n == 0) return VK_OK;
  VK_HIP(ctx, hipSetDevice(ctx->device));
  const int ne = project ? n_ell : 1;
  const size_t out_n = project ? (size_t)n * n_ell * n_s : (size_t)n * n_mu * n_s;
  const size_t grid_n = (size_t)n_s + n_mu + (size_t)ne * n_mu + 4;
  const size_t total = ((size_t)n * VK_NPAR + grid_n + out_n) * sizeof(double);
  rc = ensure_scratch(ctx, total);
  if (rc) return rc;
  double* d_par = ctx->d_scratch;
  double* d_s = d_par + (size_t)n * VK_NPAR;
  double* d_mu = d_s + n_s;
  double* d_w = d_mu + n_mu;
  double* d_out = d_w + (size_t)ne * n_mu;
  d_out = (double*)(((uintptr_t)d_out + 15) & ~(uintptr_t)15);
  VK_HIP(ctx, hipMemcpyAsync(d_par, params, (size_t)n * VK_NPAR * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  VK_HIP(ctx, hipMemcpyAsync(d_s, s, n_s * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  VK_HIP(ctx, hipMemcpyAsync(d_mu, mu, n_mu * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (project)
    VK_HIP(ctx, hipMemcpyAsync(d_w, w_ell, (size_t)n_ell * n_mu * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  else
    VK_HIP(ctx, hipMemsetAsync(d_w, 0, (size_t)n_mu * sizeof(double), ctx->stream));
  TheoryArgs a{};
  int nlr = 1;
  rc = theory_args(ctx, opts, &a, &nlr);
  if (rc) return rc;
  a.params = d_par;
  a.n = n;
  a.n_s = n_s; a.n_mu = n_mu; a.n_ell = ne;
  a.s = d_s; a.mu = d_mu; a.w_ell = d_w;
  a.out = d_out;
  if (project) {
    rc = launch_theory(ctx, a, nlr);
    if (rc) return rc;
  } else {
    const LdsPlan pl = make_plan(a.n_mu, a.n_x, a.n_ell, a.sv.n_int, a.vr.n_int, a.xi.n_int, nlr, a.n_beta_r);
    const size_t lds = (size_t)pl.total * sizeof(double);
    if (lds > 160 * 1024) return fail(ctx, VK_E_ARG, "tables need %zu bytes of LDS (> 160 KiB)", lds);
    const long long cap = 8LL * ctx->n_cu;
    const int grid = (int)(n < cap ? n : cap);
    a.sbins_per_item = 1;
    a.team = 1;
    switch (a.rsd) {
      case VK_RSD_STREAMING: rc = launch_xi_smu<VK_RSD_STREAMING>(ctx, a, nlr, grid, lds); break;
      case VK_RSD_DISPERSION: rc = launch_xi_smu<VK_RSD_DISPERSION>(ctx, a, nlr, grid, lds); break;
      case VK_RSD_KAISER: rc = launch_xi_smu<VK_RSD_KAISER>(ctx, a, nlr, grid, lds); break;
      default: rc = launch_xi_smu<VK_RSD_EUCLID>(ctx, a, nlr, grid, lds); break;
    }
    if (rc) return rc;
  }
  VK_HIP(ctx, hipMemcpyAsync(out, d_out, out_n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VK_OK;
}

int vk_theory_batch(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n, const double* s,
                    int32_t n_s, const double* mu, int32_t n_mu, const double* w_ell, int32_t n_ell, double* out) {
  return general_grid(ctx, opts, params, n, s, n_s, mu, n_mu, w_ell, n_ell, out, true);
}

int vk_xi_smu_batch(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n, const double* s,
                    int32_t n_s, const double* mu, int32_t n_mu, double* out) {
  return general_grid(ctx, opts, params, n, s, n_s, mu, n_mu, nullptr, 0, out, false);
}

// ---- RCCL -----------------------------------------------------------------------------------------
int vk_comm_unique_id(char* id_out) {
  void* lib = open_rccl();
  if (!lib || !id_out) return VK_E_RCCL;
  auto get = (fn_get_id)dlsym(lib, "ncclGetUniqueId");
  if (!get) return VK_E_RCCL;
  rccl_id_t id;
  if (get(&id) != 0) return VK_E_RCCL;
  memcpy(id_out, id.internal, VK_COMM_ID_BYTES);
  return VK_OK;
}

int vk_comm_init(vk_ctx* ctx, const char* id, int rank, int nranks) {
  if (!ctx || !id) return VK_E_ARG;
  void* lib = open_rccl();
  if (!lib) return fail(ctx, VK_E_RCCL, "cannot load librccl: %s", dlerror());
  auto init = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
  if (!init) return fail(ctx, VK_E_RCCL, "ncclCommInitRank not found");
  VK_HIP(ctx, hipSetDevice(ctx->device));
  rccl_id_t uid;
  memcpy(uid.internal, id, VK_COMM_ID_BYTES);
  int rc = init(&ctx->comm, nranks, uid, rank);
  if (rc != 0) {
    auto es = (fn_errstr)dlsym(lib, "ncclGetErrorString");
    ctx->comm = nullptr;
    return fail(ctx, VK_E_RCCL, "ncclCommInitRank failed: %s", es ? es(rc) : "?");
  }
  ctx->rccl_lib = lib;
  return VK_OK;
}

int vk_comm_allgather_async(vk_ctx* ctx, const double* d_send, double* d_recv, int64_t count) {
  if (!ctx || !ctx->comm) return fail(ctx, VK_E_RCCL, "communicator not initialised");
  auto ag = (fn_allgather)dlsym(ctx->rccl_lib, "ncclAllGather");
  if (!ag) return fail(ctx, VK_E_RCCL, "ncclAllGather not found");
  const int kNcclDouble = 8;  // ncclFloat64 in rccl.h
  int rc = ag(d_send, d_recv, (size_t)count, kNcclDouble, ctx->comm, ctx->stream);
  if (rc != 0) return fail(ctx, VK_E_RCCL, "ncclAllGather failed (%d)", rc);
  return VK_OK;
}

int vk_comm_destroy(vk_ctx* ctx) {
  if (!ctx || !ctx->comm) return VK_OK;
  auto destroy = (fn_destroy)dlsym(ctx->rccl_lib, "ncclCommDestroy");
  if (destroy) destroy(ctx->comm);
  ctx->comm = nullptr;
  return VK_OK;
}

}  // extern "C"
