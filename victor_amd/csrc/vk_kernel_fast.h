// vk_kernel_fast.h: point-major fast theory kernel and the shared fast-path building blocks - part of libvictor_hip.so (see victor_hip.hip for the overview and DESIGN.md section 5).
#pragma once
#include "vk_common.h"
#include "vk_kernel_like.h"

namespace vk {

// --------------------------------------------------------------------------------------------------
// Shared building blocks of the fast theory kernels (point-major, lanes, cells).
// They run on the UNIFIED GRID the host prepares when the r grid and the sigma_v grid are uniform and commensurate
// (vk_tables.uni_*): every table is re-expressed on the common refinement of the two grids, extended down to
// u <= 0, in interval units, so
//   * the kernels work in INDEX UNITS: every length of a point is multiplied by k = 1/(c h) once per point / mu row,
//     and the interval coordinate of an integrand point is t = r2' * rsqrt(r2') + off - one fma on the refined
//     1/sqrt, no separate r, no u = r/c; one clamp pair + v_cvt + v_fract then yields THE interval and local
//     coordinate for all five cubics (sigma_v, V, xi_0, xi_2, xi_4), with no knot read and no second index;
//   * the five cubics of a refined interval sit in one LDS record, padded to 4*(2+NLR)+2 doubles so that the
//     ds_read_b128 of 16 consecutive intervals hit distinct banks;
//   * refined intervals outside a table's range hold its (constant) boundary value, which reproduces the
//     clamped spline evaluation of the reference (FITPACK ext=3) exactly; the V table's leading interval
//     [0.01, r_0] of ccf_model.py:625 is part of the grid, and its clamp V(u < 0.01) = V(0.01) is the lower clamp
//     of t itself (every other table is constant down there);
//   * 1/r comes from one refined v_rsq_f64, 1/sigma_v from a refined v_rcp_f64, exp(-z^2/2) from a 256-entry
//     2^(j/256) table and a degree-4 polynomial on the pre-scaled y = z sqrt(128/ln2) (vk_devmath.h; ~2 ulp).
// The exp table and the records sit at fixed LDS offsets (0 and 2 KB) so their addresses are immediates.
// --------------------------------------------------------------------------------------------------
typedef double vk_d2 __attribute__((ext_vector_type(2)));
constexpr int kMuRec = 6;   // {mu, sqrt(1-mu^2), W_0, W_1, W_2, pad}
constexpr int kEtabOff = 0;                  // doubles; must stay 0: vkm::exp_gauss addresses the table from LDS address 0
// EXPT: form of the exp table at the start of LDS (vk_devmath.h: ExpCfg); the records follow it
template <int EXPT> __host__ __device__ constexpr int recs_off() { return vkm::ExpCfg<EXPT>::kDoubles; }
__host__ __device__ constexpr int recs_off_rt(int expt) { return expt ? vkm::ExpCfg<1>::kDoubles : vkm::ExpCfg<0>::kDoubles; }

__host__ __device__ constexpr int uni_stride(int nlr) { return 4 * (2 + nlr) + 2; }   // doubles per refined interval
// exp table + records (+ two sentinel records and the u16 look-up table of the union-grid mode), at fixed offsets
__host__ __device__ constexpr int fast_fixed_doubles(int uni_n, int nlr, int lut_n, int expt = 0) {
  return recs_off_rt(expt) + (uni_n + (lut_n > 0 ? 2 : 0)) * uni_stride(nlr) + (lut_n + 3) / 4;
}

// GRID = 0: the unified grid is a uniform lattice (index arithmetic).  GRID = 1: it is the union of arbitrary knot
// sets (vk_tables.uni_lut_n > 0): a u16 look-up table gives the interval of the cell's left edge, comparisons with
// the next two knots (kept in the pad slots of the next two records; a cell holds at most two knots) correct it, and
// the record's own pad holds its left knot and 1/width for the local coordinate.  Costs 8 more VALU instructions and
// three more LDS reads per integrand point.
struct FastConsts {
  double inv_h;                   // callers form k = inv_h / c per point and scale their lengths by it (GRID 1: 1)
  double off, t_lo, n_eps;        // GRID 0: t = r' + off, clamped to [t_lo, n_eps]; t_lo is u = 0.01, the first V knot
                                  // GRID 1: u = r' clamped to [t_lo, n_eps] = [first knot, last knot)
  double inv_g;                   // GRID 1: cells of the look-up table per unit length
  double rlo2, rhi2;              // the clamped coordinate stays inside (t_lo, n_eps) while rlo2 <= X < rhi2 (cell_in_table),
                                  // X = r'^2 (full units) or r'^2 / 4 (half units, the streaming modes)
  int lut_off;                    // GRID 1: byte offset of the look-up table in LDS
  // SVA (anisotropic sigma_v(r, mu) template, bicubic patches in LDS; lattice form only)
  double sv_a, sv_b, sv_tmax;     // interval coordinate on the sigma_v r grid: ts = r' sv_a + sv_b, clamped to [0, sv_tmax]
  double mu_lo, mu_hi, mu_inv_h;  // mu box of the template; mu_inv_h > 0: uniform mu knots
  int sva_off, svmu_off, sv_nm;   // byte offsets of the patches and of the mu knots in LDS; mu intervals
};

// v_min_f64 / v_max_f64 without the canonicalising v_max hipcc puts in front of fmin()/fmax() for a bound it cannot
// prove quiet (the bounds are finite table constants; a NaN first operand yields the bound, i.e. a valid index)
__device__ __forceinline__ double vmin_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmax_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
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

// HALF: the caller works in half units (every length times k/2, see uni_point), so the squares it tests are r'^2 / 4
template <int NLR, int EXPT = 0>
__device__ __forceinline__ FastConsts make_fast_consts(const TheoryArgs& a, bool half, int sva_off_doubles = 0) {
  FastConsts fc;
  const double sq = half ? 0.25 : 1.0;
  fc.sv_a = fc.sv_b = fc.sv_tmax = fc.mu_lo = fc.mu_hi = fc.mu_inv_h = 0.0;
  fc.sva_off = fc.svmu_off = fc.sv_nm = 0;
  if (a.sv_n_mu > 0 && a.sva_doubles > 0 && a.uni_lut_n == 0) {
    // r' (index units of the unified lattice) -> interval coordinate of the template's own uniform r grid
    fc.sv_a = a.sv.inv_h / a.uni_inv_h;
    fc.sv_b = -a.sv.knots[0] * a.sv.inv_h;
    fc.sv_tmax = (double)a.sv.n_int * (1.0 - 0x1p-52);
    fc.sv_nm = a.sv_n_mu - 1;
    fc.mu_lo = a.sv_mu[0];
    fc.mu_hi = a.sv_mu[fc.sv_nm];
    fc.mu_inv_h = a.sv_mu_inv_h;
    fc.sva_off = sva_off_doubles * 8;
    fc.svmu_off = (sva_off_doubles + a.sv.n_int * fc.sv_nm * 16) * 8;
  }
  if (a.uni_lut_n > 0) {
    fc.inv_h = 1.0;
    fc.off = 0.0;
    fc.t_lo = a.uni_knots[0];
    fc.n_eps = a.uni_knots[a.uni_n] * (1.0 - 0x1p-52);
    fc.inv_g = a.uni_lut_inv_g;
    fc.rlo2 = sq * fc.t_lo * fc.t_lo * (1.0 + 1e-9);       // u = r' itself: inside [first knot, last knot) as for the lattice form below
    fc.rhi2 = sq * fc.n_eps * fc.n_eps * (1.0 - 1e-9);
    fc.lut_off = (recs_off<EXPT>() + (a.uni_n + 2) * uni_stride(NLR)) * 8;
  } else {
    fc.inv_h = a.uni_inv_h;
    fc.off = -a.uni_u0 * a.uni_inv_h;
    fc.t_lo = (a.vr.knots[0] - a.uni_u0) * a.uni_inv_h;
    fc.n_eps = (double)a.uni_n * (1.0 - 0x1p-52);
    fc.inv_g = 0.0;
    // margins of 1e-9 relative: r' = X * (4 / r') is good to a few ulp, so t cannot cross a bound the squares stay clear of
    const double lo = fc.t_lo - fc.off, hi = fc.n_eps - fc.off;
    fc.rlo2 = sq * lo * lo * (1.0 + 1e-9);
    fc.rhi2 = sq * hi * hi * (1.0 - 1e-9);
    fc.lut_off = 0;
  }
  return fc;
}

// Is the interval coordinate of every velocity node of this lane's (s, mu) cell inside the table, so that the clamp pair of
// `locate` is the identity?  With r_par' = s_par' - x_k' Bk and `xi_max` = max|x_k'| |Bk|,
//   s_perp'^2 + max(|s_par'| - xi_max, 0)^2 <= r'^2 <= s_perp'^2 + (|s_par'| + xi_max)^2
// by the monotonicity of the roundings (the lower bound says that the mu = 1 cells, s_perp' = 0, only come near r = 0 in the
// s bins the velocity nodes can reach).  False for NaN operands (the clamped form then yields a valid index as before).
// All lengths in the caller's units (half units: fc was made with half = true).
__device__ __forceinline__ bool cell_in_table(const FastConsts& fc, double s_par, double sperp2, double xi_max) {
  const double spx = fabs(s_par) + xi_max;
  const double spn = fmax(fabs(s_par) - xi_max, 0.0);
  return (fma(spn, spn, sperp2) >= fc.rlo2) && (fma(spx, spx, sperp2) < fc.rhi2);
}

// Stage the batch-constant parts of the records: sigma_v and V always, xi^r_l when it does not depend on beta;
// also the exp table.  All threads of the workgroup; `lds` is the start of dynamic LDS.
template <int NLR, int EXPT = 0>
__device__ __forceinline__ void stage_uni_records(const TheoryArgs& a, double* lds) {
  constexpr int stride = uni_stride(NLR);
  double* recs = lds + recs_off<EXPT>();
  const int tid = threadIdx.x;
  for (int e = tid; e < a.uni_n * 8; e += kBlock) recs[(e >> 3) * stride + (e & 7)] = a.uni_sv_v[e];
  if (a.n_beta_r == 0) {
    const double* src = (NLR > 1) ? a.uni_xic : a.uni_xi;      // anisotropic sum: the mu_r^2-power regrouping
    const int per_l = a.uni_n * 4;
    for (int e = tid; e < NLR * per_l; e += kBlock) {
      const int l = e / per_l, iq = e - l * per_l;
      recs[(iq >> 2) * stride + 8 + 4 * l + (iq & 3)] = src[e];
    }
  }
  const double* etab = EXPT ? a.exp_tab_rep : a.exp_tab;
  for (int j = tid; j < vkm::ExpCfg<EXPT>::kDoubles; j += kBlock) lds[kEtabOff + j] = etab[j];
  if (a.uni_lut_n > 0) {
    // union-grid mode: pad slots {left knot, 1/width}; two sentinel records whose left knot is the last knot; the table
    for (int q = tid; q <= a.uni_n + 1; q += kBlock) {
      const double left = a.uni_knots[q < a.uni_n ? q : a.uni_n];
      recs[q * stride + stride - 2] = left;
      recs[q * stride + stride - 1] = (q < a.uni_n) ? 1.0 / (a.uni_knots[q + 1] - left) : 0.0;
      if (q >= a.uni_n)
        for (int e = 0; e < stride - 2; ++e) recs[q * stride + e] = 0.0;
    }
    unsigned short* lut = reinterpret_cast<unsigned short*>(recs + (a.uni_n + 2) * stride);
    for (int c = tid; c < a.uni_lut_n; c += kBlock) lut[c] = a.uni_lut[c];
  }
}

// Per-point xi^r records when the real-space input depends on the reconstruction beta (PCHIP piece kb, extrapolating
// with the end pieces as PchipInterpolator does; ccf_model.py:323-326).  `bg` = beta grid in LDS.
// degree-6 polynomial in db, coefficients c[0..6] (the empirical_corr tables of a beta-dependent velocity profile)
__device__ __forceinline__ double poly6(const double* __restrict__ c, double db) {
  double v = c[6];
#pragma unroll
  for (int p = 5; p >= 0; --p) v = fma(v, db, c[p]);
  return v;
}

// `da`: LDS table of the dispersion model's v_r' (NULL in the streaming modes); `av`: the point's empirical_corr amplitude.
template <int NLR>
__device__ __forceinline__ void rebuild_uni_xi(const TheoryArgs& a, double* recs, const double* bg, double beta, double vs = 1.0,
                                               double av = 0.0, double* da = nullptr) {
  constexpr int stride = uni_stride(NLR);
  const int tid = late_tid();        // fresh per work item: nothing derived from it is carried through the integrand loops
  // PCHIP piece: last i in [1, n-2] with beta >= bg[i], else 0 - a count over the lanes for grids of up to 64 nodes
  int kb = 0;
  if (a.n_beta_r <= 64) {
    const int lane = tid & 63;
    kb = __popcll(__ballot(lane >= 1 && lane < a.n_beta_r - 1 && beta >= bg[lane < a.n_beta_r ? lane : 0]));
  } else {
    for (int i = 1; i < a.n_beta_r - 1; ++i) kb = (beta >= bg[i]) ? i : kb;
  }
  const double db = beta - bg[kb];
  const int per_l = a.uni_n * 4;
  const size_t stride_l = (size_t)(a.n_beta_r - 1) * per_l * 4;
  const double* src = ((NLR > 1) ? a.uni_xic : a.uni_xi) + (size_t)kb * per_l * 4;
  const int total = NLR * per_l;
  // four entries per thread per pass, their eight 16-byte coefficient loads in flight together
  for (int base = tid; base < total; base += 4 * kBlock) {
    vk_d2 c01[4], c23[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = min(base + k * kBlock, total - 1);
      const int l = (NLR > 2 && e >= 2 * per_l) ? 2 : ((NLR > 1 && e >= per_l) ? 1 : 0);
      const double* c = src + l * stride_l + (size_t)(e - l * per_l) * 4;
      c01[k] = *reinterpret_cast<const vk_d2*>(c);
      c23[k] = *reinterpret_cast<const vk_d2*>(c + 2);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = base + k * kBlock;
      if (e < total) {
        const int l = (NLR > 2 && e >= 2 * per_l) ? 2 : ((NLR > 1 && e >= per_l) ? 1 : 0);
        const int iq = e - l * per_l;
        recs[(iq >> 2) * stride + 8 + 4 * l + (iq & 3)] = fma(fma(fma(c23[k].y, db, c23[k].x), db, c01[k].y), db, c01[k].x);
      }
    }
  }
  if (a.vr_beta_dep) {   // linear_bias on a reconstructed real-space ccf: V1 follows xi^r_0(beta) (ccf_model.py:358-370)
    // with empirical_corr, V = V1 + av V2 and (dispersion model) v_r' from Ge1 + av Ge2 instead of Da (ccf_model.py:451-459):
    // products of PCHIP cubics, degree 6 in beta (vk_tables.uni_empb)
    const size_t var6 = (size_t)(a.n_beta_r - 1) * per_l * 7;
    const double* e6 = a.uni_empb + (size_t)kb * per_l * 7;
    for (int iq = tid; iq < per_l; iq += kBlock) {
      const double* c = a.uni_vb + ((size_t)kb * per_l + iq) * 4;
      double v = fma(fma(fma(c[3], db, c[2]), db, c[1]), db, c[0]);
      if (a.empirical) v = fma(av, poly6(e6 + (size_t)iq * 7, db), v);
      recs[(iq >> 2) * stride + 4 + (iq & 3)] = vs * v;
      if (da) {
        if (a.empirical) {
          da[iq] = fma(av, poly6(e6 + 2 * var6 + (size_t)iq * 7, db), poly6(e6 + var6 + (size_t)iq * 7, db));
        } else {
          const double* d = a.uni_dab + ((size_t)kb * per_l + iq) * 4;
          da[iq] = fma(fma(fma(d[3], db, d[2]), db, d[1]), db, d[0]);
        }
      }
    }
  }
}

// empirical_corr (ccf_model.py:451-455): v_r carries the factor (1 + Av delta), i.e. V = V1 + av V2 with the per-point
// av; the kernels that own a point per workgroup rewrite the V half of the records for it
template <int NLR>
__device__ __forceinline__ void rebuild_uni_v_emp(const TheoryArgs& a, double* recs, double av, double vs = 1.0) {
  constexpr int stride = uni_stride(NLR);
  for (int iq = late_tid(); iq < a.uni_n * 4; iq += kBlock)
    recs[(iq >> 2) * stride + 4 + (iq & 3)] = vs * fma(av, a.uni_v2[iq], a.uni_sv_v[(iq >> 2) * 8 + 4 + (iq & 3)]);
}

// V cubics of the records = vs * (batch-constant V1, kept unscaled in LDS at `v1`): the per-point amplitude of the mean
// velocity (FastPoint::AVk) folded into the table once per work item instead of once per integrand point
template <int NLR>
__device__ __forceinline__ void scale_uni_v(int uni_n, double* recs, const double* v1, double vs) {
  constexpr int stride = uni_stride(NLR);
  for (int iq = late_tid(); iq < uni_n * 4; iq += kBlock) recs[(iq >> 2) * stride + 4 + (iq & 3)] = vs * v1[iq];
}

// Per-item tables of the kernels that own a point per workgroup.  PV: fold AVk into the V cubics (streaming modes).
// `da`: the dispersion modes' LDS table of v_r' (NULL in the streaming modes) - rebuilt here when the velocity profile depends on beta.
template <int NLR, int PV>
__device__ __forceinline__ void rebuild_point_tables(const TheoryArgs& a, double* lds, int betar_off, int v1_off, double beta,
                                                     double av, double AVk, double* da = nullptr) {
  const double vs = PV ? AVk : 1.0;
  double* recs = lds + recs_off<0>();                        // the kernels that own a point per workgroup use the plain exp table
  if (a.n_beta_r > 0) rebuild_uni_xi<NLR>(a, recs, lds + betar_off, beta, vs, av, a.vr_beta_dep ? da : nullptr);
  if (a.empirical && !a.vr_beta_dep) rebuild_uni_v_emp<NLR>(a, recs, av, vs);
  else if (PV && !a.vr_beta_dep) scale_uni_v<NLR>(a.uni_n, recs, lds + v1_off, vs);
}

// per-point factors of the index-unit formulation (wave-uniform in the point-major and cells kernels, per lane in
// the lanes kernel).  The streaming modes work in HALF units - every length times k/2, so that twice the refined 1/sqrt comes
// out of four instructions (vkm::rsqrt_nr_x2) and the factors of two land where they cost nothing: X = r'^2/4,
// yy = 4/r', X yy = r', (num/2) yy = 2 mu_r, AVk halved, and the mu_r^2-power coefficients of the records divided by 4 and
// 16 on the host (vk_tables.uni_xic) - all exact.  The dispersion modes too (disp_value); kaiser / euclid_special keep full units.
struct FastPoint {
  double fa, fp2;         // from_data: c/apar and (c/aperp)^2 (fiducial coordinates of xi^r, see uni_point)
  double Gk, gD;          // dispersion model: aH^-1 v_r/r = -Gk V(u) / r' (half units: -Gk V(u) (4 / r'), Gk carries the 1/4)
                          // and aH^-1 v_r' = -gD Da(u) (see disp_value, kaiser_value)
  double k_perp, k_par;   // aperp k, apar k: s_perp' = s sqrt(1-mu^2) k_perp, s_par' = s mu k_par
  double Bk;              // sigma_v iaH_true k / kExpScale: r_par' = s_par' - x_k' Bk with x_k' = kExpScale x_k
  double AVk;             // kExpScale g / (3 iaH_true sigma_v): y = (x_k' + AVk V mu_r) / SV
};

__device__ __forceinline__ FastPoint make_fast_point(const PointScalars& ps, const FastConsts& fc, bool half) {
  FastPoint fp;
  const double hs = half ? 0.5 : 1.0;
  const double k = ps.inv_c * fc.inv_h;
  const double kl = k * hs;                                // the scale of the lengths that enter the integrand
  fp.k_perp = ps.aperp * kl;
  fp.k_par = ps.apar * kl;
  fp.Bk = ps.B * kl * (1.0 / vkm::kExpScale);
  fp.AVk = ps.A * vkm::kExpScale * hs;                     // multiplies V * (2 mu_r) in half units
  const double c = 1.0 / ps.inv_c;
  fp.fa = c * ps.inv_apar;
  fp.fp2 = (c * ps.inv_aperp) * (c * ps.inv_aperp);
  fp.Gk = ps.G * k * (half ? 0.25 : 1.0);
  fp.gD = ps.gD;
  return fp;
}

// Record and local coordinate of a radius: `x` is the interval coordinate t = r' + off (GRID 0) or the radius u = r'
// itself (GRID 1, union grid), not yet clamped.
// CL = 0: the caller has shown that t lies inside [t_lo, n_eps] for every lane (cell_in_table), the clamp pair is dropped.
template <int NLR, int GRID, int CL = 1, int EXPT = 0>
__device__ __forceinline__ const double* locate(const double* __restrict__ lds, const FastConsts& fc, double x,
                                                double& tq, int& qi) {
  constexpr int stride = uni_stride(NLR);
  constexpr int roff = recs_off<EXPT>();
  if (GRID == 0) {
    const double t = CL ? vmin_f64(vmax_f64(x, fc.t_lo), fc.n_eps) : x;
    tq = __builtin_amdgcn_fract(t);
    qi = (int)t;
    return lds_at(lds + roff, __mul24(qi, stride * 8));
  }
  const double u = CL ? vmin_f64(vmax_f64(x, fc.t_lo), fc.n_eps) : x;
  const int cell = (int)(u * fc.inv_g);
  const int q0 = *reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(lds) + fc.lut_off + 2 * cell);
  const double* rec0 = lds_at(lds + roff, __mul24(q0, stride * 8));
  const double k1 = rec0[2 * stride - 2], k2 = rec0[3 * stride - 2];      // left knots of the next two records
  const int q = q0 + (u >= k1) + (u >= k2);
  qi = q;
  const double* rec = lds_at(lds + roff, __mul24(q, stride * 8));
  const vk_d2 kw = *reinterpret_cast<const vk_d2*>(rec + stride - 2);
  tq = (u - kw.x) * kw.y;
  return rec;
}

// sigma_v(r, mu_r) / sigma_v of the 3-key anisotropic template (ccf_model.py:271-277, 654-655): the tensor-product
// not-a-knot spline of RectBivariateSpline as bicubic patches in LDS, both arguments clamped to the template's box as FITPACK's
// bispeu does (a negative mu_r reads mu = mu_0).  `rp` = r' in index units of the unified lattice, `mu2` = 2 mu_r.
__device__ __forceinline__ double sv_aniso(const double* __restrict__ lds, const FastConsts& fc, double rp, double mu2) {
  const double ts = vmin_f64(vmax_f64(fma(rp, fc.sv_a, fc.sv_b), 0.0), fc.sv_tmax);
  const double du = __builtin_amdgcn_fract(ts);
  const int i = (int)ts;
  const double m = vmin_f64(vmax_f64(0.5 * mu2, fc.mu_lo), fc.mu_hi);
  const double* muk = lds_at(lds, fc.svmu_off);
  int j;
  if (fc.mu_inv_h > 0.0) {
    j = min((int)((m - fc.mu_lo) * fc.mu_inv_h), fc.sv_nm - 1);
  } else {
    j = 0;
    for (int k = 1; k < fc.sv_nm; ++k) j += (m >= muk[k]) ? 1 : 0;
  }
  const double dm = m - muk[j];
  const double* c = lds_at(lds, fc.sva_off + (i * fc.sv_nm + j) * 128);
  double acc = cubic_b128(c + 12, dm);
  acc = fma(acc, du, cubic_b128(c + 8, dm));
  acc = fma(acc, du, cubic_b128(c + 4, dm));
  return fma(acc, du, cubic_b128(c, dm));
}

// One integrand point of the streaming model (ccf_model.py:648-657, 681-690): returns p = (1 + xi^r) exp(-z^2/2) and
// 1/SV in `inv_sv`; the caller accumulates inv_sv * p (times the velocity node's weight, taken per weight group where the
// node loop is wave-uniform).  HALF units (see FastPoint): `num` = r_par'/2, `sperp2` = s_perp'^2/4, `AVh` = AVk/2 (or
// unused with PV), `xk` the scaled velocity node.
//   geometry   X = num^2 + sperp2, yy = 4/r' (four instructions), mu2 = num yy = 2 mu_r, t = X yy + off
//   records    sigma_v and V cubics at (interval, tq); xi^r with the constant "+1" of 1 + xi^r already in its constant
//              coefficient, and for NLR > 1 the Legendre sum regrouped in powers of mu2^2 = 4 mu_r^2 (A + 1, B/4, C/16 of
//              vk_tables.uni_xic), so xi^r + 1 = A' + mu2^2 (B' + mu2^2 C')
//   Gaussian   1/SV from one Newton step (vkm::recip_nr), exp from vkm::exp_gauss<EXPT>
// FD = 1: realspace_ccf_from_data (ccf_model.py:618-619, 675-679) - xi^r is read at the fiducial coordinates
// (r_par / apar, s_perp / aperp) on an abscissa that is not rescaled by c, i.e. at num * fa and sperp2 * fp^2 with
// fa = c/apar, fp = c/aperp (`sperp2x` carries the second product), through a second interval look-up.
// PV = 1: the V cubics in the records already carry the per-point factor AVh (the kernels that own a point per workgroup
// rescale them once per work item, scale_uni_v) - one multiply less per integrand point.
// SVA = 1: sigma_v from the anisotropic template's bicubic patches (sv_aniso) instead of the record's cubic; lattice form only.
template <int NLR, int GRID, int FD, int PV = 0, int CL = 1, int EXPT = 0, int SVA = 0>
__device__ __forceinline__ double uni_point(const double* __restrict__ lds, const FastConsts& fc, double AVh,
                                            double num, double sperp2, double xk, double fa, double sperp2x,
                                            unsigned lane_off, double& inv_sv) {
  const double X = fma(num, num, sperp2);
  const double yy = vkm::rsqrt_nr_x2(X);
  const double mu2 = num * yy;
  double mu_x = mu2;                            // (twice) the mu at which xi^r is read
  double tq;
  int qi;
  const double rp = SVA ? X * yy : 0.0;
  const double* rec = locate<NLR, GRID, CL, EXPT>(lds, fc, GRID == 0 ? (SVA ? rp + fc.off : fma(X, yy, fc.off)) : X * yy, tq, qi);
  const double SV = SVA ? sv_aniso(lds, fc, rp, mu2) : cubic_b128(rec, tq);
  const double V = cubic_b128(rec + 4, tq);
  const double ynum = PV ? fma(V, mu2, xk) : fma(AVh * V, mu2, xk);
  if (FD) {
    const double rp = num * fa;
    const double Xx = fma(rp, rp, sperp2x);
    const double yyx = vkm::rsqrt_nr_x2(Xx);
    mu_x = rp * yyx;
    rec = locate<NLR, GRID, 1, EXPT>(lds, fc, GRID == 0 ? fma(Xx, yyx, fc.off) : Xx * yyx, tq, qi);
  }
  double xi1 = cubic_b128(rec + 8, tq);
  if (NLR > 1) {
    const double m2 = mu_x * mu_x;
    if (NLR == 2) {
      xi1 = fma(cubic_b128(rec + 12, tq), m2, xi1);
    } else {
      xi1 = fma(fma(cubic_b128(rec + 16, tq), m2, cubic_b128(rec + 12, tq)), m2, xi1);
    }
  }
  inv_sv = vkm::recip_nr(SV);
  const double e = vkm::exp_gauss<EXPT>(ynum, inv_sv, lds + kEtabOff, lane_off);
  return e * xi1;
}

// The dispersion model (ccf_model.py:658-671) on the same records: zero-mean Gaussian pdf of width sigma_v SV(r), the
// real-space coordinate from the reference's fixed-point iteration r_par <- (s_par - v/aH) / (1 + q(r)),
// q(r) = aH^-1 v_r(r)/r, started at the redshift-space separation and repeated `niter` more times, and the Jacobian
// 1 / (1 + q + mu_r^2 (dq - q)) with dq = aH^-1 v_r'(r).  `da` = LDS table of Da = delta - 2 Delta/3 on the unified
// grid, [uni_n][4].  All lengths in index units, HALF units as in uni_point (fc and fp made with half = true): `num` = r_par'/2
// before the iteration, `sperp2` = s_perp'^2 / 4, X = r'^2 / 4, yy = 4 / r' (vkm::rsqrt_nr_x2 in the early passes, rsqrt3_x2 -
// third order - in the last pass and behind it), X yy = r', r_par yy = 2 mu_r; r_par <- num / (1 + q) keeps its halved scale.
// The V cubics of the records carry the point's factor -Gk / 4 (rebuild_point_tables with vs = -fp.Gk, once per work item):
// q = V yy is one multiply, 1 + q one fma - a multiply less in every pass and in the evaluation behind them.
// The LAST pass of the iteration, the evaluation behind it and the Jacobian keep the third-order 1/sqrt and reciprocals (where
// 1 + q comes close to zero the result is ill-conditioned in them); the earlier passes run on the second-order forms - the map
// is a contraction wherever the reference itself converges, so what they leave in the last bits is damped by the passes that
// follow.  Same-box A/B (tools/gpu_rsd_ab.py, BOSS, 16384 points): 17.06 -> 16.31 ms, max rel dchi2 between the builds 4.9e-13
// (budget 1e-10), tests/test_gpu_options.py::test_dispersion_model_where_it_is_ill_conditioned unchanged and green
// (profiles/r04/c_early_passes_second_order_ab.txt).
// The FIRST pass starts from the redshift-space separation - r^2 = s_par^2 + s_perp^2, the same for every velocity node of an
// (s, mu) cell: 1 / (1 + q(s)) does not depend on v.  The kernels whose node loop is the inner loop (cells) take it once per cell
// and hand it to disp_value (`inv_den0`): five table look-ups per integrand point instead of six, the same arithmetic.
// It is an early pass (second-order forms) unless it is the only one (niter = 0).
template <int NLR, int GRID>
__device__ __forceinline__ double disp_first_pass(const double* __restrict__ lds, const FastConsts& fc, const FastPoint& fp,
                                                  int niter, double s_par, double sperp2) {
  double tq;
  int qi;
  const bool last = niter == 0;
  const double X = fma(s_par, s_par, sperp2);
  const double yy = last ? vkm::rsqrt3_x2(X) : vkm::rsqrt_nr_x2(X);
  const double* rec = locate<NLR, GRID>(lds, fc, GRID == 0 ? fma(X, yy, fc.off) : X * yy, tq, qi);
  const double den = fma(cubic_b128(rec + 4, tq), yy, 1.0);          // the records hold -Gk V / 4 (see disp_value)
  return last ? vkm::recip(den) : vkm::recip_nr(den);
}

// SVA = 1: sigma_v from the anisotropic template's bicubic patches (sv_aniso) as in uni_point; lattice form only.
template <int NLR, int GRID, int FD, int SVA = 0>
__device__ __forceinline__ double disp_value(const double* __restrict__ lds, const double* __restrict__ da,
                                             const FastConsts& fc, const FastPoint& fp, int niter, double num,
                                             double inv_den0, double sperp2, double xk) {
  double tq;
  int qi;
  auto pass = [&](double rp, bool last) {
    const double X = fma(rp, rp, sperp2);
    const double yy = last ? vkm::rsqrt3_x2(X) : vkm::rsqrt_nr_x2(X);
    const double* rec = locate<NLR, GRID>(lds, fc, GRID == 0 ? fma(X, yy, fc.off) : X * yy, tq, qi);
    const double den = fma(cubic_b128(rec + 4, tq), yy, 1.0);
    return num * (last ? vkm::recip(den) : vkm::recip_nr(den));
  };
  double r_par = num * inv_den0;                 // the first pass (disp_first_pass)
  // the trip count is a kernel argument: kept in a scalar register (the compiler otherwise counts the passes in a vector
  // register - two vector instructions per pass for a wave-uniform loop)
  for (int it = __builtin_amdgcn_readfirstlane(niter) - 1; it > 0; --it) r_par = pass(r_par, false);
  if (niter >= 1) r_par = pass(r_par, true);
  const double X = fma(r_par, r_par, sperp2);
  const double yy = vkm::rsqrt3_x2(X);           // 4 / r'
  const double mu2 = r_par * yy;                 // 2 mu_r
  const double rp = SVA ? X * yy : 0.0;
  const double* rec = locate<NLR, GRID>(lds, fc, GRID == 0 ? (SVA ? rp + fc.off : fma(X, yy, fc.off)) : X * yy, tq, qi);
  const double SV = SVA ? sv_aniso(lds, fc, rp, mu2) : cubic_b128(rec, tq);
  const double q = cubic_b128(rec + 4, tq) * yy;
  const double dq = -fp.gD * cubic_b128(da + 4 * qi, tq);
  double mx2 = mu2 * mu2;                        // the records hold the coefficients of powers of (2 mu)^2, see uni_point
  const double m2 = 0.25 * mx2;
  if (FD) {   // xi^r at the fiducial coordinates, as in uni_point
    const double rpx = r_par * fp.fa;
    const double Xx = fma(rpx, rpx, sperp2 * fp.fp2);
    const double yyx = vkm::rsqrt3_x2(Xx);
    const double mu_x = rpx * yyx;
    mx2 = mu_x * mu_x;
    rec = locate<NLR, GRID>(lds, fc, GRID == 0 ? fma(Xx, yyx, fc.off) : Xx * yyx, tq, qi);
  }
  double xi1 = cubic_b128(rec + 8, tq);          // 1 + xi^r_0 (the "+1" sits in the constant coefficient)
  if (NLR == 2) xi1 = fma(cubic_b128(rec + 12, tq), mx2, xi1);
  if (NLR == 3) xi1 = fma(fma(cubic_b128(rec + 16, tq), mx2, cubic_b128(rec + 12, tq)), mx2, xi1);
  const double inv_sv = vkm::recip(SV);
  const double jac = vkm::recip(1.0 + q + m2 * (dq - q));
  const double e = vkm::exp_gauss<0>(xk, inv_sv, lds + kEtabOff);
  return inv_sv * jac * (e * xi1);
}

// The models without a velocity integral - kaiser (ccf_model.py:692-741) and euclid_special (:743-784) - on the same records:
// ONE evaluation per (s, mu) cell.  With q(r) = aH^-1 v_r(r)/r = -Gk V(u)/r' and dq(r) = aH^-1 v_r'(r) = -gD Da(u) the
// reference's expressions read
//   coordinate shift (kaiser_coord_shift, `niter` + 1 passes, started at the redshift-space separation)
//       r_par <- s_par / (1 + M q(r))
//   kaiser          J = M q + M Q mu_r^2 (dq - q);   xi^s + 1 = (1 + M xi^r) / (1 + J),  or linearised 1 + M xi^r - J
//   euclid_special  J = 3 M q + 2 M Q mu_r^2 (dq - q);   xi^s + 1 = 1 + M xi^r - J
// Lengths in index units, FULL units (fc and fp made with half = false), as in disp_value; `da`: the LDS table of Da (or of
// Ge1 + av Ge2 with empirical_corr).  from_data (`fd`, wave-uniform): xi^r at the fiducial coordinates as in uni_point.
// Returns xi^s + 1 of the cell (the projection subtracts sum_i W_l[i] like everywhere else).
template <int NLR, int GRID>
__device__ __forceinline__ double kaiser_value(const double* __restrict__ lds, const double* __restrict__ da, const FastConsts& fc,
                                               const FastPoint& fp, double M, double Q, int niter, bool coord_shift, bool linear,
                                               bool euclid, bool fd, double s_par, double sperp2) {
  double tq;
  int qi;
  // One pass of the coordinate shift.  The map r_par -> s_par / (1 + M q(r)) is a contraction (|M r dq/dr| << 1 wherever the
  // reference itself converges), so what an early pass leaves behind in the last bits is damped by every later one: all passes
  // but the last run on the second-order 1/sqrt and reciprocal (two instructions less each, vk_devmath.h), the last one and the
  // evaluation behind it on the third-order forms (same-box A/B: 0.4585 -> 0.442 ms per 16384 BOSS points, max rel dchi2
  // between the builds 3.3e-15; profiles/r04/c_early_passes_second_order_ab.txt).
  const double MG = -M * fp.Gk;
  auto pass = [&](double rp, bool last) {
    const double r2 = fma(rp, rp, sperp2);
    const double inv_r = last ? vkm::rsqrt3(r2) : vkm::rsqrt_nr(r2);
    const double* rec = locate<NLR, GRID>(lds, fc, GRID == 0 ? fma(r2, inv_r, fc.off) : r2 * inv_r, tq, qi);
    const double den = fma(cubic_b128(rec + 4, tq) * inv_r, MG, 1.0);
    return s_par * (last ? vkm::recip(den) : vkm::recip_nr(den));
  };
  double r_par = s_par;
  if (coord_shift) {
    for (int it = 0; it < niter; ++it) r_par = pass(r_par, false);
    r_par = pass(r_par, true);
  }
  const double r2 = fma(r_par, r_par, sperp2);
  const double inv_r = vkm::rsqrt3(r2);
  const double mu_r = r_par * inv_r;
  const double* rec = locate<NLR, GRID>(lds, fc, GRID == 0 ? fma(r2, inv_r, fc.off) : r2 * inv_r, tq, qi);
  const double q = -fp.Gk * cubic_b128(rec + 4, tq) * inv_r;
  const double dq = -fp.gD * cubic_b128(da + 4 * qi, tq);
  const double m2 = mu_r * mu_r;
  double mx2 = 4.0 * m2;                        // the records hold the coefficients of powers of (2 mu)^2, see uni_point
  if (fd) {
    const double rp = r_par * fp.fa;
    const double r2x = fma(rp, rp, sperp2 * fp.fp2);
    const double inv_rx = vkm::rsqrt3(r2x);
    const double mu_x = rp * inv_rx;
    mx2 = 4.0 * (mu_x * mu_x);
    rec = locate<NLR, GRID>(lds, fc, GRID == 0 ? fma(r2x, inv_rx, fc.off) : r2x * inv_rx, tq, qi);
  }
  double xi1 = cubic_b128(rec + 8, tq);          // 1 + xi^r_0 (the "+1" sits in the constant coefficient)
  if (NLR == 2) xi1 = fma(cubic_b128(rec + 12, tq), mx2, xi1);
  if (NLR == 3) xi1 = fma(fma(cubic_b128(rec + 16, tq), mx2, cubic_b128(rec + 12, tq)), mx2, xi1);
  const double mxi = M * (xi1 - 1.0);             // M xi^r
  const double J = euclid ? 3.0 * M * q + 2.0 * M * Q * m2 * (dq - q) : M * q + M * Q * m2 * (dq - q);
  if (linear || euclid) return 1.0 + (mxi - J);
  return (1.0 + mxi) * vkm::recip(1.0 + J);
}

// MODE of the kernels that own a point per workgroup: streaming, streaming on a measured real-space ccf, dispersion (both),
// and the models without a velocity integral (cells kernel only; kaiser / euclid_special, from_data or not: wave-uniform flags)
constexpr int kModeStreaming = 0, kModeFromData = 1, kModeDispersion = 2, kModeDispersionFromData = 3, kModeKaiser = 4;
__host__ __device__ constexpr bool mode_is_dispersion(int mode) { return mode == kModeDispersion || mode == kModeDispersionFromData; }
// modes that read the Da table (v_r') and work in full index units with the unscaled V cubics
__host__ __device__ constexpr bool mode_has_da(int mode) { return mode >= kModeDispersion; }
// LDS layout of a mode (make_fast_plan / make_cells_plan `with_da`): 0 V1 copy, 1 Da table, 2 both
__host__ __device__ constexpr int mode_layout(int mode) { return mode_is_dispersion(mode) ? 2 : (mode_has_da(mode) ? 1 : 0); }

template <int NLR>
__device__ __forceinline__ void stage_da(const TheoryArgs& a, double* da) {
  if (!a.uni_da) return;                                     // beta-dependent velocity profile: rebuilt per point (rebuild_uni_xi)
  for (int e = threadIdx.x; e < a.uni_n * 4; e += kBlock) da[e] = a.uni_da[e];
}

// empirical_corr in the dispersion model: v_r' comes from the numerical-gradient tables, Dq = Ge1 + av Ge2
// (ccf_model.py:455-459), rebuilt per point like V = V1 + av V2
__device__ __forceinline__ void rebuild_da_emp(const TheoryArgs& a, double* da, double av) {
  const int n4 = a.uni_n * 4;
  for (int e = threadIdx.x; e < n4; e += kBlock) da[e] = fma(av, a.uni_ge[n4 + e], a.uni_ge[e]);
}

// --------------------------------------------------------------------------------------------------
// Fused tail of the kernels that own whole points or parts of points (point-major, cells): once a point's theory vector
// is complete - in this workgroup's LDS, or in the global workspace after the last of the workgroups sharing the point
// has finished (point_completed) - the same workgroup takes the chi-square and the log-likelihood
// (like_point_workgroup), so a batch needs ONE launch and the theory vector makes no round trip through HBM before it
// is used.  `th`: LDS, like_lds_doubles(N): the theory vector / residual TWICE over (2 M doubles, M = N rounded up to even: the
// quadratic form reads r[(i + k) mod M] as r2[i + k], vk_kernel_like.h), kLikeRed of reduction scratch, the completion flag.
// --------------------------------------------------------------------------------------------------
// (like_red_off, like_lds_doubles: vk_kernel_like.h)

// partial projections of a split plane: [point][l][s bin][kMaxParts], the parts of one (l, s bin) adjacent (64 bytes)
__device__ __forceinline__ double* partial_slot(double* partial, int n_s, long long point, int l, int j) {
  return partial + (((point * kMaxEll + l) * n_s + j) * (long long)kMaxParts);
}
__device__ __forceinline__ double* partial_slot(const TheoryArgs& a, long long point, int l, int j) {
  return partial_slot(a.partial, a.n_s, point, l, j);
}

template <int NL, int RB = kLikeRows>
__device__ __forceinline__ void finish_point(const TheoryArgs& a, long long point, double beta, double poison, double* th,
                                             bool gather_partials, const double* lds_beta_r, bool poll = false) {
  const int N = a.n_ell * a.n_s;
  double* red = th + like_red_off(N);
  const double w0 = a.wsum[0], w1 = a.wsum[1], w2 = a.wsum[2];
  LikePrefetch<RB> pf;
  if (a.fuse) pf.issue(a.like, beta, late_tid(), lds_beta_r);   // everything the chi-square needs besides the theory vector travels with the gather
  for (int e = threadIdx.x; e < N; e += kBlock) {
    double v;
    if (gather_partials) {
      const int l = (e >= 2 * a.n_s) ? 2 : (e >= a.n_s ? 1 : 0), j = e - l * a.n_s;
      double part[8];
      double* slot = partial_slot(a, point, l, j);
      load_shared_x8(slot, part);
      if (poll) {                                      // hand-off by polling (vk_common.h: kPollEmpty)
        long long t0 = 0;
        for (unsigned it = 1;; ++it) {
          bool empty = false;
#pragma unroll
          for (int q = 0; q < kMaxParts; ++q) empty = empty || (q < a.parts && poll_is_empty(part[q]));
          if (!empty) break;
          if ((it & 63u) == 0) {                       // the clock is looked at every 64th round trip (~50 us) only
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (now - t0 > kPollTicks) {               // never in a sound launch: fail the call, loudly
              *a.poll_failed = 1;
              __threadfence_system();
              break;
            }
          }
          load_shared_x8(slot, part);
        }
        const double empty_v = __longlong_as_double((long long)kPollEmpty);
#pragma unroll
        for (int q = 0; q < kMaxParts; ++q)
          if (q < a.parts) store_shared(slot + q, empty_v);       // the area is left as it was found, for the next launch
      }
      v = 0.0;
#pragma unroll
      for (int q = 0; q < kMaxParts; ++q) v += (q < a.parts) ? part[q] : 0.0;   // fixed order: independent of which part finished last
      v = v - (l == 0 ? w0 : (l == 1 ? w1 : w2)) + poison;
      a.out[point * (long long)N + e] = v;
    } else {
      v = load_shared(a.out + point * (long long)N + e);
    }
    th[e] = v;
  }
  __syncthreads();
  VK_STAMP(a, 6);
  if (a.fuse) like_point_workgroup(a.like, point, beta, th, red, pf);
}

// --------------------------------------------------------------------------------------------------
// K1 point-major fast kernel: one wave owns one (point, s bin) - or a share of it; lanes sweep the flattened (mu, v) plane.
// Work items are (point, group of s bins, part of the plane): `sbins_per_item` s bins per workgroup visit, `team` waves
// per s bin, `parts` workgroups per plane - a single point spreads over n_s * parts workgroups (the reference's calling
// convention is one point per call, CCFLikelihood.py:32-39).
// --------------------------------------------------------------------------------------------------
// Every kernel starts by filling LDS with tables that are the same for the whole batch - and for every launch of a
// context.  Staging them entry by entry costs a dozen dependent round trips to L2 per workgroup (~10 us, most of a
// single-point launch), so the host keeps, per kernel variant, an IMAGE of that part of LDS in global memory (made once by
// vk_image_kernel running the very same staging code) and the kernels copy it with all their loads in flight at once.
// Plans put the batch-constant regions first: LDS doubles [0, image_end) come from the image.
__device__ __forceinline__ void copy_image(double* lds, const double* __restrict__ image, int n_doubles) {
  const vk_d2* src = reinterpret_cast<const vk_d2*>(image);
  vk_d2* dst = reinterpret_cast<vk_d2*>(lds);
  const int n2 = n_doubles >> 1;                       // image_end is even
  for (int base = threadIdx.x; base < n2; base += 8 * kBlock) {
    vk_d2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + k * kBlock;
      v[k] = src[i < n2 ? i : base];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = base + k * kBlock;
      if (i < n2) dst[i] = v[k];
    }
  }
}

struct FastPlan {
  int murec, xrec, betar, da, v1, sva, image_end, red, like, total;
};

// n_sva: doubles of the anisotropic sigma_v block (TheoryArgs::sva_doubles) for the SVA instantiations, else 0
// with_da: LDS layout of the mode (mode_layout): 0 streaming (V1 copy), 1 kaiser / euclid_special (Da table), 2 dispersion (both)
__host__ __device__ inline FastPlan make_fast_plan(int n_mu, int n_x, int uni_n, int nlr, int n_beta_r, int lut_n,
                                                   int with_da, int n_like, int n_sva = 0) {
  FastPlan p;
  int o = fast_fixed_doubles(uni_n, nlr, lut_n);   // exp table + records first (fixed offsets)
  o = (o + 1) & ~1;
  p.murec = o; o += n_mu * kMuRec;
  p.xrec = o;  o += n_x * 2;
  p.betar = o; o += (n_beta_r + 1) & ~1;
  p.da = o;    o += with_da ? uni_n * 4 : 0;  // Da table of the dispersion model
  p.v1 = o;    o += with_da == 1 ? 0 : uni_n * 4;  // unscaled V1 cubics (the records hold AVk * V - streaming - or -Gk * V - dispersion, see scale_uni_v)
  o = (o + 1) & ~1;
  p.sva = o;   o += (n_sva + 1) & ~1;         // anisotropic sigma_v patches + mu knots
  p.image_end = o;                            // everything up to here is batch-constant (or rebuilt per point)
  p.red = o;   o += kWaves * kMaxEll;
  o = (o + 1) & ~1;
  p.like = o;  o += n_like > 0 ? like_lds_doubles(n_like) : 0;   // theory vector + reduction scratch of the fused tail
  p.total = o;
  return p;
}

// batch-constant LDS contents of the point-major kernel (entry by entry: the image builder and the general-grid calls)
template <int NLR>
__device__ __forceinline__ void stage_fast(const TheoryArgs& a, const FastPlan& pl, double* lds, int with_da) {
  const int tid = threadIdx.x;
  if (a.stage_mu) {
    for (int e = tid; e < a.n_mu * kMuRec; e += kBlock) lds[pl.murec + e] = a.stage_mu[e];
  } else {
    for (int i = tid; i < a.n_mu; i += kBlock) {
      const double m = a.mu[i];
      double* rec = lds + pl.murec + i * kMuRec;
      rec[0] = m;
      rec[1] = sqrt(1.0 - m * m);
#pragma unroll
      for (int l = 0; l < kMaxEll; ++l) rec[2 + l] = (l < a.n_ell) ? a.w_ell[l * a.n_mu + i] : 0.0;
      rec[5] = 0.0;
    }
  }
  for (int e = tid; e < 2 * a.n_x; e += kBlock) lds[pl.xrec + e] = a.xw_scaled[e];     // {kExpScale x_k, w_k}
  stage_uni_records<NLR>(a, lds);
  if (with_da) stage_da<NLR>(a, lds + pl.da);
  if (with_da != 1) for (int e = tid; e < a.uni_n * 4; e += kBlock) lds[pl.v1 + e] = a.uni_sv_v[(e >> 2) * 8 + 4 + (e & 3)];
  if (a.n_beta_r > 0)
    for (int i = tid; i < a.n_beta_r; i += kBlock) lds[pl.betar + i] = a.beta_r[i];
  if (pl.image_end > pl.sva)
    for (int e = tid; e < a.sva_doubles; e += kBlock) lds[pl.sva + e] = a.sva[e];
}

template <int NLR, int NL, int GRID, int MODE, int SVA = 0>
__global__ __launch_bounds__(kBlock, MODE == kModeStreaming && !SVA ? 3 : 2) void vk_theory_fast_kernel(TheoryArgs a) {
  extern __shared__ double lds[];
  vkm::clamp_keeps_nan();
  warm_kernarg_lines<sizeof(TheoryArgs)>();
  const int N = a.n_ell * a.n_s;
  const int Q = a.parts;
  const bool tail = a.fuse || Q > 1;
  const FastPlan pl = make_fast_plan(a.n_mu, a.n_x, a.uni_n, NLR, a.n_beta_r, a.uni_lut_n, mode_layout(MODE), tail ? N : 0,
                                     SVA ? a.sva_doubles : 0);
  const int tid = threadIdx.x;
  VK_STAMP(a, 0);
  // The first work item's per-point scalars (parameter row from global memory, AP integral, reciprocals: a serial chain
  // of ~1.5 us) are started before the tables are staged so that the two latencies overlap.
  const int groups = (a.n_s + a.sbins_per_item - 1) / a.sbins_per_item;
  const unsigned items = (unsigned)a.n * (unsigned)(groups * Q);     // the host keeps n * groups * parts below 2^31
  unsigned item = blockIdx.x;
  long long point = item < items ? (item / (unsigned)Q) / (unsigned)groups : 0;
  PointScalars ps = point_scalars(a, param_row(a, point));
  // ---- batch-constant tables: from the context's LDS image when there is one ------------------------
  if (a.image) copy_image(lds, a.image, pl.image_end);
  else stage_fast<NLR>(a, pl, lds, mode_layout(MODE));
  constexpr bool kHalf = true;                            // streaming and dispersion modes: half units (FastPoint)
  const FastConsts fc = make_fast_consts<NLR>(a, kHalf, SVA ? pl.sva : 0);
  __syncthreads();
  VK_STAMP(a, 1);

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int team = a.team;
  const int nteams = kWaves / team;
  const int my_team = wave / team;
  const int my_rank = wave - my_team * team;
  const int plane = a.n_mu * a.n_x;
  const int step = 64 * team;
  const int rounds = (a.sbins_per_item + nteams - 1) / nteams;
  const double* murec = lds + pl.murec;
  const double* xrec = lds + pl.xrec;
  double* l_red = lds + pl.red;
  // byte offsets of a node's mu record and (x, w) record follow idx = i n_x + k incrementally: no node table, no division
  const unsigned x_wrap = (unsigned)a.n_x * 16u;
  const unsigned d_mu = (unsigned)(step / a.n_x) * (kMuRec * 8u), d_x = (unsigned)(step % a.n_x) * 16u;

  // ONE work item per workgroup, no grid-stride loop (the host launches `items` workgroups), as in the cells kernel: out of a
  // loop the compiler hoisted fifty registers' worth of per-item set-up and tail values to the top of the kernel (146 -> 90-102
  // registers without the tail's rows, no SGPR spills; a single point 13.6 -> 12.2 us, 23 points 36.5 -> 24.4 us).
  if (item < items) {
    const unsigned pg = item / (unsigned)Q;
    const int q = (int)(item - pg * (unsigned)Q);
    const int g = (int)(pg - (unsigned)point * (unsigned)groups);
    const double* row = param_row(a, point);
    const FastPoint fp = make_fast_point(ps, fc, kHalf);
    constexpr int PV = 1;      // the V cubics carry the point's factor: AVk (streaming) or -Gk (dispersion, see disp_value)
    if (PV || a.n_beta_r > 0 || a.empirical) {
      __syncthreads();  // previous item's readers are done with the per-point records
      rebuild_point_tables<NLR, PV>(a, lds, pl.betar, pl.v1, row[VK_P_BETA], ps.av, mode_is_dispersion(MODE) ? -fp.Gk : fp.AVk,
                                    mode_is_dispersion(MODE) ? lds + pl.da : nullptr);
      if (mode_is_dispersion(MODE) && a.empirical && !a.vr_beta_dep) rebuild_da_emp(a, lds + pl.da, ps.av);
      __syncthreads();
    }
    VK_STAMP(a, 2);
    const int lo = (int)((long long)plane * q / Q), hi = (int)((long long)plane * (q + 1) / Q);
    const unsigned idx0 = (unsigned)(lo + lane + 64 * my_rank);
    const unsigned i0 = __umulhi(idx0, a.nx_magic);
    for (int rd = 0; rd < rounds; ++rd) {
      const int jl = rd * nteams + my_team;
      const int j = g * a.sbins_per_item + jl;
      const bool valid = (jl < a.sbins_per_item) && (j < a.n_s);
      double acc[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[l] = 0.0;
      if (valid) {
        const double sj = a.s[j];
        const double s_aperp = sj * fp.k_perp;
        const double s_apar = sj * fp.k_par;
        const char* mu_bytes = reinterpret_cast<const char*>(murec);
        const char* x_bytes = reinterpret_cast<const char*>(xrec);
        unsigned moff = i0 * (kMuRec * 8u), xoff = (idx0 - i0 * (unsigned)a.n_x) * 16u;
        for (int idx = (int)idx0; idx < hi; idx += step) {
          const double* mr = reinterpret_cast<const double*>(mu_bytes + moff);
          const vk_d2 m01 = *reinterpret_cast<const vk_d2*>(mr);
          const vk_d2 xw = *reinterpret_cast<const vk_d2*>(x_bytes + xoff);
          const double s_perp = s_aperp * m01.y;
          const double sperp2 = s_perp * s_perp;
          const double s_par = s_apar * m01.x;
          const double num = fma(-xw.x, fp.Bk, s_par);
          double f;
          if (mode_is_dispersion(MODE)) {
            f = xw.y * disp_value<NLR, GRID, MODE == kModeDispersionFromData>(lds, lds + pl.da, fc, fp, a.niter, num,
                                                                              disp_first_pass<NLR, GRID>(lds, fc, fp, a.niter, s_par, sperp2), sperp2, xw.x);
          } else {
            double inv_sv;
            const double p = uni_point<NLR, GRID, MODE == kModeFromData, 1, 1, 0, SVA>(lds, fc, 0.0, num, sperp2, xw.x, fp.fa, sperp2 * fp.fp2, 0u, inv_sv);
            f = (xw.y * inv_sv) * p;
          }
          const vk_d2 w01 = *reinterpret_cast<const vk_d2*>(mr + 2);
          acc[0] = fma(w01.x, f, acc[0]);
          if (NL > 1) acc[1] = fma(w01.y, f, acc[1]);
          if (NL > 2) acc[2] = fma(mr[4], f, acc[2]);
          // next node of this lane: k += step mod n_x with at most one wrap, i += step / n_x (+ 1 on a wrap)
          const unsigned xn = xoff + d_x;
          xoff = min(xn, xn - x_wrap);                 // xn < x_wrap: the subtraction wraps to a huge value and min keeps xn
          moff += d_mu + (xoff != xn ? kMuRec * 8u : 0u);
        }
      }
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[l] = wave_sum(acc[l]);
      // this (point, s bin)'s share: final when the plane is not split over workgroups, else a partial for finish_point
      double* dst = Q > 1 ? partial_slot(a, point, 0, j) + q : a.out + point * (long long)N + j;
      const long long dst_stride = Q > 1 ? (long long)a.n_s * kMaxParts : a.n_s;
      if (team == 1) {
        if (valid && lane < NL) {
          double v = acc[0], ws = a.wsum[0];
#pragma unroll
          for (int l = 1; l < NL; ++l) {
            v = (lane == l) ? acc[l < NL ? l : 0] : v;
            ws = (lane == l) ? a.wsum[l] : ws;
          }
          const double r = Q > 1 ? v : v - ws + ps.poison;
          if (tail) store_shared(dst + lane * dst_stride, r); else dst[lane * dst_stride] = r;
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
          for (int r = 0; r < team; ++r) v += l_red[(wave + r) * kMaxEll + lane];
          double ws = a.wsum[0];
#pragma unroll
          for (int l = 1; l < NL; ++l) ws = (lane == l) ? a.wsum[l] : ws;
          const double r = Q > 1 ? v : v - ws + ps.poison;
          if (tail) store_shared(dst + lane * dst_stride, r); else dst[lane * dst_stride] = r;
        }
      }
    }
    VK_STAMP(a, 3);
    if (tail) {
      // Fused / split launches run ONE item per workgroup (the host sizes the grid so) and leave from here: nothing is
      // live after the tail, so its registers (the chi-square needs ~100) do not spill the state of the loop above.
      double* th = lds + pl.like;
      int* flag = reinterpret_cast<int*>(th + like_red_off(N) + kLikeRed + 2);
      // a point owned by this workgroup alone needs no counter (and batches beyond the counter array have none): its theory
      // vector is re-read from L2 once this workgroup's own write-through stores have landed
      bool last = true;
      if (groups * Q == 1) {
        drain_shared_stores();
        __syncthreads();
      } else if (a.poll) {
        // no counter: the workgroup of the point's last work item collects the partial sums as they appear (vk_common.h)
        last = g == groups - 1 && q == Q - 1;
      } else {
        last = point_completed(a.counters, point, (unsigned)(groups * Q), flag);
      }
      VK_STAMP(a, 4);
      if (last) {
        finish_point<NL>(a, point, row[VK_P_BETA], ps.poison, th, Q > 1, a.n_beta_r > 0 ? lds + pl.betar : nullptr, a.poll != 0);
        VK_STAMP(a, 5);
      }
      return;
    }
  }
}


}  // namespace vk
