// vk_kernel_fast.h: point-major fast theory kernel and the shared fast-path building blocks - part of libvictor_hip.so (see victor_hip.hip for the overview and DESIGN.md section 5).
#pragma once
#include "vk_common.h"

namespace vk {

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
  p.etab = o;  o += vkm::kExpTab;
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

// The same integrand for the kernels that loop over the velocity nodes with a wave-uniform x_k (lanes, cells):
// s_par and s_perp^2 of the cell are formed once outside the loop, the Simpson weight is applied by the caller.
// Returns (1 + xi^r) * exp(-z^2/2) / SV.
template <int NLR>
__device__ __forceinline__ double node_value(const double* __restrict__ svrec, const double* __restrict__ vxrec,
                                             const double* __restrict__ leadrec, const double* __restrict__ etab,
                                             const FastConsts& fc, double B, double inv_c, double AV, double s_par,
                                             double sperp2, double xk) {
  constexpr int vx_stride = 4 * (1 + NLR) + 2;
  const double r_par = fma(-xk, B, s_par);
  const double r2 = fma(r_par, r_par, sperp2);
  double r, inv_r;
  vkm::sqrt_rsqrt(r2, r, inv_r);
  const double mu_r = r_par * inv_r;
  const double u = r * inv_c;
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
  return inv_sv * fma(e, xir, e);
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
  for (int j = tid; j < vkm::kExpTab; j += kBlock) lds[pl.etab + j] = vkm::exp2_frac(j);
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


}  // namespace vk
