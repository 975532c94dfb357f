// vk_kernel_lanes.h: lanes-over-the-batch theory kernel - part of libvictor_hip.so (see victor_hip.hip for the overview and DESIGN.md section 5).
#pragma once
#include "vk_kernel_fast.h"

namespace vk {

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
  int smu, image_end, total;
};

// exp-table form of the lanes kernel (vk_devmath.h: ExpCfg).  With the anisotropic sum the kernel reads ten 16-byte record
// pieces per integrand point and the LDS array is ~80 % busy, a sixth of it bank conflicts of the random exp-table reads; the
// replicated table (EXPT 1) halves those conflicts (16.4 % -> 9.4 % of the LDS cycles, command-FIFO-full cycles 1.1e9 -> 0.4e9
// per launch) but costs two vector instructions, 14 KB of LDS and with it the fifth workgroup per CU: 25.8 ms against 24.7 ms
// for the plain table on config 3 (same box, profiles/r03/c_*) - the vector ALU, 89-90 % busy, is what bounds the launch.
// The plain table is the default for every NLR; -D'VK_LANES_EXPT(NLR)=((NLR)>=2)' rebuilds the A/B.
#ifndef VK_LANES_EXPT
#define VK_LANES_EXPT(NLR) 0
#endif
__host__ __device__ constexpr int lanes_expt(int nlr) { return VK_LANES_EXPT(nlr); }

__host__ __device__ inline LanesPlan make_lanes_plan(int n_mu, int n_x, int uni_n, int nlr, int lut_n) {
  LanesPlan p;
  int o = fast_fixed_doubles(uni_n, nlr, lut_n, lanes_expt(nlr));   // exp table + records first (fixed offsets)
  o = (o + 1) & ~1;
  p.smu = o;   o += 2 * n_mu;           // {mu_i, sqrt(1 - mu_i^2)}
  p.image_end = o;                      // all of it is batch-constant (LDS image, see vk_kernel_fast.h)
  p.total = o;
  return p;
}

template <int NLR>
__device__ __forceinline__ void stage_lanes(const TheoryArgs& a, const LanesPlan& pl, double* lds) {
  for (int i = threadIdx.x; i < a.n_mu; i += kBlock) {
    const double m = a.mu[i];
    lds[pl.smu + 2 * i] = m;
    lds[pl.smu + 2 * i + 1] = sqrt(1.0 - m * m);
  }
  stage_uni_records<NLR, lanes_expt(NLR)>(a, lds);
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
      double v, ir;
      vkm::sqrt_rsqrt(fma(1.0 - mm * mm, e2, 1.0), v, ir);
      acc += (m == 0 || m == 49) ? 0.5 * v : v;
    }
    c = ps.apar * acc * h;
  } else {
    c = row[VK_P_ASTAR];
  }
  ps.inv_c = vkm::recip(c);
  const double iaH_true = a.iaH * ps.apar;
  double extra = 0.0;
  const double gb = growth_amplitude(a, row, fs8, &ps.av, &extra);
  ps.B = sigv * iaH_true;
  ps.A = gb * vkm::recip(3.0 * iaH_true * sigv);
  ps.G = gb * (1.0 / 3.0);
  ps.gD = gb * ps.inv_c;
  ps.M = row[VK_P_M];
  ps.Q = row[VK_P_Q];
  ps.inv_aperp = vkm::recip(ps.aperp);
  ps.inv_apar = vkm::recip(ps.apar);
  ps.poison = 0.0 * (gb + sigv + ps.aperp + ps.apar + eps + c + ps.A + extra);
  return ps;
}

#ifndef VK_LANES_WG_PER_CU
#define VK_LANES_WG_PER_CU 5
#endif

template <int NLR, int NL, int GRID>
__global__ __launch_bounds__(kBlock, VK_LANES_WG_PER_CU) void vk_theory_lanes_kernel(TheoryArgs a) {
  extern __shared__ double lds[];
  vkm::clamp_keeps_nan();
  const LanesPlan pl = make_lanes_plan(a.n_mu, a.n_x, a.uni_n, NLR, a.uni_lut_n);
  const int tid = threadIdx.x;
  if (a.image) copy_image(lds, a.image, pl.image_end);
  else stage_lanes<NLR>(a, pl, lds);
  constexpr int EXPT = lanes_expt(NLR);
  const FastConsts fc = make_fast_consts<NLR, EXPT>(a, true);
  __syncthreads();

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const unsigned lane_off = (unsigned)(lane & 31) << 3;      // this lane's replica of the exp table (EXPT 1)
  const double* l_smu = lds + pl.smu;
  // velocity nodes through the scalar cache: wave-uniform, read-only for the whole launch (constant address space
  // tells the compiler so), which keeps them out of the VALU and LDS pipes
  const long long chunks = (a.n + 63) >> 6;
  const long long items = chunks * a.n_s;
  // XCD-aware block order: the n_s waves of a 64-point chunk read the same 6 KB of parameter rows; consecutive
  // workgroup ids round-robin over the 8 XCDs (each with a private L2), so give every XCD a contiguous range of
  // logical ids (bijective for any grid size) and a chunk's rows are fetched from HBM once, not eight times
  const unsigned nwg = gridDim.x, orig = blockIdx.x;
  const unsigned xq = nwg >> 3, xr = nwg & 7, xcd = orig & 7;
  const unsigned bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (orig >> 3);
  // Work items (chunk-major: item = chunk * n_s + s bin) are dealt to the workgroups in blocks of `per_block` consecutive
  // items, a wave taking every fourth one of its block: per_block = 4 is one item per wave (the default); per_block = n_s
  // gives a workgroup ALL s bins of one 64-point chunk - the mapping a chi-square fused into this kernel would need (its
  // theory vectors would then sit in one workgroup's LDS) - kept as an A/B knob: see DESIGN.md "Measured and rejected".
  const int per_block = a.lanes_per_block;
  const long long nblocks = (items + per_block - 1) / per_block;
  for (long long blk = bid; blk < nblocks; blk += gridDim.x)
  for (int sub = wave; sub < per_block; sub += kWaves) {
    const long long item = blk * per_block + sub;
    if (item >= items) break;
    const long long chunk = item / a.n_s;
    const int j = (int)(item - chunk * a.n_s);
    long long point = chunk * 64 + lane;
    const bool valid = point < a.n;
    if (!valid) point = a.n - 1;
    const PointScalars ps = point_scalars_lane(a, a.params + point * VK_NPAR);
    const FastPoint fp = make_fast_point(ps, fc, true);     // half units (vk_kernel_fast.h: FastPoint)
    const double sj = a.s[j];
    const double sa = sj * fp.k_perp, sp = sj * fp.k_par;
    const double xi_max = a.xw_max * fabs(fp.Bk);
    double acc[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) acc[l] = 0.0;
    for (int i = 0; i < a.n_mu; ++i) {
      const vk_d2 mm = *reinterpret_cast<const vk_d2*>(l_smu + 2 * i);
      const double s_perp = sa * mm.y;
      const double sperp2 = s_perp * s_perp;
      const double s_par = sp * mm.x;
      double g = 0.0;
      // rows whose 64 x 50 radii all fall inside the table (most of them: the records run to the last knot of the longest
      // table, and only the mu = 1 row reaches r < 0.01) skip the clamp pair of the interval coordinate.  Extending the
      // records past the last knot so that the top s bins qualify too was measured and dropped: nothing on config 3, and the
      // larger LDS footprint costs BOSS a workgroup per CU (profiles/r02/i_clamp_exp_ab.txt)
      const bool inside = !__any(!cell_in_table(fc, s_par, sperp2, xi_max));
      // nodes in groups of equal quadrature weight: `gs` sums a group, its weight arrives with the group's last node
      // (a wave-uniform test on a scalar register) - one multiply per group instead of one per integrand point
      double gs = 0.0;
#define VK_GROUP_END(xw)                                                                                                  \
  if ((xw).last != 0) {            /* wave-uniform: a scalar branch, taken at the last node of a group */   \
    asm volatile("" ::: "memory");              /* (keeps the compiler from turning it into per-lane selects) */        \
    g = fma((xw).w, gs, g);                                                                                               \
    gs = 0.0;                                                                                                             \
  }
      if (inside) {
        for (int k = 0; k < a.n_xg; ++k) {
          const VelocityNode xw = load_node(a.xgw, k);
          double inv_sv;
          const double p = uni_point<NLR, GRID, 0, 0, 0, EXPT>(lds, fc, fp.AVk, fma(-xw.x, fp.Bk, s_par), sperp2, xw.x, 0.0, 0.0, lane_off, inv_sv);
          gs = fma(inv_sv, p, gs);
          VK_GROUP_END(xw)
        }
      } else {
        for (int k = 0; k < a.n_xg; ++k) {
          const VelocityNode xw = load_node(a.xgw, k);
          double inv_sv;
          const double p = uni_point<NLR, GRID, 0, 0, 1, EXPT>(lds, fc, fp.AVk, fma(-xw.x, fp.Bk, s_par), sperp2, xw.x, 0.0, 0.0, lane_off, inv_sv);
          gs = fma(inv_sv, p, gs);
          VK_GROUP_END(xw)
        }
      }
#undef VK_GROUP_END
#pragma unroll
      for (int l = 0; l < NL; ++l) acc[l] = fma(a.w_ell[l * a.n_mu + i], g, acc[l]);
    }
    if (valid) {
      double* o = a.out + point * (long long)(a.n_ell * a.n_s) + j;
#pragma unroll
      for (int l = 0; l < NL; ++l) o[(long long)l * a.n_s] = acc[l] - a.wsum[l] + ps.poison;
    }
  }
}


}  // namespace vk
