// vk_kernel_cells.h: cells theory kernel (per-point tables, velocity loop innermost) - part of libvictor_hip.so (see victor_hip.hip for the overview and DESIGN.md section 5).
#pragma once
#include "vk_kernel_fast.h"

namespace vk {

// --------------------------------------------------------------------------------------------------
// K1 "cells" variant: the lanes kernel's inner loop for per-point tables (reconstruction beta, BOSS) and for batches too
// small for the lanes kernel.  A point's n_s * n_mu (s bin, mu) cells are flattened (s bin major) and cut into `parts`
// ranges of `cells_per_item` cells; one workgroup owns one range of one point, with the xi^r records rebuilt in LDS as in the
// point-major kernel.  The range is dealt to the four waves in trips of 64 cells (wave w takes trips w, w + 4, ...: every
// wave gets the same number of trips to within one, whatever n_s is), lanes over the cells, the 50 velocity nodes as the
// inner, wave-uniform loop.  Like the lanes kernel this forms s_perp and s_par once per cell, reads x_k, w_k through the
// scalar cache and closes the v sum before the projection, so the integrand costs the same instructions; the projection
// sum over mu is a two-segment wave reduction per trip (a trip's 64 cells straddle at most two s bins when n_mu >= 64),
// accumulated by lane 0 in wave-private LDS and combined over the waves at the end.
//   parts == 1: the finished theory vector sits in LDS and the chi-square is taken there (`fuse`);
//   parts  > 1: ranges need not respect s-bin boundaries - every (l, s bin) sum is handed over as one partial per
//               contributing range (partial_slot: at most ceil((n_mu - 1) / cells_per_item) + 1 of them) and the workgroup
//               that completes the point adds them in range order, so the result does not depend on who finishes last.
//               64 points x 4000 cells in ranges of 256 are 1024 workgroups of one trip per wave: this is what makes a
//               64-point batch fill the chip.
// --------------------------------------------------------------------------------------------------
struct CellsPlan {
  int mu, w, s, betar, da, v1, sva, image_end, acc, like, total;
};

// s bins a range of `cpi` cells can touch
__host__ __device__ inline int cells_range_bins(int n_mu, int cpi) { return (cpi + n_mu - 2) / n_mu + 1; }

__host__ __device__ inline CellsPlan make_cells_plan(int n_mu, int n_x, int n_s, int uni_n, int nlr, int n_beta_r,
                                                     int lut_n, int with_da, int cpi, int n_like, int n_sva = 0) {
  CellsPlan p;
  int o = fast_fixed_doubles(uni_n, nlr, lut_n);          // exp table + records first (fixed offsets)
  o = (o + 1) & ~1;
  p.mu = o;    o += 2 * n_mu;                      // {mu_i, sqrt(1 - mu_i^2)}
  p.w = o;     o += kMaxEll * n_mu;                // W_l[i]
  p.s = o;     o += (n_s + 1) & ~1;
  p.betar = o; o += (n_beta_r + 1) & ~1;
  p.da = o;    o += with_da ? uni_n * 4 : 0;      // Da table of the dispersion model
  p.v1 = o;    o += with_da == 1 ? 0 : uni_n * 4; // unscaled V1 cubics (streaming and dispersion modes, see scale_uni_v)
  o = (o + 1) & ~1;
  p.sva = o;   o += (n_sva + 1) & ~1;             // anisotropic sigma_v patches + mu knots (SVA instantiations)
  p.image_end = o;                                 // batch-constant up to here (LDS image, see vk_kernel_fast.h)
  int bins = cells_range_bins(n_mu, cpi);
  if (bins > n_s) bins = n_s;
  p.acc = o;   o += kMaxEll * (bins + 1) * kWaves;   // [l][local bin][wave] (+ one spill bin)
  o = (o + 1) & ~1;
  p.like = o;  o += n_like > 0 ? like_lds_doubles(n_like) : 0;
  p.total = o;
  return p;
}

template <int NLR>
__device__ __forceinline__ void stage_cells(const TheoryArgs& a, const CellsPlan& pl, double* lds, int with_da) {
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
  if (with_da != 1) for (int e = tid; e < a.uni_n * 4; e += kBlock) lds[pl.v1 + e] = a.uni_sv_v[(e >> 2) * 8 + 4 + (e & 3)];
  if (a.n_beta_r > 0)
    for (int i = tid; i < a.n_beta_r; i += kBlock) lds[pl.betar + i] = a.beta_r[i];
  if (pl.image_end > pl.sva)
    for (int e = tid; e < a.sva_doubles; e += kBlock) lds[pl.sva + e] = a.sva[e];
}

// The workgroup that completed a point whose cells were split into ranges: add every (l, s bin)'s partials in range order,
// publish the theory vector, take the chi-square.  `th`: LDS, see like_lds_doubles.
// `gather` false: the point was this workgroup's alone and its theory vector is in `th` already (written, not yet synchronised).
template <int NL>
__device__ __forceinline__ void finish_point_ranges(const TheoryArgs& a, long long point, double beta, double poison, double* th,
                                                    bool gather, const double* lds_beta_r) {
  const int N = a.n_ell * a.n_s;
  const double w0 = a.wsum[0], w1 = a.wsum[1], w2 = a.wsum[2];
  const bool keep_out = !a.fuse || a.want_theory;       // see TheoryArgs::want_theory
  LikePrefetch<kLikeRowsCells> pf;
  if (a.fuse) pf.issue(a.like, beta, late_tid(), lds_beta_r);   // travels with the gather (vk_kernel_like.h)
  for (int e = late_tid(); gather && e < N; e += kBlock) {
    const int l = (e >= 2 * a.n_s) ? 2 : (e >= a.n_s ? 1 : 0), j = e - l * a.n_s;
    const int q_first = (j * a.n_mu) / a.cells_per_item, q_last = (j * a.n_mu + a.n_mu - 1) / a.cells_per_item;
    double part[8];
    load_shared_x8(partial_slot(a, point, l, j), part);
    double v = 0.0;
#pragma unroll
    for (int c = 0; c < kMaxParts; ++c) v += (c <= q_last - q_first) ? part[c] : 0.0;
    v = v - (l == 0 ? w0 : (l == 1 ? w1 : w2)) + poison;
    if (keep_out) a.out[point * (long long)N + e] = v;
    th[e] = v;
  }
  __syncthreads();
  if (a.fuse) like_point_workgroup(a.like, point, beta, th, th + like_red_off(N), pf);
}

// Projection of one trip: sum_lanes W_l[i] g over the lanes of the trip's first s bin and over those of its second, for every
// l, added to the wave's accumulators (acc: this wave's entry of the first bin for l = 0; `lstride` doubles from one l to
// the next, kWaves from a bin to the next).  The 2 NL sums are folded together (fold32, fold16: two values per addition) down
// to rows of 16 lanes, the rows to octets with one exchange, the octets by DPP: 50 vector instructions for NL = 3 where six
// full-wave DPP chains took 150.  Lane -> sum it ends up with (NL = 3): bits 5, 4, 3 of the lane = (second bin, l & 1, l == 2).
template <int NL>
__device__ __forceinline__ void project_trip(double g, bool first, bool second, const double* w_i, int n_mu, double* acc, int lstride, int lane) {
  const double ga = first ? g : 0.0, gb = second ? g : 0.0;
  double z[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const double w = w_i[l * n_mu];
    z[l] = vkm::fold32(w * ga, w * gb);               // lanes 0-31: first bin, 32-63: second bin
  }
  double x;
  int l_mine;
  bool mine;
  const int row = lane >> 4;
  if (NL == 3) {
    const double u = vkm::fold16(z[0], z[1]);          // rows: (l 0, bin 0), (l 1, bin 0), (l 0, bin 1), (l 1, bin 1)
    const double v = vkm::fold16(z[2], 0.0);           // rows 0 and 2: l = 2; rows 1 and 3: zero
    const bool up = (lane & 8) != 0;                   // the upper octet of a row keeps v, the lower one u
    const double keep = up ? v : u, send = up ? u : v;
    x = keep + vkm::dpp_move<0x128, 0xF, true>(send);  // row_ror:8: from the lane eight away, which keeps the other one
    l_mine = up ? 2 : (row & 1);
    mine = (lane & 7) == 0 && !(up && (row & 1));
  } else {
    x = vkm::fold16(z[0], NL == 2 ? z[NL - 1] : 0.0);
    x += vkm::dpp_move<0x128, 0xF, true>(x);
    l_mine = row & 1;
    mine = (lane & 15) == 0 && (NL == 2 || !(row & 1));
  }
  x = vkm::octet_sum(x);
  if (mine) acc[l_mine * lstride + (lane >> 5) * kWaves] += x;
}

// 5 workgroups per CU for the streaming mode (<= 96 VGPRs); the from_data and dispersion modes need more registers and
// run 4 per CU without spills
template <int NLR, int NL, int GRID, int MODE, int SVA = 0>
__global__ __launch_bounds__(kBlock, (MODE == kModeStreaming && !SVA) || MODE == kModeKaiser ? 5 : 4) void vk_theory_cells_kernel(TheoryArgs a) {
  extern __shared__ double lds[];
  vkm::clamp_keeps_nan();
  warm_kernarg_lines<sizeof(TheoryArgs)>();
  const int N = a.n_ell * a.n_s;
  const int R = a.parts;
  const int cpi = a.cells_per_item;
  // theory_xi (TheoryArgs::xi_out; the n_ell = 1 instantiations only, wave-uniform): the cells are taken mu-major - the order of
  // out[n][n_mu][n_s], so a trip's 64 stores are consecutive - and stored as they are; ranges of a point need no hand-over
  const bool xi_out = NL == 1 && a.xi_out != 0;
  const bool tail = !xi_out && (a.fuse || R > 1);
  CellsPlan pl = make_cells_plan(a.n_mu, a.n_x, a.n_s, a.uni_n, NLR, a.n_beta_r, a.uni_lut_n, mode_layout(MODE), xi_out ? 0 : cpi,
                                 tail ? N : 0, SVA ? a.sva_doubles : 0);
  // the offset behind the accumulators depends on an integer division by n_mu, which the compiler evaluates on the vector ALU:
  // wave-uniform, but held - and once spilled - as a vector register unless it is made a scalar here
  pl.like = __builtin_amdgcn_readfirstlane(pl.like);
  const int tid = threadIdx.x;
  VK_STAMP(a, 0);
  if (a.image) copy_image(lds, a.image, pl.image_end);
  else stage_cells<NLR>(a, pl, lds, mode_layout(MODE));
  constexpr bool kHalf = MODE != kModeKaiser;             // streaming and dispersion modes: half units (vk_kernel_fast.h: FastPoint)
  const FastConsts fc = make_fast_consts<NLR>(a, kHalf, SVA ? pl.sva : 0);
  __syncthreads();
  VK_STAMP(a, 1);

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: a scalar register, not one VGPR per lane
  const double* l_mu = lds + pl.mu;
  const double* l_w = lds + pl.w;
  const double* l_s = lds + pl.s;
  const int slots = xi_out ? 1 : min(cells_range_bins(a.n_mu, cpi), a.n_s) + 1;   // local bins of a range (+ one that only ever receives zeros)
  double* l_acc = lds + pl.acc;                              // [l][local bin][wave]: each entry touched by one wave only
  const unsigned items = (unsigned)a.n * (unsigned)R;                // the host keeps n * parts below 2^31
  const int all_cells = a.n_s * a.n_mu;

  // ONE work item per workgroup, no grid-stride loop (the host launches `items` workgroups): around a loop the compiler hoists
  // what the per-point set-up and the tail derive from the thread index and the kernel arguments to the top of the kernel and
  // carries it through the node loops - seven values that cost the denser schedule of vk_cells_streaming.hip 32 bytes of scratch.
  {
    const unsigned item = blockIdx.x;
    if (item >= items) return;
    const long long point = item / (unsigned)R;
    const int q = (int)(item - (unsigned)point * (unsigned)R);
    const int c0 = q * cpi, c1 = min(c0 + cpi, all_cells);           // this item's cells
    const int jf = (int)__umulhi((unsigned)c0, a.nmu_magic);         // first s bin it touches
    const int nb = (int)__umulhi((unsigned)(c1 - 1), a.nmu_magic) - jf + 1;
    const double* row = a.params + point * VK_NPAR;
    const PointScalars ps = point_scalars(a, row);
    const FastPoint fp = make_fast_point(ps, fc, kHalf);
    // the V cubics carry the point's factor: AVk (streaming), -Gk (dispersion, see disp_value), none (kaiser / euclid_special)
    constexpr int PV = MODE == kModeKaiser ? 0 : 1;
    __syncthreads();      // every wave is done with the previous item's records and accumulators
    rebuild_point_tables<NLR, PV>(a, lds, pl.betar, pl.v1, row[VK_P_BETA], ps.av, mode_is_dispersion(MODE) ? -fp.Gk : fp.AVk,
                                  mode_has_da(MODE) ? lds + pl.da : nullptr);
    if (mode_has_da(MODE) && a.empirical && !a.vr_beta_dep) rebuild_da_emp(a, lds + pl.da, ps.av);
    for (int e = lane; e < kMaxEll * slots; e += 64) l_acc[e * kWaves + wave] = 0.0;
    __syncthreads();
    VK_STAMP(a, 2);
    for (int base = c0 + 64 * wave; base < c1; base += 64 * kWaves) {
      const int e = base + lane;
      const bool live = e < c1;
      const unsigned ec = (unsigned)(live ? e : c1 - 1);
      // ec / n_mu: s bin major (ec / n_s: mu major for xi_out; a single s bin has no 32-bit magic number - ceil(2^32 / 1) - and
      // needs none)
      const int hi = xi_out ? (a.n_s == 1 ? (int)ec : (int)__umulhi(ec, a.ns_magic)) : (int)__umulhi(ec, a.nmu_magic);
      const int lo = (int)ec - hi * (xi_out ? a.n_s : a.n_mu);
      const int j = xi_out ? lo : hi;
      const int i = xi_out ? hi : lo;
      const int jj = j - jf;
      const double sj = l_s[j];
      const vk_d2 mm = *reinterpret_cast<const vk_d2*>(l_mu + 2 * i);
      const double s_perp = sj * fp.k_perp * mm.y;
      const double sperp2 = s_perp * s_perp;
      const double sperp2x = sperp2 * fp.fp2;      // from_data only
      const double s_par = sj * fp.k_par * mm.x;
      double g = 0.0;
      // trips whose 64 x 50 radii all fall inside the table skip the clamp pair of the interval coordinate (see the lanes
      // kernel; a trip that holds a mu = 1 cell reaches r < 0.01 and keeps it)
      const bool inside = !mode_has_da(MODE) && !__any(!cell_in_table(fc, s_par, sperp2, a.xw_max * fabs(fp.Bk)));
      double gs = 0.0;
      if (MODE == kModeKaiser) {
        g = kaiser_value<NLR, GRID>(lds, lds + pl.da, fc, fp, ps.M, ps.Q, a.niter, a.coord_shift != 0, a.kaiser_approx != 0,
                                    a.rsd == VK_RSD_EUCLID, a.from_data != 0, s_par, sperp2);
      } else if (mode_is_dispersion(MODE)) {
        const double inv_den0 = disp_first_pass<NLR, GRID>(lds, fc, fp, a.niter, s_par, sperp2);     // once per cell, not per node
        for (int k = 0; k < a.n_xg; ++k) {
          const VelocityNode xw = load_node(a.xgw, k);
          gs += disp_value<NLR, GRID, MODE == kModeDispersionFromData, SVA>(lds, lds + pl.da, fc, fp, a.niter, fma(-xw.x, fp.Bk, s_par), inv_den0,
                                                                     sperp2, xw.x);
          if (xw.last != 0) {            // wave-uniform: a scalar branch, taken at the last node of a group
            asm volatile("" ::: "memory");            // (keeps the compiler from turning it into per-lane selects)
            g = fma(xw.w, gs, g);
            gs = 0.0;
          }
        }
      } else if (inside) {
        VelocityNode nxt = load_node(a.xgw, 0);
        for (int k = 0; k < a.n_xg; ++k) {
          const VelocityNode xw = nxt;
          nxt = load_node(a.xgw, k + 1);            // one node ahead (the table has n_xg + 1 entries)
          double inv_sv;
          const double p = uni_point<NLR, GRID, MODE == kModeFromData, 1, 0, 0, SVA>(lds, fc, 0.0, fma(-xw.x, fp.Bk, s_par), sperp2, xw.x, fp.fa,
                                                                            sperp2x, 0u, inv_sv);
          gs = fma(inv_sv, p, gs);
          if (xw.last != 0) {            // wave-uniform: a scalar branch, taken at the last node of a group
            asm volatile("" ::: "memory");            // (keeps the compiler from turning it into per-lane selects)
            g = fma(xw.w, gs, g);
            gs = 0.0;
          }
        }
      } else {
        VelocityNode nxt = load_node(a.xgw, 0);
        for (int k = 0; k < a.n_xg; ++k) {
          const VelocityNode xw = nxt;
          nxt = load_node(a.xgw, k + 1);            // one node ahead (the table has n_xg + 1 entries)
          double inv_sv;
          const double p = uni_point<NLR, GRID, MODE == kModeFromData, 1, 1, 0, SVA>(lds, fc, 0.0, fma(-xw.x, fp.Bk, s_par), sperp2, xw.x, fp.fa,
                                                                            sperp2x, 0u, inv_sv);
          gs = fma(inv_sv, p, gs);
          if (xw.last != 0) {            // wave-uniform: a scalar branch, taken at the last node of a group
            asm volatile("" ::: "memory");            // (keeps the compiler from turning it into per-lane selects)
            g = fma(xw.w, gs, g);
            gs = 0.0;
          }
        }
      }
      if (xi_out) {                                  // xi^s = sum - 1 (ccf_model.py:690), cell e = (i, j) of out[point][n_mu][n_s]
        if (live) a.out[point * (long long)all_cells + e] = g - 1.0 + ps.poison;
        continue;
      }
      // projection: this trip's cells belong to local bin jj0 or jj0 + 1 (the latter may be the spill bin)
      const int jj0 = __builtin_amdgcn_readfirstlane(jj);
      project_trip<NL>(g, live && jj == jj0, live && jj != jj0, l_w + i, a.n_mu, l_acc + jj0 * kWaves + wave, slots * kWaves, lane);
    }
    if (xi_out) return;
    // this range's share of the theory vector is complete in LDS once every wave has finished its trips
    __syncthreads();
    double* th = lds + pl.like;
    for (int e = late_tid(); e < NL * nb; e += kBlock) {
      const int l = e / nb, jl = e - l * nb;
      const int j = jf + jl;
      const double* pa = l_acc + (l * slots + jl) * kWaves;
      const double sum = (pa[0] + pa[1]) + (pa[2] + pa[3]);
      if (R > 1) {
        store_shared(partial_slot(a, point, l, j) + (q - (j * a.n_mu) / cpi), sum);
      } else {
        const double v = sum - (l == 0 ? a.wsum[0] : (l == 1 ? a.wsum[1] : a.wsum[2])) + ps.poison;
        if (!a.fuse || a.want_theory) a.out[point * (long long)N + l * a.n_s + j] = v;   // see TheoryArgs::want_theory
        if (tail) th[l * a.n_s + j] = v;
      }
    }
    VK_STAMP(a, 3);
    if (tail) {
      // fused / split launches run one item per workgroup and leave from here (see vk_kernel_fast.h)
      bool last = true;
      if (R > 1) {
        int* flag = reinterpret_cast<int*>(th + like_red_off(N) + kLikeRed + 2);
        last = point_completed(a.counters, point, (unsigned)R, flag);
      }
      VK_STAMP(a, 4);
      if (last) {
        finish_point_ranges<NL>(a, point, row[VK_P_BETA], ps.poison, th, R > 1, a.n_beta_r > 0 ? lds + pl.betar : nullptr);
        VK_STAMP(a, 5);
      }
      return;
    }
  }
}


}  // namespace vk
