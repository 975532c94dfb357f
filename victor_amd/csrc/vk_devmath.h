// vk_devmath.h - FP64 building blocks for the gfx950 kernels.
//
// The hot loop needs, per integrand point, one sqrt together with the matching 1/r, one reciprocal and
// one exp of a non-positive argument.  The IEEE-correct library forms cost ~8 + 12 + 12 + 20 vector
// instructions; the forms below cost ~9 + 5 + 12 and stay within 2 ulp (measured on hardware by
// tools/devmath_check.hip, asserted in tests/test_gpu_devmath.py), far inside the 1e-6 relative parity
// budget of the likelihood (the kernels are tested to 1e-9).
#pragma once

#include <hip/hip_runtime.h>

namespace vkm {

// Sum over the 64 lanes of a wavefront, returned to every lane (wave-uniform: the compiler may keep it in SGPRs).
// Cross-lane traffic goes through DPP (quad permutes, row mirrors, row broadcasts), not through ds_bpermute: the LDS pipe
// is the second-busiest unit of the theory kernels and a reduction there costs 12 LDS instructions per sum.
// Fixed association order: pairs, quads, half rows, rows of 16, then rows 0+1, 2+3, and the two halves.
// BOUND_CTRL: lanes without a source read 0 instead of keeping the old value of the destination.  The permutes within a
// row give every lane a source, so nothing is ever kept - but only with bound_ctrl set may the compiler drop the two
// v_mov_b32 that materialise the "old" operand in front of every pair of DPP moves (32 of them per wave_sum pair of a trip).
template <int CTRL, int ROW_MASK, bool BOUND_CTRL = false>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, BOUND_CTRL);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, BOUND_CTRL);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0xB1, 0xF, true>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xF, true>(v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141, 0xF, true>(v);   // row_half_mirror
  v += dpp_move<0x140, 0xF, true>(v);   // row_mirror: every lane of a row of 16 holds the row's sum
  // row_bcast:15 (lane 15 of a row to the next row; row 0 has no source and reads 0) and row_bcast:31 (lane 31 to rows 2
  // and 3; rows 0 and 1 read 0), all rows enabled: row 2 also picks up row 1's sum in the first step, which only lane 63's
  // chain (r3 + r2) + (r1 + r0) never sees - same association order as the row-masked form, no "old" operand to set up
  v += dpp_move<0x142, 0xF, true>(v);
  v += dpp_move<0x143, 0xF, true>(v);   // lane 63 holds the total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// Transposing reductions for several sums at once (the cells kernel's projection).  v_permlane32_swap / v_permlane16_swap
// (new on gfx950) exchange half-waves / odd and even rows of two registers, so that one addition folds TWO values at once:
//   fold32(x, y): lanes 0-31 hold x[L] + x[L + 32], lanes 32-63 hold y[L - 32] + y[L]
//   fold16(x, y): rows 0 and 2 hold x's row pairs (0 + 1, 2 + 3), rows 1 and 3 hold y's
// i.e. three instructions per stage for two values where the DPP chain spends three per value.
__device__ __forceinline__ double fold32(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double fold16(double x, double y) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// sum over the eight lanes of an octet, in every lane of it
__device__ __forceinline__ double octet_sum(double v) {
  v += dpp_move<0x141, 0xF, true>(v);   // row_half_mirror
  v += dpp_move<0xB1, 0xF, true>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xF, true>(v);    // quad_perm [2,3,0,1]
  return v;
}

// g ~ sqrt(x), returns also ir ~ 1/sqrt(x).  x must be a positive normal number (r^2 of a separation in
// Mpc/h); x = 0 gives NaN, which is what the reference's r_par / r produces there too.
// One third-order (Halley-type) step from the 2^-24 hardware seed: y' = y (1 + e/2 + 3e^2/8), e = 1 - x y^2,
// leaves a relative error of order e^3 ~ 1e-22, i.e. the result is limited by the final roundings only.
__device__ __forceinline__ void sqrt_rsqrt(double x, double& g, double& ir) {
  const double y = __builtin_amdgcn_rsq(x);   // v_rsq_f64
  const double e = fma(-(x * y), y, 1.0);
  const double p = fma(0.375, e, 0.5);
  ir = fma(y * e, p, y);
  g = x * ir;
}

// 1/sqrt(x) alone, third-order like sqrt_rsqrt (the dispersion model's fixed-point iteration, which amplifies rounding)
__device__ __forceinline__ double rsqrt3(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  const double p = fma(0.375, e, 0.5);
  return fma(y * e, p, y);
}

// 1/sqrt(x) for the streaming integrand of the fast kernels (callers fold x * ir into their next fma): ONE Newton step
// from the hardware seed, y' = y (1 + e/2), e = 1 - x y^2.  The seed is good to 5.3e-8, i.e. |e| <= 1.05e-7, so the result
// is good to 3/8 e^2 <= 4.4e-15 - 20 ulp at worst, 1.0 ulp on average (measured on the hardware, tools/devmath_check.hip) -
// against one ulp for the third-order step of sqrt_rsqrt / rsqrt3.  It feeds mu_r = r_par / r and the interval coordinate
// r / (c h), where an error of that size is of the order of what the conditioning of the Gaussian factor already makes of
// the last-bit errors of 1/sigma (2 y^2 eps).  Measured end to end: 200 oracle-checked points per configuration, max rel
// dchi2 7.3e-15 -> 1.1e-14 (config 3) and 1.2e-14 -> 1.7e-14 (BOSS), medians unchanged; 131072 points of the wide fuzz box
// against the generic kernel (third order throughout), theory vectors 1.7e-14 -> 2.6e-14 (config 3), 1.3e-15 -> 3.1e-15
// (BOSS) of their maximum; bench batch max rel dchi2 vs the oracle 2.6e-14 -> 3.6e-14 - for one instruction of ~53 per
// integrand point (same-box A/B: config 3 25.50 -> 25.10 ms, BOSS 15.30 -> 15.00 ms; profiles/r02/p_rsq_second_order.txt).
// 1/sigma keeps its third-order step (recip): its error enters the exponent at full weight.
__device__ __forceinline__ double rsqrt_nr(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  return fma(y * e, 0.5, y);
}

// 1/x for a normal x of either sign (velocity dispersion, Jacobians): y' = y (1 + e + e^2), e = 1 - x y, error ~ e^3
__device__ __forceinline__ double recip(double x) {
  const double y = __builtin_amdgcn_rcp(x);   // v_rcp_f64
  const double e = fma(-x, y, 1.0);
  return fma(y, fma(e, e, e), y);
}

// 2^(j/256), j = 0..255, to be staged in LDS by the caller (2 KB): table of exp_nonpos
constexpr int kExpNonposTab = 256;
__device__ __forceinline__ double exp2_frac(int j) { return exp2((double)j * (1.0 / kExpNonposTab)); }

// p*f + c as ONE v_fma_f64.  hipcc prefers the two-address v_fmac_f64 and then needs a v_mov_b64 per step to
// re-materialise the constant addend it overwrites; naming the three-address form keeps Horner steps at one
// instruction each.
__device__ __forceinline__ double fma3(double p, double f, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(f), "v"(c));
  return r;
}

// exp(a) for a <= 0.  a is split as a = (256 m + j) ln2/256 + f with |f| <= ln2/512, so
// exp(a) = 2^m * T[j] * exp(f) and a degree-4 Taylor polynomial in f is exact to 4e-17.
// The clamp keeps the range reduction exact for absurdly negative arguments (exp(-750) already underflows
// to 0); v_max_f64 returns the bound for a NaN argument, so callers that must propagate NaN inputs add
// their own poison term (the theory kernels do).
__device__ __forceinline__ double exp_nonpos(double a, const double* __restrict__ tab) {
  a = fmax(a, -750.0);
  const double n = rint(a * 369.3299304675746);                  // 256 / ln 2
  double f = fma(n, -0x1.62e42fee00000p-9, a);                   // ln2/256, high 32 bits: n*hi is exact
  f = fma(n, -0x1.a39ef35793c76p-41, f);                         //          remainder
  const int ni = (int)n;
  const double t = tab[ni & (kExpNonposTab - 1)];
  double p = fma3(f, 1.0 / 24.0, 1.0 / 6.0);
  p = fma(p, f, 0.5);
  p = fma(p, f, 1.0);
  p = fma(p, f, 1.0);
  return ldexp(t * p, ni >> 8);
}

// p*f + c with the constant addend in a scalar register pair (one constant-bus operand is allowed per VALU
// instruction on gfx9): no v_mov to re-materialise constants inside the hot loop.
__device__ __forceinline__ double fma_s(double p, double f, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(f), "s"(c));
  return r;
}

// 1/x from ONE Newton step, y' = y (1 + e), e = 1 - x y: error e^2.  The v_rcp_f64 seed is good to 2^-24-ish (measured:
// tools/devmath_check.hip), so the result is good to ~3e-15 - the streaming integrand's 1/sigma_v, whose error enters the
// Gaussian's exponent as z^2 eps, i.e. <= 1e-13 relative where the integrand has any weight (z^2 <= 36), against the
// end-to-end budget of 1e-10 on chi2 / xi_l ("accuracy budget"; the contract is 1e-6).
__device__ __forceinline__ double recip_nr(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, y, 1.0);
  return fma(y, e, y);
}

// Twice the refined 1/sqrt(x) in four instructions: y' = y (3 - x y^2) / 2 is Newton's step, and the callers want 2 y' anyway
// once every length they feed in is halved (an exact power-of-two rescaling on the host side of the loop: X = r'^2 / 4,
// 2 y' = 4 / r', X * 2y' = r', (num / 2) * 2y' = 2 mu_r - see vk_kernel_fast.h: uni_accum).  Same accuracy as rsqrt_nr.
__device__ __forceinline__ double rsqrt_nr_x2(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double h = x * y;
  const double w = fma(-h, y, 3.0);
  return y * w;
}

// Twice the third-order 1/sqrt(x) (rsqrt3) in the same six instructions: y (2 + e p2) with p2 = 1 + 3/4 e.  For the half-unit
// callers whose result is ill-conditioned in 1/r (the dispersion model's last pass and final evaluation, vk_kernel_fast.h).
__device__ __forceinline__ double rsqrt3_x2(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  const double p2 = fma(0.75, e, 1.0);
  return y * fma(e, p2, 2.0);
}

// exp(-z^2/2) for callers that carry ynum and 1/SV with z = ynum / SV in units of kExpScale, i.e. y = kExpScale * z
// (kExpScale^2 = 128/ln2 * 2^-18, folded into the per-point amplitude and the velocity nodes).
//   * y' = min(|ynum| |inv_sv|, 1) comes out of ONE v_mul_f64: abs modifiers on the inputs and the VOP3 clamp bit on the
//     result.  y' = 1 is |z| = 2^9 / sqrt(128/ln2) = 37.7, where exp(-z^2/2) = 4e-309 is below the normal range anyway:
//     the saturation returns a denormal-sized value instead of the true denormal-or-zero, and it bounds n below, which is
//     what lets the power of two be applied by integer arithmetic on the exponent field (no v_ldexp_f64, no v_ashr).
//   * with yn = -(y' 2^9)^2 = -(z^2/2) 256/ln2 the range reduction is n = rint(yn / step), d = yn - n step, with a table of
//     T = 256 / step entries per octave.  n comes from the add-a-magic-number rounding: nd = fma(-y', y', 1.5 2^34 step)
//     has ulp 2^-18 step, i.e. it holds 1.5 2^52 + n in its mantissa - the low word of nd IS n (two's complement,
//     |n| <= 2^18) with no v_rndne / v_cvt -, nn = nd - magic = n 2^-18 step exactly, and ds = fma(-y', y', -nn) is exact as
//     well (|d| <= 1/2, up to a tie of the single rounding).
//   * exp = 2^(n >> log2 T) 2^(j/T) p(d), j = n mod T.  The table entry t_j = L 2^(j/T) (L = the polynomial's leading
//     coefficient, so that the Horner form is monic and its addends are scalar constants) is stored with j 2^(20 - log2 T)
//     subtracted from its high word: adding n 2^(20 - log2 T) to that word - ONE v_lshl_add_u32 - then adds exactly
//     (n - j) / T = n >> log2 T to the exponent field.  The field cannot underflow: n >> log2 T >= -1024 and L puts t_j at
//     2^25 or above.
//   * two table forms, chosen per kernel (template parameter EXPT of the fast kernels):
//       EXPT 0: T = 256, degree-3 Taylor polynomial (|f| <= ln2/512: remainder f^4/24 <= 1.4e-13, always positive,
//               2.8e-14 on average); 2 KB; its byte offset (j << 3) is one SDWA shift of the low byte of n.  The reads are
//               random over the 32 bank pairs: ~3.5 LDS cycles per group of 32 lanes instead of 1.
//       EXPT 1: T = 64, degree 4 (|f| <= ln2/128: remainder f^5/120 <= 3.9e-14), each entry replicated for the 32 lanes of
//               a ds_read_b64 group, entry j of lane l at byte (j << 8) + ((l & 31) << 3): 16 KB, conflict-free by
//               construction.  Two more vector instructions than EXPT 0; for the kernels whose LDS pipe is the co-limiter
//               (lanes kernel with the anisotropic sum: 10 ds_read_b128 per integrand point).
// NaN: the clamp bit turns a NaN product into 0 when MODE.DX10_CLAMP is set (the HSA default); the kernels that use this
// clear that bit at entry (clamp_keeps_nan) so a NaN still propagates, and their final multiply by inv_sv carries a NaN
// 1/SV in any case.
constexpr double kExpScale = 13.589148804608305 * 0x1p-9;        // sqrt(128/ln 2) 2^-9
constexpr double kExpC = 709.782712893383996843;                 // 1024 ln 2: f = ds * kExpC for both table forms
template <int EXPT> struct ExpCfg;
template <> struct ExpCfg<0> {
  static constexpr int kBits = 8, kEntries = 256, kDoubles = 256, kDegree = 3;
  static constexpr double kMagic = 0x1.8p34;                     // ulp 2^-18
  static constexpr double kLead = kExpC * kExpC * kExpC / 6.0;
};
template <> struct ExpCfg<1> {
  static constexpr int kBits = 6, kEntries = 64, kDoubles = 64 * 32, kDegree = 4;
  static constexpr double kMagic = 0x1.8p36;                     // ulp 2^-16: one step = four steps of the 256-entry form
  static constexpr double kLead = kExpC * kExpC * kExpC * kExpC / 24.0;
};
constexpr int kExpTab = ExpCfg<0>::kDoubles;                      // legacy name: doubles of the plain table

// entry j of the table, as stored: L 2^(j/T) with j << (20 - log2 T) taken off the high word (see above)
template <int EXPT>
__device__ __forceinline__ double exp_table_entry(int j) {
  typedef ExpCfg<EXPT> C;
  const double t = C::kLead * exp2((double)j * (1.0 / C::kEntries));
  return __hiloint2double(__double2hiint(t) - (j << (20 - C::kBits)), __double2loint(t));
}
// slot e of the staged table (EXPT 1: [entry][lane of the group of 32])
template <int EXPT>
__device__ __forceinline__ double exp_table_slot(int e) { return exp_table_entry<EXPT>(EXPT == 1 ? (e >> 5) : e); }

// MODE.DX10_CLAMP = 0 for this wave: v_*_f64 ... clamp then returns NaN for a NaN result instead of 0
__device__ __forceinline__ void clamp_keeps_nan() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 8, 1), 0"); }

// `tab`: LDS address of the staged table; `lane_off` = (lane & 31) << 3 (EXPT 1 only)
template <int EXPT>
__device__ __forceinline__ double exp_gauss(double ynum, double inv_sv, const double* __restrict__ tab, unsigned lane_off = 0) {
  typedef ExpCfg<EXPT> C;
  double yp;
  asm("v_mul_f64 %0, |%1|, |%2| clamp" : "=v"(yp) : "v"(ynum), "v"(inv_sv));
  const double nd = fma(-yp, yp, C::kMagic);
  const double nn = nd - C::kMagic;
  const double d = fma(-yp, yp, -nn);
  const int n = __double2loint(nd);
  unsigned off;
  if (EXPT == 0) {
    // (n & 255) << 3 as one SDWA shift of the low byte of n
    asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(3), "v"(n));
    __builtin_assume(off < 2048u);       // lets the table's LDS base fold into the ds_read offset field
  } else {
    // ((n & 63) << 8) | lane_off in two instructions (left to itself the compiler shifts, masks and adds: three)
    asm("v_and_b32 %0, 63, %1\n\tv_lshl_or_b32 %0, %0, 8, %2" : "=&v"(off) : "v"(n), "v"(lane_off));
  }
  // The table starts the kernel's dynamic LDS, whose base address is 0 (no kernel that calls this has static LDS:
  // tests/test_host.py reads .group_segment_fixed_size = 0 out of the code object), so the byte offset IS the LDS address - the
  // compiler, which only knows the base as a link-time symbol, would spend a v_add_u32 on it (1 of ~37 instructions of the
  // BOSS loop: 14.34 -> 14.23 ms per 65536 points, config 3 25.18 -> 24.96, same box).  `tab` documents the contract.
  (void)tab;
  double t;
  asm("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(off));
  const double ts = __hiloint2double(__double2hiint(t) + (n << (20 - C::kBits)), __double2loint(t));
  double q;
  if (C::kDegree == 3) {
    q = d + 3.0 / kExpC;
    q = fma_s(q, d, 6.0 / (kExpC * kExpC));
    q = fma_s(q, d, 6.0 / (kExpC * kExpC * kExpC));
  } else {
    q = d + 4.0 / kExpC;
    q = fma_s(q, d, 12.0 / (kExpC * kExpC));
    q = fma_s(q, d, 24.0 / (kExpC * kExpC * kExpC));
    q = fma_s(q, d, 24.0 / (kExpC * kExpC * kExpC * kExpC));
  }
  return ts * q;
}

}  // namespace vkm
