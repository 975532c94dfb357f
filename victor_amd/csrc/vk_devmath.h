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

// 2^(j/256), j = 0..255, to be staged in LDS by the caller (2 KB)
constexpr int kExpTab = 256;
__device__ __forceinline__ double exp2_frac(int j) { return exp2((double)j * (1.0 / kExpTab)); }

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
  const double t = tab[ni & (kExpTab - 1)];
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

// exp(-z^2/2) for callers that carry ynum and 1/SV with z = ynum / SV in units of kExpScale, i.e. y = kExpScale * z
// (kExpScale^2 = 128/ln2 * 2^-30, folded into the per-point amplitude and the velocity nodes).
//   * y' = min(|ynum| |inv_sv|, 1) comes out of ONE v_mul_f64: abs modifiers on the inputs and the VOP3 clamp bit on the
//     result.  y' = 1 is |z| = 2^15 / sqrt(128/ln2) = 2411, where exp(-z^2/2) = 0 in double anyway (it underflows beyond
//     |z| = 38.6), so the saturation is exact and bounds n below.
//   * with yn = -(y' 2^15)^2 = -(z^2/2) 256/ln2 the range reduction is n = rint(yn), d = yn - n, exp = 2^(n/256) exp(d ln2/256).
//     n comes from the add-a-magic-number rounding: nd = fma(-y', y', 1.5 2^22) has ulp 2^-30, i.e. it holds
//     1.5 2^52 + n in its mantissa - the low word of nd IS n (two's complement, |n| <= 2^30) with no v_rndne / v_cvt -,
//     nn = nd - magic = n 2^-30 exactly, and ds = fma(-y', y', -nn) = d 2^-30 is exact as before (|d| <= 1/2, up to a
//     tie of the single rounding).
//   * the degree-4 Taylor polynomial in f = d ln2/256 is evaluated in ds with every coefficient divided by the leading
//     one, c4 2^120 with c4 = (ln2/256)^4/24, so the Horner addends are scalar constants and the leading one lives in
//     the table: `tab_c4[j]` = c4 2^120 2^(j/256) (exp2_frac_c4 below).  Powers of two only rescale: same bits as a
//     polynomial in d.
// NaN: the clamp bit turns a NaN product into 0 when MODE.DX10_CLAMP is set (the HSA default); the kernels that use this
// clear that bit at entry (clamp_keeps_nan) so a NaN still propagates, and their final multiply by inv_sv carries a NaN
// 1/SV in any case.
constexpr double kExpScale = 13.589148804608305 * 0x1p-15;       // sqrt(128/ln 2) 2^-15
constexpr double kExpMagic = 0x1.8p22;                           // ulp 2^-30
constexpr double kExpC1 = 0.0027076061740622863;                 // ln2/256
constexpr double kExpC4 = kExpC1 * kExpC1 * kExpC1 * kExpC1 / 24.0;
__device__ __forceinline__ double exp2_frac_c4(int j) { return kExpC4 * 0x1p120 * exp2((double)j * (1.0 / kExpTab)); }

// MODE.DX10_CLAMP = 0 for this wave: v_*_f64 ... clamp then returns NaN for a NaN result instead of 0
__device__ __forceinline__ void clamp_keeps_nan() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 8, 1), 0"); }

__device__ __forceinline__ double exp_gauss(double ynum, double inv_sv, const double* __restrict__ tab_c4) {
  double yp;
  asm("v_mul_f64 %0, |%1|, |%2| clamp" : "=v"(yp) : "v"(ynum), "v"(inv_sv));
  const double nd = fma(-yp, yp, kExpMagic);
  const double nn = nd - kExpMagic;
  const double d = fma(-yp, yp, -nn);
  const int ni = __double2loint(nd);
  const double t = tab_c4[ni & (kExpTab - 1)];
  double q = d + 4.0 / kExpC1 * 0x1p-30;                                  // c3/c4
  q = fma_s(q, d, 12.0 / (kExpC1 * kExpC1) * 0x1p-60);                    // c2/c4
  q = fma_s(q, d, 24.0 / (kExpC1 * kExpC1 * kExpC1) * 0x1p-90);           // c1/c4
  q = fma_s(q, d, 24.0 / (kExpC1 * kExpC1 * kExpC1 * kExpC1) * 0x1p-120); // 1/c4
  return ldexp(t * q, ni >> 8);
}

}  // namespace vkm
