// vk_common.h: device-side views, table evaluation helpers, per-point scalars - part of libvictor_hip.so (see victor_hip.hip for the overview and DESIGN.md section 5).
#pragma once
#include <hip/hip_runtime.h>

#include "victor_hip.h"
#include "vk_devmath.h"
#include "vk_views.h"

namespace vk {

// threadIdx.x as a fresh value for code that runs once at the end of a work item (hand-over of partial sums, chi-square
// tail): what such code derives from the thread index is then formed there instead of being hoisted to the top of the
// kernel and carried - or spilled: 8 bytes per thread were 134 MB of scratch writes per 65536-point launch - through
// the integrand loops.
__device__ __forceinline__ int late_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}


constexpr int kBlock = 256;               // 4 wavefronts
constexpr int kWaves = kBlock / 64;
constexpr int kMaxEll = 3;
constexpr int kVrVars = 5;               // V1, Da, V2, Ge1, Ge2 (see vk_tables.vr)
constexpr int kMaxParts = 8;             // workgroups that may share one (point, s bin) plane of the point-major kernel

// --------------------------------------------------------------------------------------------------
// device-side views
// --------------------------------------------------------------------------------------------------
// (PPView - a vk_pp living in global memory, device pointers - is in vk_views.h: the host-compiled units hold it in vk_ctx)

constexpr int kLikeRed = 4 * kWaves;     // doubles of reduction scratch behind the theory vector of a fused tail (4 sums x kWaves)
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
  int grids_in_lds;       // bit 0 / 1: beta_d / beta_c equal the real-space tables' grid beta_r bit for bit (BOSS: one 31-value grid
                          // serves all three) - a fused tail then searches the copy its kernel holds in LDS instead of global memory
  const double* tri;      // [slices][M/2+1][M+2], M = N rounded up to even: every slice's quadratic form folded onto its upper triangle, by circular diagonals (vk_kernel_like.h)
  const double* logdet;
  const double* eig;
  int like_form;
  double nmocks, nparams;
  double* lnl;
  double* chi2;
#ifdef VK_PHASES
  long long* stamps;      // profiling build only: the theory kernel's marks (slots 8-12 belong to the fused tail)
#endif
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
  const double* xw_scaled;  // [n_x + 1][2]: {kExpScale x_k, w_k} (point-major fast kernel; last pair is padding)
  double xw_max;            // max |kExpScale x_k| (cell_in_table)
  // velocity nodes in groups of equal quadrature weight (Simpson: a handful of distinct values) for the kernels whose node
  // loop is wave-uniform (lanes, cells): the weight multiplies the group's sum once instead of every integrand point
  const double* xgw;        // [n_xg + 1][2]: {kExpScale x_k in group order, the group's weight at its LAST node else 0} (scalar-cache reads)
  int n_xg;                 // nodes in xgw: the n_x nodes without those of weight zero (which contribute nothing and could not end a group)
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
  const double* sva;      // anisotropic sigma_v(r, mu) for the fast kernels: [sv.n_int][sv_n_mu-1][4][4] patches in powers of the
                          // r-interval's local coordinate in [0, 1) and of (mu - mu_j), followed by the sv_n_mu mu knots (built in vk_create)
  int sva_doubles;        // length of that block (0: not available)
  int uni_n;              // unified refined grid (vk_tables.uni_*): one interval index for sigma_v, V1, xi^r_l
  double uni_u0, uni_inv_h;
  const double* uni_sv_v;
  const double* uni_xi;
  const double* uni_xic;  // Legendre sum regrouped in powers of mu_r^2 (anisotropic sum)
  const double* uni_vb;   // beta-dependent V1 on the unified grid (vr_beta_dep), [n_beta_r-1][uni_n][4][4]
  const double* uni_v2;   // V2 = r Delta delta on the unified grid (fixed velocity tables), [uni_n][4]
  const double* uni_da;   // Da = delta - 2 Delta/3 on the unified grid (fixed velocity tables), [uni_n][4]
  const double* uni_ge;   // Ge1, Ge2 (empirical_corr gradient tables) on the unified grid, [2][uni_n][4]
  const double* uni_dab;  // beta-dependent Da on the unified grid (vr_beta_dep), [n_beta_r-1][uni_n][4][4]
  const double* uni_empb; // beta-dependent V2, Ge1, Ge2 (vr_beta_dep), degree 6 in beta: [3][n_beta_r-1][uni_n][4][7]
  int uni_lut_n;          // > 0: union-grid form (arbitrary knots), cells of the look-up table
  double uni_lut_inv_g;
  const unsigned short* uni_lut;
  const double* uni_knots;  // [uni_n + 1]
  int matter_lb;          // linear_bias matter model: amplitudes carry 1/bias (ccf_model.py:358-370,426-435)
  int vr_beta_dep;        // velocity tables are PCHIP-in-beta polynomials (rebuilt per point)
  const double* vr_emp;   // [3][n_beta_r-1][vr.n_int][4][7]: V2, Ge1, Ge2 of the empirical_corr branch, degree 6 in beta
  int from_data;          // ccf_model.py:618-619,675-679
  int empirical;          // ccf_model.py:451-459
  int rsd;                // VK_RSD_*
  int niter;              // fixed-point iterations of the dispersion / Kaiser coordinate shift
  int kaiser_approx;      // ccf_model.py:730-738
  int coord_shift;        // ccf_model.py:698-707
  int lanes_per_block;    // lanes kernel: consecutive work items per workgroup (kWaves, or n_s: a 64-point chunk per workgroup)
  int sbins_per_item;     // s bins handled by one workgroup visit
  int team;               // waves cooperating on one s bin (1, 2 or 4)
  double* out;            // theory: [n][n_ell*n_s];  xi_smu: [n][n_mu][n_s]
  // ---- staging tables that do not depend on the batch (built once in vk_create) --------------------------------
  const double* exp_tab;  // [ExpCfg<0>::kDoubles] exp table, plain form (vk_devmath.h), computed on the device so every launch copies the same bits
  const double* exp_tab_rep;  // [ExpCfg<1>::kDoubles] 64 entries x 32 lane replicas (the lanes kernel with the anisotropic sum)
  const double* stage_mu; // [n_mu][kMuRec] {mu, sqrt(1-mu^2), W_0, W_1, W_2, 0} of the context's own (mu, W) grid, or NULL
  const double* image;    // LDS image of this kernel variant's batch-constant tables (vk_kernel_fast.h: copy_image), or NULL
  double wsum[3];         // sum_i W_l[i]: the "-1" of ccf_model.py:690 projects to -sum_i W_l[i] (not 0 for l > 0)
  unsigned nx_magic;      // ceil(2^32 / n_x):  idx / n_x  == __umulhi(idx, nx_magic)  for every idx of the (mu, v) plane
  unsigned nmu_magic;     // ceil(2^32 / n_mu): cell / n_mu likewise (cells kernel)
  unsigned ns_magic;      // ceil(2^32 / n_s):  cell / n_s, the mu-major cell order of `xi_out` (unused for n_s = 1, where it does not fit)
  // 1 (cells kernel, n_ell = 1 instantiations): CCFModel.theory_xi (ccf_model.py:538-690) - every cell's xi^s(mu_i, s_j) is the
  // result, stored to out[n][n_mu][n_s], instead of being projected onto multipoles; no partial sums, no tail
  int xi_out;
  // ---- small and medium batches: finer work split and the chi-square in the same launch ----------------------------
  int parts;              // point-major: workgroups sharing one (point, s-bin group) plane; cells: workgroups per point
  int cells_per_item;     // cells kernel: (s bin, mu) cells of one work item (multiple of 64); parts = ceil(n_s n_mu / this)
  int fuse;               // 1: the workgroup that completes a point's theory vector also computes its chi2 / lnL (`like`)
  unsigned* counters;     // [n] workgroups finished per point; zero on entry, reset to zero by the finishing workgroup
  double* partial;        // [n][n_s][parts][kMaxEll] partial projections (point-major, parts > 1)
  // 1 (point-major kernel, parts > 1, launches whose workgroups are all resident at once): no completion counter - `partial`
  // is the context's polling area, whose slots hold kPollEmpty between launches; the workgroup of a point's LAST work item
  // waits for every partial sum of the point to appear and puts kPollEmpty back (finish_point)
  int poll;
  int* poll_failed;       // pinned host word, set when a polling workgroup gave up (kPollTicks): the context then reports an error
  LikeArgs like;
  // A single-point call through host buffers carries its parameter row HERE, in the kernel arguments: read from the pinned
  // host buffer it is a PCIe round trip of ~2.7 us in front of every workgroup's loads (vector loads return in order, so
  // even the table image waits for it), from the argument segment it is one more line of it (tools/gpu_phases.py ... api).
  double row0[VK_NPAR];
  int inline_row;
  // 1: the theory vectors in `out` are a result of this call (the caller reads them back, or a separate chi-square launch does).
  // 0 with `fuse`: nobody reads `out` - the chi-square is taken from LDS in the same workgroup - and the kernels that hold a
  // point's whole vector in LDS (cells) do not write it to HBM at all (BOSS, 65536 points: 31.5 MB per launch that nobody read).
  int want_theory;
#ifdef VK_PHASES
  long long* stamps;      // profiling build only (make phases): [workgroup][16] wall_clock64() marks of the point-major kernel
#endif
};

#ifdef VK_PHASES
#define VK_STAMP(a, k) do { if ((a).stamps && threadIdx.x == 0 && blockIdx.x < 4096) (a).stamps[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#else
#define VK_STAMP(a, k) do { } while (0)
#endif

// piecewise-cubic table resident in LDS
struct PPLds {
  const double* knots;
  const double* coef;
  int n_int;
  int lead;
  double inv_h;
  double lo, hi, x_u0;
};

// Interval of u (already clamped to the table range): exact for uniform knots (inv_h > 0); for nearly uniform knots
// (inv_h < 0: |inv_h| is the mean spacing, every knot within 0.4 spacings of the uniform position - e.g. bin centres
// that are mean pair separations) the uniform estimate is corrected by at most one interval against the knots;
// arbitrary knots (inv_h == 0) use a binary search.
__device__ __forceinline__ int pp_interval(const PPLds& t, double u) {
  int i;
  if (t.inv_h != 0.0) {
    const int n_uniform = t.n_int - t.lead;
    const double tt = (u - t.x_u0) * fabs(t.inv_h);
    i = (int)tt;
    i = min(max(i, 0), n_uniform - 1) + t.lead;
    if (t.lead && u < t.x_u0) i = 0;
    if (t.inv_h < 0.0) {
      if (u < t.knots[i]) {
        i = max(i - 1, 0);
      } else if (u >= t.knots[i + 1]) {
        i = min(i + 1, t.n_int - 1);
      }
    }
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

__device__ __forceinline__ double wave_sum(double v) { return vkm::wave_sum(v); }

// One entry of TheoryArgs::xgw - {x'_k, weight of the group that ends with node k, else 0} - read through the scalar cache.
// The four dwords are loaded as integers so that "is this the last node of its group" stays a 32-bit scalar compare on the
// weight's high word (as a test on the double the compiler forms a 64-bit compare, which gfx950 only has on the vector ALU).
struct VelocityNode { double x, w; int last; };
__device__ __forceinline__ VelocityNode load_node(const double* xgw, int k) {
  typedef int node_words __attribute__((ext_vector_type(4)));
  typedef const node_words __attribute__((address_space(4))) * node_ptr;
  const node_words q = ((node_ptr)(unsigned long long)xgw)[k];
  VelocityNode n;
  n.x = __hiloint2double(q.y, q.x);
  n.w = __hiloint2double(q.w, q.z);
  n.last = q.w;
  return n;
}

// The kernel-argument struct spans thirteen 64-byte lines and the compiler fetches its fields where it first needs them (they
// are rematerialised rather than kept in SGPRs), so a workgroup meets the lines one at a time: up to ten dependent
// round trips from the scalar cache to L2 spread over its serial path.  Touch every line once at kernel entry - all loads in
// flight together - and the later fetches hit the scalar cache.  Measured on a single-point launch (tools/gpu_phases.py):
// see DESIGN.md section 5.
template <int BYTES>
__device__ __forceinline__ void warm_kernarg_lines() {
  static_assert(BYTES <= 896, "extend the list of lines");
  const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  unsigned d0, d1, d2, d3, d4, d5, d6, d7, d8, d9, d10, d11;
  asm volatile(
      "s_load_dword %0, %12, 0x0\n\t"
      "s_load_dword %1, %12, 0x40\n\t"
      "s_load_dword %2, %12, 0x80\n\t"
      "s_load_dword %3, %12, 0xc0\n\t"
      "s_load_dword %4, %12, 0x100\n\t"
      "s_load_dword %5, %12, 0x140\n\t"
      "s_load_dword %6, %12, 0x180\n\t"
      "s_load_dword %7, %12, 0x1c0\n\t"
      "s_load_dword %8, %12, 0x200\n\t"
      "s_load_dword %9, %12, 0x240\n\t"
      "s_load_dword %10, %12, 0x280\n\t"
      "s_load_dword %11, %12, 0x2c0\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7), "=&s"(d8), "=&s"(d9),
        "=&s"(d10), "=&s"(d11)
      : "s"(kp)
      : "memory");
  if constexpr (BYTES > 768) {          // lines 12 and 13 (the inline parameter row, the profiling build's pointer)
    unsigned e0, e1;
    if constexpr (BYTES > 832)
      asm volatile("s_load_dword %0, %2, 0x300\n\ts_load_dword %1, %2, 0x340\n\ts_waitcnt lgkmcnt(0)" : "=&s"(e0), "=&s"(e1) : "s"(kp) : "memory");
    else
      asm volatile("s_load_dword %0, %1, 0x300\n\ts_waitcnt lgkmcnt(0)" : "=&s"(e0) : "s"(kp) : "memory");
  }
}

// Data handed from one workgroup to another INSIDE a launch (partial projections, theory vectors awaiting their
// chi-square) is written and read with device-scope relaxed atomics: on gfx950 those are write-through stores / L2-coherent
// loads (sc1), which is all the coherence the eight per-XCD L2s need.  The alternative - ordinary stores plus
// __threadfence() - costs a write-back and an invalidation of the whole L2 per workgroup (buffer_wbl2 / buffer_inv): every
// table the next workgroup stages then misses, and a 64-point batch ran 30 % slower than without the fused tail.
__device__ __forceinline__ void store_shared(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_shared(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Eight consecutive doubles written with store_shared(), fetched as four 16-byte L2-coherent loads in flight together
// (the compiler keeps atomic loads in program order with a wait after each: eight dependent round trips to memory).
// `p` must be 16-byte aligned; v[0..7] receive p[0..7].
typedef double vk_shared2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void load_shared_x8(const double* p, double (&v)[8]) {
  vk_shared2 a, b, c, d;
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\t"
      "global_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
      "global_load_dwordx4 %2, %4, off offset:32 sc1\n\t"
      "global_load_dwordx4 %3, %4, off offset:48 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
      : "v"(p)
      : "memory");
  v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y; v[6] = d.x; v[7] = d.y;
}

// "This workgroup finished one of the `total` work items of `point`": returns true (to every thread) in the workgroup
// that finished the last one.  Everything the workgroups stored with store_shared() before their call is then readable
// with load_shared().  The ordering is spelled out, not implied: EVERY thread first waits for its own write-through (sc1)
// stores - `s_waitcnt vmcnt(0)`; a workgroup-scope release fence emits only lgkmcnt(0) on gfx950 and the back-off barrier
// puts no vmcnt wait in front of s_barrier -, the barrier then orders thread 0's device-scope counter increment behind the
// waits of all four waves, and the finishing workgroup's sc1 loads are issued after its own increment has returned and a
// second barrier (MI355X_MICROARCH.md, "Valid forms": sc1 stores, every storing wave drained, one lane signalling behind a
// workgroup barrier, sc1 loads after the returned add).  tests/test_host.py checks the wait in the shipped code object.
// The finishing workgroup resets counters[point] for the next launch.  `flag`: one int of LDS.
// (The round-2 form - a workgroup-scope release fence without the vmcnt wait - measured the same speed and is not ordered:
// profiles/r03/a_handoff_drain_ab.txt.  It is gone from the source; there is no build switch that removes the wait.)
// Hand-off by POLLING (TheoryArgs::poll; one point per call - the reference's calling convention - and the mailbox server's
// launches): a slot of the polling area holds kPollEmpty, a NaN pattern no sum can take (hardware NaNs are 0x7ff8000000000000;
// only a caller's NaN parameter with exactly this payload could propagate it, and then the wait below ends in the time-out), until
// its producer's ONE 8-byte write-through store lands; the finishing workgroup re-reads its slots (sc1 loads) until none is
// empty.  Every slot vouches for itself, so no ordering between stores is needed - no drain, no barrier, no counter round trip:
// a single point's launch is 1.5 us shorter.  Deadlock-free by construction: the finisher is the workgroup of the point's LAST
// work item, and the host only polls in launches that fit on the chip at once.  kPollTicks (wall_clock64 counts 100 MHz: 5 s)
// is a guard against waiting for ever on a broken launch, not a code path: it fails the call (TheoryArgs::poll_failed).
constexpr unsigned long long kPollEmpty = 0x7ff8a5a57ff8a5a5ULL;    // both halves equal: the area is filled by hipMemsetD32
constexpr long long kPollTicks = 500000000LL;
__device__ __forceinline__ bool poll_is_empty(double v) { return (unsigned long long)__double_as_longlong(v) == kPollEmpty; }

__device__ __forceinline__ void drain_shared_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// parameter row of a point (see TheoryArgs::row0)
__device__ __forceinline__ const double* param_row(const TheoryArgs& a, long long point) {
  return a.inline_row ? a.row0 : a.params + point * VK_NPAR;
}

__device__ __forceinline__ bool point_completed(unsigned* counters, long long point, unsigned total, int* flag) {
  drain_shared_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(counters + point, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (done == total) __hip_atomic_store(counters + point, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = done == total ? 1 : 0;
  }
  __syncthreads();
  return *flag != 0;
}

// Growth term and powers of the bias (ccf_model.py:426-443, 358-370): v_r(r) = -gb [V1 + av V2](r/c) / (3 aH_true).
// `extra` collects the inputs that must poison the outputs when they are NaN/inf.
__device__ __forceinline__ double growth_amplitude(const TheoryArgs& a, const double* row, double fs8, double* av,
                                                   double* extra) {
  double growth = fs8 * a.inv_sigma8;
  double binv = 1.0;
  if (a.matter_lb) {
    const double bias = row[VK_P_BIAS];
    if (a.from_data) growth = row[VK_P_BETA] * bias;
    binv = vkm::recip(bias);
    *extra += bias;
  }
  // velocity template: v_r = growth_t V_t(r/c), growth_t = fsigma8 vt_amp / apar  ==  -gb V_t / (3 aH_true)
  if (a.matter_vt) growth = -3.0 * a.iaH * a.vt_amp * fs8;
  *av = 0.0;
  if (a.empirical && !a.matter_vt) {
    *av = row[VK_P_AV] * binv;
    *extra += *av;
  }
  return growth * binv;
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
      double g, ir;
      vkm::sqrt_rsqrt(fma(1.0 - m * m, e2, 1.0), g, ir);     // <= 2 ulp (vk_devmath.h); the argument is >= min(1, eps^2) > 0
      v = ps.apar * g;
      if (lane == 0 || lane == 49) v *= 0.5;
    }
    c = wave_sum(v) * h;
  } else {
    c = row[VK_P_ASTAR];
  }
  // per-point reciprocals through the refined v_rcp_f64 (<= 1 ulp, vk_devmath.h) instead of IEEE divisions: this set-up is a
  // serial chain every workgroup runs before it can start, ~3 us of a 15 us single-point launch with the library forms
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
  ps.poison = 0.0 * (gb + sigv + ps.aperp + ps.apar + eps + c + ps.A + extra + (a.n_beta_r > 0 ? row[VK_P_BETA] : 0.0));
  return ps;
}

}  // namespace vk
