// victor_hip.hip - the launch side of libvictor_hip.so: context, table upload, kernel selection, the evaluating entry points of
// the C ABI (include/victor_hip.h).  The host-only parts of the library are translation units of their own, compiled by the host
// compiler (vk_host.h is what they share with this file): vk_ledger.cpp (the polling hand-off's launch rule and the device-wide
// ledger of reserved waiters), vk_walk.cpp (vk_walk_*: the walkers' step loop), vk_serve.cpp (vk_serve_mailboxes), vk_rccl.cpp
// (vk_comm_*: RCCL through dlopen).
//
// Device code lives in the headers next to this file:
//   vk_common.h          argument structs, LDS table evaluation, per-point scalars (AP factors, growth amplitudes)
//   vk_devmath.h         FP64 sqrt/rsqrt, reciprocal and exp building blocks (<= 2 ulp, measured on hardware)
//   vk_kernel_generic.h  K1 generic: every RSD model and option, library math, knot search   + K1x xi(s, mu)
//   vk_kernel_fast.h     K1 point-major fast path (wave = point x s bin, lanes over the (mu, v) plane)
//   vk_kernel_lanes.h    K1 lanes-over-the-batch (wave = s bin x 64 points; batch-constant tables, large batches) - the A/B
//                        yardstick of tools/ and of the mapping tests, compiled into the DEVELOPMENT build only (VK_DEV_LANES)
//   vk_kernel_cells.h    K1 cells (workgroup = point, lanes over (s, mu) cells, v loop innermost; per-point tables)
//   vk_kernel_like.h     K2 residual . precision . residual, log det, likelihood form, NaN guard
//
// K1 restates CCFModel.theory_xi (streaming branch victor/ccf_model.py:589-690; the other branches :658-784),
// theory_multipoles (:816-825) and utils.multipoles_from_fn (victor/utils.py:45-56); K2 restates CCFFit.chi_squared
// (victor/ccf_fit.py:349-354), get_interpolated_{covariance,precision} (:195-260) and log_likelihood (:444-481).
// All arithmetic is IEEE binary64 on the vector ALU: an evaluation is n_s*n_mu*n_x (= 200 000) integrand points of
// ~80 FP64 instructions each against ~100 bytes of HBM traffic, so the kernels are laid out for VALU issue and LDS
// gather bandwidth, not for HBM or MFMA.  launch_theory() picks the K1 variant per call.

#include <hip/hip_runtime.h>
#include <chrono>
#include <cstddef>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <algorithm>
#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "victor_hip.h"
#include "vk_host.h"
#include "vk_ledger.h"
#include "vk_kernel_cells.h"
#include "vk_kernel_fast.h"
#include "vk_kernel_generic.h"
#ifdef VK_DEV_LANES          // development build only (libvictor_hip_dev.so: `make dev`, build_native(dev=True)): the yardstick kernel
#include "vk_kernel_lanes.h"
#endif
#include "vk_kernel_like.h"

// The theory kernels' instantiations are generated in translation units of their own (vk_instances.h names what lives where);
// here they are declared only.
#include "vk_instances.h"
namespace vk {
VK_UNIT_CELLS_STREAMING(extern template)
VK_UNIT_CELLS_DISPERSION(extern template)
VK_UNIT_CELLS_KAISER(extern template)
VK_UNIT_FAST_STREAMING(extern template)
VK_UNIT_FAST_DISPERSION(extern template)
VK_UNIT_GENERIC(extern template)
}  // namespace vk

using namespace vk;

// LDS image of a kernel variant's batch-constant tables (vk_kernel_fast.h: copy_image): run that variant's own staging
// code once and keep what it left in LDS.  kind: 0 point-major, 1 cells, 2 lanes (development build).
template <int NLR>
__global__ __launch_bounds__(kBlock) void vk_image_kernel(TheoryArgs a, int kind, int with_da, double* image, int n) {
  extern __shared__ double lds[];
  for (int e = threadIdx.x; e < n; e += kBlock) lds[e] = 0.0;
  __syncthreads();
  const int n_sva = (with_da & 4) ? a.sva_doubles : 0;        // bit 2: the SVA instantiation's layout
  with_da &= 3;                                               // the mode's layout (vk_kernel_fast.h: mode_layout)
  if (kind == 0) {
    stage_fast<NLR>(a, make_fast_plan(a.n_mu, a.n_x, a.uni_n, NLR, a.n_beta_r, a.uni_lut_n, with_da, 0, n_sva), lds, with_da);
  } else if (kind == 1) {
    stage_cells<NLR>(a, make_cells_plan(a.n_mu, a.n_x, a.n_s, a.uni_n, NLR, a.n_beta_r, a.uni_lut_n, with_da, 64, 0, n_sva), lds, with_da);
  }
#ifdef VK_DEV_LANES
  else {
    stage_lanes<NLR>(a, make_lanes_plan(a.n_mu, a.n_x, a.uni_n, NLR, a.uni_lut_n), lds);
  }
#endif
  __syncthreads();
  for (int e = threadIdx.x; e < n; e += kBlock) image[e] = lds[e];
}

// joint fits (block-diagonal covariance): lnL and chi2 of the blocks add, in block order; a failed block fails the point
__global__ void vk_joint_sum_kernel(const double* ws, long long n, int n_ctx, long long block_stride, double* lnl, double* chi2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double l = 0.0, c = 0.0;
  for (int q = 0; q < n_ctx; ++q) {
    l += ws[q * block_stride + i];
    c += ws[q * block_stride + n + i];
  }
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  if (!(fabs(l) < inf)) {          // -inf from a block's guard, or NaN
    l = -inf;
    c = inf;
  }
  if (lnl) lnl[i] = l;
  if (chi2) chi2[i] = c;
}

// vk_create: the tables every launch copies into LDS unchanged - the scaled exp table of vk_devmath.h and the mu records
// {mu, sqrt(1 - mu^2), W_0, W_1, W_2, 0} of the context's own grid - computed once, on the device (same bits as the
// in-kernel staging of the general-grid entry points)
__global__ void vk_init_stage_kernel(const double* mu, const double* w_ell, int n_mu, int n_ell, double* exp_tab, double* exp_tab_rep,
                                     double* stage_mu) {
  for (int j = threadIdx.x; j < vkm::ExpCfg<0>::kDoubles; j += blockDim.x) exp_tab[j] = vkm::exp_table_slot<0>(j);
  for (int j = threadIdx.x; j < vkm::ExpCfg<1>::kDoubles; j += blockDim.x) exp_tab_rep[j] = vkm::exp_table_slot<1>(j);
  for (int i = threadIdx.x; i < n_mu; i += blockDim.x) {
    const double m = mu[i];
    double* rec = stage_mu + i * kMuRec;
    rec[0] = m;
    rec[1] = sqrt(1.0 - m * m);
    for (int l = 0; l < kMaxEll; ++l) rec[2 + l] = (l < n_ell) ? w_ell[l * n_mu + i] : 0.0;
    rec[5] = 0.0;
  }
}


// ==================================================================================================
// host side
// ==================================================================================================
using vkh::check_opts;
using vkh::cpu_relax;
using vkh::fail;
using vkh::host_scratch;
using vkh::HostScratch;
using vkh::sync_knobs;
using vkh::zc_begin;
using vkh::zc_finish;
using vkl::kPollPoints;

namespace {

thread_local std::string g_create_err;
std::atomic<int> g_poll_reserved{0};    // waiters reserved by the contexts of this copy of the library (<= kPollBudget)
constexpr int kPollRetryLaunches = 256; // a context whose reservation was refused asks again after this many launches at the latest

#ifdef VK_PHASES
long long* g_stamps = nullptr;
#endif
std::atomic<unsigned> g_knob_gen{1};

void load_knobs(vk_ctx* ctx) {
  Knobs k;
  ctx->knob_gen = g_knob_gen.load(std::memory_order_relaxed);
  const char* dev = getenv("VICTOR_HIP_DEV");
  if (!dev || strcmp(dev, "1") != 0) {       // not a development run: every knob at its default, whatever the environment holds
    ctx->knobs = k;
    return;
  }
  if (const char* env = getenv("VICTOR_HIP_SPLIT")) {
    int sp = 0, t = 0, q = 1;
    const int got = sscanf(env, "%d,%d,%d", &sp, &t, &q);
    if (got >= 2 && sp >= 1 && (t == 1 || t == 2 || t == 4) && (t == 1 || sp == 1) && q >= 1 && q <= 16) {
      k.split_s = sp;
      k.split_t = t;
      k.split_q = got == 3 ? q : 0;
    }
  }
  if (const char* env = getenv("VICTOR_HIP_FUSE_MAX")) k.fuse_max = atoll(env);
  if (const char* env = getenv("VICTOR_HIP_CELLS_PARTS")) k.cells_parts = atoi(env);
  if (const char* env = getenv("VICTOR_HIP_LIKE_WIDE")) k.like_wide = atoi(env) ? 1 : 0;
  k.no_zero_copy = getenv("VICTOR_HIP_NO_ZERO_COPY") != nullptr;
  if (const char* env = getenv("VICTOR_HIP_SPIN_MAX")) k.spin_max = atoll(env);
  if (const char* env = getenv("VICTOR_HIP_NO_POLL")) k.no_poll = atoi(env) != 0;
  k.force_generic = getenv("VICTOR_HIP_FORCE_GENERIC") != nullptr;
  if (const char* env = getenv("VICTOR_HIP_POINT_CAP")) k.point_cap = atoll(env);
  if (const char* env = getenv("VICTOR_HIP_MAPPING"))
    k.mapping = !strcmp(env, "point") ? 1 : !strcmp(env, "cells") ? 2 : !strcmp(env, "lanes") ? 3 : -1;
  k.like_untiled = getenv("VICTOR_HIP_LIKE_UNTILED") != nullptr;
  k.no_graph = getenv("VICTOR_HIP_NO_GRAPH") != nullptr;
  k.no_fuse = getenv("VICTOR_HIP_NO_FUSE") != nullptr;
  k.no_inline_row = getenv("VICTOR_HIP_NO_INLINE_ROW") != nullptr;
  ctx->knobs = k;
}

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

void drop_graphs(vk_ctx* ctx) {
  for (auto& kv : ctx->graphs) (void)hipGraphExecDestroy(kv.second);
  ctx->graphs.clear();
  ctx->graph_seen.clear();
  ctx->graph_kernel.clear();
}

int ensure_scratch(vk_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return VK_OK;
  drop_graphs(ctx);   // captured graphs hold pointers into the old scratch allocation
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  ctx->d_scratch = nullptr;
  ctx->scratch_bytes = 0;
  VK_HIP(ctx, hipMalloc((void**)&ctx->d_scratch, bytes));
  ctx->scratch_bytes = bytes;
  return VK_OK;
}

// Launch on the context's stream.  Dynamic LDS above the 64 KiB default needs an explicit per-kernel opt-in
// (gfx950 has 160 KiB per CU).
template <typename Kern, typename Args>
int launch_on_stream(vk_ctx* ctx, Kern kern, int grid, size_t lds, const Args& a) {
  if (lds > 64 * 1024) {
    int& granted = ctx->lds_opt_in[reinterpret_cast<const void*>(kern)];      // (the opt-in is per kernel and sticky: asked for once)
    if ((int)lds > granted) {
      VK_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      granted = (int)lds;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, ctx->stream, a);
  VK_HIP(ctx, hipGetLastError());
  return VK_OK;
}

void choose_split(const vk_ctx* ctx, long long n, int n_s, int* spi, int* team, int* parts) {
  *parts = 1;
  // A/B knob "spi,team[,parts]": s bins per workgroup visit, waves cooperating on one s bin (1, 2 or 4), workgroups per plane
  if (ctx->knobs.split_s > 0) {
    *spi = ctx->knobs.split_s < n_s ? ctx->knobs.split_s : n_s;
    *team = ctx->knobs.split_t;
    if (ctx->knobs.split_q > 0) *parts = ctx->knobs.split_q;
    return;
  }
  // Measured (tools/gpu_split_sweep.py, resident, config 3 / BOSS): four s bins per workgroup (one per wave) from ~50 points
  // on, one s bin per workgroup (four cooperating waves) below that, and for a handful of points the (mu, v) plane of every
  // s bin is shared by several workgroups so that a single point still spreads over >= 160 of them.
  const long long want = 4LL * ctx->n_cu;
  if (n >= want) { *spi = n_s; *team = 1; return; }
  if (n * ((n_s + 3) / 4) >= want / 2) { *spi = 4; *team = 1; return; }
  *spi = 1; *team = 4;
  // Workgroups per (mu, v) plane for a handful of points.  Rounds 2-3 spread a single point over up to four per plane
  // (>= 160 workgroups: the shortest launch for ONE point in flight).  The reference's calling convention under load is
  // several chains with one point each (section 6 of DESIGN.md), and what counts there is how many such launches are resident
  // at once: two per plane cost a single call +0.9 us (20.2 -> 21.2 us) and give 8 / 16 chains through the GPU owner
  // process +15 % / +14 % (260 -> 300 k, 415 -> 474 k evaluations/s; tools/gpu_single_split_ab.py,
  // profiles/r04/f_single_point_split_ab.txt).
  const long long wgs = n * n_s;
  long long q = (160 + wgs - 1) / wgs;
  *parts = (int)(q < 1 ? 1 : (q > 2 ? 2 : q));
}

template <int RSD, int NLR>
int launch_generic_nl(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  switch (a.n_ell) {
    case 1: return launch_on_stream(ctx, vk_theory_kernel<RSD, NLR, 1>, grid, lds, a);
    case 2: return launch_on_stream(ctx, vk_theory_kernel<RSD, NLR, 2>, grid, lds, a);
    case 3: return launch_on_stream(ctx, vk_theory_kernel<RSD, NLR, 3>, grid, lds, a);
  }
  return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
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

template <int NLR, int GRID, int MODE>
int launch_fast_ngf(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  switch (a.n_ell) {
    case 1: return launch_on_stream(ctx, vk_theory_fast_kernel<NLR, 1, GRID, MODE>, grid, lds, a);
    case 2: return launch_on_stream(ctx, vk_theory_fast_kernel<NLR, 2, GRID, MODE>, grid, lds, a);
    case 3: return launch_on_stream(ctx, vk_theory_fast_kernel<NLR, 3, GRID, MODE>, grid, lds, a);
  }
  return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
}

// anisotropic sigma_v(r, mu) template on the fast kernels (SVA instantiations): streaming model, lattice form
template <int NLR>
int launch_fast_sva(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  switch (a.n_ell) {
    case 1: return launch_on_stream(ctx, vk_theory_fast_kernel<NLR, 1, 0, kModeStreaming, 1>, grid, lds, a);
    case 2: return launch_on_stream(ctx, vk_theory_fast_kernel<NLR, 2, 0, kModeStreaming, 1>, grid, lds, a);
    case 3: return launch_on_stream(ctx, vk_theory_fast_kernel<NLR, 3, 0, kModeStreaming, 1>, grid, lds, a);
  }
  return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
}

// ... and the dispersion model with it (cells kernel only)
template <int NLR>
int launch_cells_sva_disp(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  switch (a.n_ell) {
    case 1: return launch_on_stream(ctx, vk_theory_cells_kernel<NLR, 1, 0, kModeDispersion, 1>, grid, lds, a);
    case 2: return launch_on_stream(ctx, vk_theory_cells_kernel<NLR, 2, 0, kModeDispersion, 1>, grid, lds, a);
    case 3: return launch_on_stream(ctx, vk_theory_cells_kernel<NLR, 3, 0, kModeDispersion, 1>, grid, lds, a);
  }
  return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
}

template <int NLR>
int launch_cells_sva(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  if (a.rsd == VK_RSD_DISPERSION) return launch_cells_sva_disp<NLR>(ctx, a, grid, lds);
  switch (a.n_ell) {
    case 1: return launch_on_stream(ctx, vk_theory_cells_kernel<NLR, 1, 0, kModeStreaming, 1>, grid, lds, a);
    case 2: return launch_on_stream(ctx, vk_theory_cells_kernel<NLR, 2, 0, kModeStreaming, 1>, grid, lds, a);
    case 3: return launch_on_stream(ctx, vk_theory_cells_kernel<NLR, 3, 0, kModeStreaming, 1>, grid, lds, a);
  }
  return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
}

template <int NLR, int GRID>
int launch_fast_ng(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  if (a.sv_n_mu > 0) return launch_fast_sva<NLR>(ctx, a, grid, lds);
  if (a.rsd == VK_RSD_DISPERSION)
    return a.from_data ? launch_fast_ngf<NLR, GRID, kModeDispersionFromData>(ctx, a, grid, lds)
                       : launch_fast_ngf<NLR, GRID, kModeDispersion>(ctx, a, grid, lds);
  return a.from_data ? launch_fast_ngf<NLR, GRID, kModeFromData>(ctx, a, grid, lds)
                     : launch_fast_ngf<NLR, GRID, kModeStreaming>(ctx, a, grid, lds);
}

template <int NLR>
int launch_fast_nl(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  return a.uni_lut_n > 0 ? launch_fast_ng<NLR, 1>(ctx, a, grid, lds) : launch_fast_ng<NLR, 0>(ctx, a, grid, lds);
}

#ifdef VK_DEV_LANES
template <int NLR, int GRID>
int launch_lanes_ng(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  switch (a.n_ell) {
    case 1: return launch_on_stream(ctx, vk_theory_lanes_kernel<NLR, 1, GRID>, grid, lds, a);
    case 2: return launch_on_stream(ctx, vk_theory_lanes_kernel<NLR, 2, GRID>, grid, lds, a);
    case 3: return launch_on_stream(ctx, vk_theory_lanes_kernel<NLR, 3, GRID>, grid, lds, a);
  }
  return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
}

template <int NLR>
int launch_lanes_nl(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  return a.uni_lut_n > 0 ? launch_lanes_ng<NLR, 1>(ctx, a, grid, lds) : launch_lanes_ng<NLR, 0>(ctx, a, grid, lds);
}
#endif

template <int NLR, int GRID, int MODE>
int launch_cells_ngf(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  switch (a.n_ell) {
    case 1: return launch_on_stream(ctx, vk_theory_cells_kernel<NLR, 1, GRID, MODE>, grid, lds, a);
    case 2: return launch_on_stream(ctx, vk_theory_cells_kernel<NLR, 2, GRID, MODE>, grid, lds, a);
    case 3: return launch_on_stream(ctx, vk_theory_cells_kernel<NLR, 3, GRID, MODE>, grid, lds, a);
  }
  return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
}

template <int NLR, int GRID>
int launch_cells_ng(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  if (a.rsd == VK_RSD_KAISER || a.rsd == VK_RSD_EUCLID) return launch_cells_ngf<NLR, GRID, kModeKaiser>(ctx, a, grid, lds);
  if (a.sv_n_mu > 0) return launch_cells_sva<NLR>(ctx, a, grid, lds);
  if (a.rsd == VK_RSD_DISPERSION)
    return a.from_data ? launch_cells_ngf<NLR, GRID, kModeDispersionFromData>(ctx, a, grid, lds)
                       : launch_cells_ngf<NLR, GRID, kModeDispersion>(ctx, a, grid, lds);
  return a.from_data ? launch_cells_ngf<NLR, GRID, kModeFromData>(ctx, a, grid, lds)
                     : launch_cells_ngf<NLR, GRID, kModeStreaming>(ctx, a, grid, lds);
}

template <int NLR>
int launch_cells_nl(vk_ctx* ctx, const TheoryArgs& a, int grid, size_t lds) {
  return a.uni_lut_n > 0 ? launch_cells_ng<NLR, 1>(ctx, a, grid, lds) : launch_cells_ng<NLR, 0>(ctx, a, grid, lds);
}

template <int RSD>
int launch_xi_smu(vk_ctx* ctx, const TheoryArgs& a, int nlr, int grid, size_t lds) {
  switch (nlr) {
    case 1: return launch_on_stream(ctx, vk_xi_smu_kernel<RSD, 1>, grid, lds, a);
    case 2: return launch_on_stream(ctx, vk_xi_smu_kernel<RSD, 2>, grid, lds, a);
    case 3: return launch_on_stream(ctx, vk_xi_smu_kernel<RSD, 3>, grid, lds, a);
  }
  return fail(ctx, VK_E_ARG, "bad number of real-space multipoles %d", nlr);
}

// K1x generic: one wave per (point, mu, s) cell, library math, any knot layout (vk_kernel_generic.h) - what serves
// vk_xi_smu_batch where the cells kernel cannot go
int launch_xi_generic(vk_ctx* ctx, TheoryArgs a, int nlr) {
  const LdsPlan pl = make_plan(a.n_mu, a.n_x, a.n_ell, a.sv.n_int, a.vr.n_int, a.xi.n_int, nlr, a.n_beta_r);
  const size_t lds = (size_t)pl.total * sizeof(double);
  if (lds > 160 * 1024) return fail(ctx, VK_E_ARG, "tables need %zu bytes of LDS (> 160 KiB)", lds);
  const long long cap = 8LL * ctx->n_cu;
  const int grid = (int)(a.n < cap ? a.n : cap);
  a.xi_out = 0;
  a.sbins_per_item = 1;
  a.team = 1;
  a.parts = 1;
  a.exp_tab = ctx->d_exp_tab;
  ctx->last_kernel = "vk_xi_smu_kernel";
  switch (a.rsd) {
    case VK_RSD_STREAMING: return launch_xi_smu<VK_RSD_STREAMING>(ctx, a, nlr, grid, lds);
    case VK_RSD_DISPERSION: return launch_xi_smu<VK_RSD_DISPERSION>(ctx, a, nlr, grid, lds);
    case VK_RSD_KAISER: return launch_xi_smu<VK_RSD_KAISER>(ctx, a, nlr, grid, lds);
    default: return launch_xi_smu<VK_RSD_EUCLID>(ctx, a, nlr, grid, lds);
  }
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
  a->sva = ctx->d_sva;
  a->sva_doubles = ctx->sva_doubles;
  a->uni_n = ctx->uni_n;
  a->uni_u0 = ctx->uni_u0;
  a->uni_inv_h = ctx->uni_inv_h;
  a->uni_sv_v = ctx->d_uni_sv_v;
  a->uni_xi = ctx->d_uni_xi;
  a->uni_xic = ctx->d_uni_xic;
  a->uni_vb = ctx->d_uni_vb;
  a->uni_v2 = ctx->d_uni_v2;
  a->uni_da = ctx->d_uni_da;
  a->uni_ge = ctx->d_uni_ge;
  a->uni_dab = ctx->d_uni_dab;
  a->uni_empb = ctx->d_uni_empb;
  a->uni_lut_n = ctx->uni_lut_n;
  a->uni_lut_inv_g = ctx->uni_lut_inv_g;
  a->uni_lut = ctx->d_uni_lut;
  a->uni_knots = ctx->d_uni_knots;
  a->vr_beta_dep = ctx->vr_beta_dep;
  a->vr_emp = ctx->d_vr_emp;
  a->from_data = o->from_data ? 1 : 0;
  a->empirical = (o->empirical_corr && !ctx->matter_vt) ? 1 : 0;   // the template-mean branch ignores Av (ccf_model.py:483-490)
  if (a->empirical && a->vr_beta_dep && !a->vr_emp)
    return fail(ctx, VK_E_ARG, "empirical_corr with a beta-dependent velocity profile needs vk_tables.vr_emp");
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
  a->xw_scaled = ctx->d_xws;
  a->xw_max = ctx->xw_max;
  a->xgw = ctx->d_xgw;
  a->n_xg = ctx->n_xg;
  *nlr = o->assume_isotropic ? 1 : ctx->n_ell_r;
  return VK_OK;
}

// LDS image for launches on the context's own grid (NULL: the kernel stages entry by entry).  Built on first use, never
// while the stream is being captured into a graph (the first, eager call of a shape has built it by then).
// with_da: the mode's LDS layout (vk_kernel_fast.h: mode_layout - 0 streaming, 1 kaiser / euclid_special, 2 dispersion)
const double* get_image(vk_ctx* ctx, const TheoryArgs& a, int kind, int nlr, int with_da, int image_end, bool sva = false) {
  if (!a.stage_mu || image_end <= 0) return nullptr;
  const int key = kind * 100 + nlr * 10 + with_da + (sva ? 4 : 0);
  auto hit = ctx->images.find(key);
  if (hit != ctx->images.end()) return hit->second;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
  const size_t lds = (size_t)image_end * sizeof(double);
  double* img = nullptr;
  if (lds > 160 * 1024 || hipMalloc((void**)&img, lds) != hipSuccess) return nullptr;
  bool ok = true;
  switch (nlr) {
    case 1:
      if (lds > 64 * 1024) ok = hipFuncSetAttribute(reinterpret_cast<const void*>(vk_image_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      if (ok) hipLaunchKernelGGL(vk_image_kernel<1>, dim3(1), dim3(kBlock), lds, ctx->stream, a, kind, with_da | (sva ? 4 : 0), img, image_end);
      break;
    case 2:
      if (lds > 64 * 1024) ok = hipFuncSetAttribute(reinterpret_cast<const void*>(vk_image_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      if (ok) hipLaunchKernelGGL(vk_image_kernel<2>, dim3(1), dim3(kBlock), lds, ctx->stream, a, kind, with_da | (sva ? 4 : 0), img, image_end);
      break;
    default:
      if (lds > 64 * 1024) ok = hipFuncSetAttribute(reinterpret_cast<const void*>(vk_image_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
      if (ok) hipLaunchKernelGGL(vk_image_kernel<3>, dim3(1), dim3(kBlock), lds, ctx->stream, a, kind, with_da | (sva ? 4 : 0), img, image_end);
      break;
  }
  if (!ok || hipGetLastError() != hipSuccess) {
    (void)hipFree(img);
    return nullptr;
  }
  ctx->images[key] = img;       // same stream as the launches that will read it: ordered without a host sync
  return img;
}

// idx / d as one mul_hi: exact for every idx < 2^32 / d when magic = ceil(2^32 / d) (the planes here are < 2^22)
unsigned div_magic(int d) { return (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); }

// `like`: the likelihood stage of this call, or NULL (theory only).  *fused is set when the theory kernel took the
// chi-square as well (the caller then skips the K2 launch).
int launch_theory(vk_ctx* ctx, TheoryArgs a, int nlr, const LikeArgs* like, bool* fused) {
  if (fused) *fused = false;
  ctx->last_polled = false;
  if (a.n <= 0) return VK_OK;
  if (a.n > (1LL << 31) / ((long long)a.n_s * kMaxParts)) return fail(ctx, VK_E_ARG, "batch of %lld points is too large for one launch", a.n);
  const int N = a.n_ell * a.n_s;
  // The mailbox server's launches: the result of a chain's point must not depend on which other chains posted at the same
  // moment, so every decision that normally follows the batch size is taken as for ONE point - the point-major kernel with the
  // single-point split; the per-point arithmetic then is that of CCFFit.log_likelihood in a process of its own, bit for bit.
  const long long n_dec = (ctx->split_as_single && a.n <= kServeMaxBatch) ? 1 : a.n;
  choose_split(ctx, n_dec, a.n_s, &a.sbins_per_item, &a.team, &a.parts);
  a.exp_tab = ctx->d_exp_tab;
  a.exp_tab_rep = ctx->d_exp_tab_rep;
  a.nx_magic = div_magic(a.n_x);
  a.nmu_magic = div_magic(a.n_mu);
  a.counters = ctx->d_counters;
  a.partial = ctx->d_partial;
  a.poll = 0;
  a.poll_failed = ctx->d_poll_failed;
  if (ctx->h_poll_failed && *ctx->h_poll_failed)
    return fail(ctx, VK_E_HIP, "a workgroup waited %.0f s for partial sums that never arrived (an earlier launch of this context); "
                               "the context is unusable", (double)kPollTicks * 1e-8);
  a.fuse = 0;
  a.image = nullptr;
#ifdef VK_PHASES
  {
    static long long* d_stamps = nullptr;
    if (!d_stamps) (void)hipMalloc((void**)&d_stamps, 4096 * 16 * sizeof(long long));
    (void)hipMemsetAsync(d_stamps, 0, 4096 * 16 * sizeof(long long), ctx->stream);
    a.stamps = d_stamps;
    g_stamps = d_stamps;
  }
#endif
  // the fast kernel (streaming only) packs LDS byte offsets of the mu and (x, w) records into 16 bits each
  // fast kernels: the streaming model, and the dispersion model on fixed velocity tables (cells / point-major only)
  // (fixed velocity tables: Da / Ge staged once; beta-dependent ones - linear_bias on a reconstructed real-space ccf - rebuilt
  // per point from their beta polynomials, uni_dab / uni_empb)
  const bool da_tabs = a.vr_beta_dep ? (a.uni_dab && (!a.empirical || a.uni_empb)) : (a.uni_da && (!a.empirical || a.uni_ge));
  const bool disp = a.rsd == VK_RSD_DISPERSION && da_tabs;
  // kaiser / euclid_special (no velocity integral, no sigma_v): one evaluation per (s, mu) cell in the cells kernel
  const bool kais = (a.rsd == VK_RSD_KAISER || a.rsd == VK_RSD_EUCLID) && da_tabs;
  const bool emp_ok = !a.empirical || (a.vr_beta_dep ? a.uni_empb != nullptr : a.uni_v2 != nullptr);
  // anisotropic sigma_v(r, mu): its bicubic patches ride in LDS for the streaming model on the lattice form (SVA instantiations)
  const bool sva = a.sv_n_mu > 0 && !kais;
  // (the dispersion model with it: the cells kernel only, at every batch size)
  const bool sva_disp = sva && disp;
  const int layout = disp ? 2 : (kais ? 1 : 0);               // LDS layout of the mode (vk_kernel_fast.h: mode_layout)
  const bool sva_ok = !sva || (a.sva_doubles > 0 && (a.rsd == VK_RSD_STREAMING || disp) && !a.from_data && a.uni_lut_n == 0);
  const int n_sva = sva ? a.sva_doubles : 0;
  bool fast = (a.rsd == VK_RSD_STREAMING || disp || kais) && ctx->fast_ok && emp_ok && sva_ok &&
              a.n_mu <= 1024 && a.n_x <= 2048 && !ctx->knobs.force_generic;
  // chi-square inside the theory kernel: point-major and cells kernels only, up to fuse_max points (A/B: DESIGN.md section 5)
  // A/B (tools/gpu_small_batch_ab.py, config 3 / BOSS, resident): the fused launch wins up to ~256 points (64 points: 41.2 vs
  // 43.7 us, 33.0 vs 34.9), ties at 512 and is 1 % behind the two-launch path from 1024 on (461.5 vs 455.4 us), where the
  // separate chi-square kernel overlaps the tail of the theory kernel
  // separate chi-square kernel overlaps the tail of the theory kernel.  Large batches (tools/gpu_fuse_ab.py, cells kernel, one
  // workgroup per point, 8192 / 65536 points): config 3 (fixed covariance, tiled K2 of 0.16 ms) 3.620 -> 3.602 / 28.16 -> 28.33 ms,
  // i.e. nothing to gain; BOSS (per-point blended precision, K2 0.37 ms) 2.190 -> 2.165 / 17.51 -> 17.08 ms: fused from 8192
  // points on when the covariance depends on beta.
  const long long kFuseMaxDefault = 512, kFuseBlendedMin = 8192;
  const bool fuse_by_size = ctx->knobs.fuse_max >= 0 ? a.n <= ctx->knobs.fuse_max
                                                     : (a.n <= kFuseMaxDefault || (like && like->n_beta_c > 0 && a.n >= kFuseBlendedMin));
  // (kaiser / euclid_special: the theory kernel is ~50 times shorter, a chi-square launch of its own would be a tenth of the step)
  const bool want_fuse = like && !ctx->knobs.no_fuse && (fuse_by_size || kais) && like_lds_doubles(N) * sizeof(double) <= 32 * 1024;
  if (like) a.like = *like;
#ifdef VK_PHASES
  a.like.stamps = a.stamps;
#endif
  const long long kDefaultCap = 256;                                // VICTOR_HIP_POINT_CAP: workgroups per CU in a launch of the generic kernel (the fast kernels take one item per workgroup)
  const long long cap = (ctx->knobs.point_cap > 0 ? ctx->knobs.point_cap : kDefaultCap) * ctx->n_cu;
  const int mapping = ctx->knobs.mapping;                           // VICTOR_HIP_MAPPING: 0 = choose by batch size
  // cells variant: one workgroup per point with the velocity loop innermost; needs n_mu >= 64 (a wave's 64 cells must not
  // straddle more than two s bins)
  const bool cells_ok = fast && a.n_mu >= 64 && a.n_mu <= 4096 && a.n_x <= 2048;
  if (a.xi_out) {
    // CCFModel.theory_xi (ccf_model.py:538-690) - xi^s on the caller's (s, mu) grid, no projection: the cells kernel's n_ell = 1
    // instantiations store every cell's value (vk_kernel_cells.h: xi_out), whatever n_mu is (the limit n_mu >= 64 above belongs to
    // the projection's two-segment reduction).  A point's cells are cut into ranges as for the projected launches - but ranges
    // hand nothing over here, so neither the counters nor the partial-sum area limit them.  Grids and table forms the fast
    // kernels cannot take go to the generic kernel.
    const long long all_cells = (long long)a.n_s * a.n_mu;
    if (fast && a.n_ell == 1 && all_cells <= (1LL << 24)) {
      const int whole = (int)((all_cells + 63) / 64 * 64);
      int cpi = a.n > kPartialPoints ? whole : (a.n < 128 ? 256 : (a.n < 256 ? 512 : 1024));
      if (cpi > whole) cpi = whole;
      const int R = (int)((all_cells + cpi - 1) / cpi);
      const CellsPlan plc = make_cells_plan(a.n_mu, a.n_x, a.n_s, a.uni_n, nlr, a.n_beta_r, a.uni_lut_n, layout, 0, 0, n_sva);
      const size_t lds_c = (size_t)plc.total * sizeof(double);
      if (lds_c <= 160 * 1024 && a.n * R < (1LL << 31)) {
        ctx->last_kernel = "vk_theory_cells_kernel";
        a.parts = R;
        a.cells_per_item = cpi;
        a.ns_magic = div_magic(a.n_s);
        a.image = nullptr;                 // the caller's own grid: staged inside the kernel
        const int grid_c = (int)(a.n * R);
        switch (nlr) {
          case 1: return launch_cells_nl<1>(ctx, a, grid_c, lds_c);
          case 2: return launch_cells_nl<2>(ctx, a, grid_c, lds_c);
          case 3: return launch_cells_nl<3>(ctx, a, grid_c, lds_c);
        }
      }
    }
    return launch_xi_generic(ctx, a, nlr);
  }
#ifdef VK_DEV_LANES
  // Development build only: the lanes-over-the-batch kernel (wave = s bin x 64 points; the north star's mapping), behind the cells
  // kernel at every batch size since round 3 (tools/gpu_lanes_vs_cells.py, same box, M evals/s at 8192 / 65536 / 262144 points:
  // config 3 2.58 / 2.66 / 2.66 against 2.23 / 2.58 / 2.65) and kept as the yardstick of tools/ and of the mapping tests:
  // VICTOR_HIP_MAPPING=lanes selects it.  The product library does not contain it (round 5).
  {
    const bool lanes_ok = fast && a.n_beta_r == 0 && !a.empirical && !a.from_data && !disp && !kais && !sva;   // per-point tables need a workgroup per point
    const long long waves = ((a.n + 63) >> 6) * (long long)a.n_s;
    const long long blocks_l = (waves + kWaves - 1) / kWaves;
    const size_t lds_l = (size_t)make_lanes_plan(a.n_mu, a.n_x, a.uni_n, nlr, a.uni_lut_n).total * sizeof(double);
    if (lanes_ok && lds_l <= 160 * 1024 && mapping == 3) {
      ctx->last_kernel = "vk_theory_lanes_kernel";
      a.parts = 1;
      a.image = get_image(ctx, a, 2, nlr, 0, make_lanes_plan(a.n_mu, a.n_x, a.uni_n, nlr, a.uni_lut_n).image_end);
      a.lanes_per_block = kWaves;
      const long long blocks = blocks_l;
      // One workgroup per four items, never a grid-stride loop by default (a cap that makes workgroups loop leaves a ragged
      // tail of 0.6 ms items: 131072 points ran at 1.61 M evals/s under a 64-per-CU cap against 2.35 M without)
      const long long capl = (long long)INT32_MAX;
      const int grid_l = (int)(blocks < capl ? blocks : capl);
      switch (nlr) {
        case 1: return launch_lanes_nl<1>(ctx, a, grid_l, lds_l);
        case 2: return launch_lanes_nl<2>(ctx, a, grid_l, lds_l);
        case 3: return launch_lanes_nl<3>(ctx, a, grid_l, lds_l);
      }
    }
  }
#endif
  // crossover against the point-major kernel measured between 512 and 768 points (config 3) and near 500 (BOSS),
  // tools/gpu_small_batch_ab.py: one workgroup per point needs ~2.5 workgroups per CU to keep the SIMDs fed
  // crossover against the point-major kernel (whose finer split wins for a handful of points): config 3 / BOSS 8 points
  // 21.1 / 20.3 us point-major vs 25.9 / 24.2 cells, 16: 25.7 / 24.5 vs 27.2 / 25.1, 32: 41.3 / 36.7 vs 30.6 / 28.0
  // (re-measured after both kernels lost their grid-stride loops, tools/gpu_cells_min_sweep.py, profiles/r03/z_*: point-major
  // ahead up to 12 / 16 points, level at 20, the cells kernel ahead from 24 / 28 on)
  const long long cells_min = 20;
  const bool cells = cells_ok && (kais || sva_disp || (mapping ? mapping == 2 : n_dec >= cells_min));   // kaiser, dispersion x sigma_v(r, mu): this kernel only
  if (cells) {
    ctx->last_kernel = "vk_theory_cells_kernel";
    // A point's n_s * n_mu cells may be cut into `parts` ranges, one workgroup each (vk_kernel_cells.h): enough ranges to give
    // every CU ~6 workgroups when the batch alone would not (a trip per wave - 256 cells - at least, and a range never
    // so short that an s bin spreads over more than kMaxParts of them); VICTOR_HIP_CELLS_PARTS overrides.
    const int all_cells = a.n_s * a.n_mu;
    const int min_cpi = std::max(256, ((a.n_mu + 5) / 7 + 63) / 64 * 64);
    int R = 1;
    if (ctx->knobs.cells_parts > 0) {
      R = ctx->knobs.cells_parts;
      R = std::max(1, std::min(R, (all_cells + min_cpi - 1) / min_cpi));
    } else if (a.n <= kPartialPoints) {
      // measured (tools/gpu_small_batch_ab.py, config 3 / BOSS): ranges of one trip per wave (256 cells) below 128 points, two
      // below 256, four from there on - 64 points: 56.6 -> 41.7 us against the point-major kernel, 1024 points: 496 -> 462 us
      // against one workgroup per point; whole trips per wave only (a multiple of 256 cells)
      // (kaiser: a cell is one evaluation, not 50 - ranges only for a handful of points)
      // (n_dec, not a.n: the mailbox server's launches split every point as a single-point call does, whatever shares the launch -
      // with a.n the ranges of a kaiser request changed from 256 to 1024 cells once eight chains posted together, and with them
      // the rounding of its sums)
      const int cpi_want = kais ? (n_dec < 8 ? 256 : (n_dec < 32 ? 1024 : all_cells)) : (n_dec < 128 ? 256 : (n_dec < 256 ? 512 : 1024));
      R = (all_cells + cpi_want - 1) / cpi_want;
      R = std::max(1, std::min(R, (all_cells + min_cpi - 1) / min_cpi));
    }
    int cpi = ((all_cells + R - 1) / R + 63) / 64 * 64;
    if (ctx->knobs.cells_parts <= 0 && R > 1) cpi = (cpi + 255) / 256 * 256;
    R = (all_cells + cpi - 1) / cpi;
    if (R > 1 && (a.n > kCounterCap || (size_t)a.n * kMaxEll * a.n_s * kMaxParts > ctx->partial_doubles)) {
      R = 1;
      cpi = (all_cells + 63) / 64 * 64;
    }
    a.parts = R;
    a.cells_per_item = cpi;
    a.fuse = want_fuse ? 1 : 0;
    const bool tail = a.fuse || R > 1;
    const CellsPlan plc = make_cells_plan(a.n_mu, a.n_x, a.n_s, a.uni_n, nlr, a.n_beta_r, a.uni_lut_n, layout, cpi, tail ? N : 0, n_sva);
    const size_t lds_c = (size_t)plc.total * sizeof(double);
    if (lds_c > 160 * 1024) return fail(ctx, VK_E_ARG, "tables need %zu bytes of LDS (> 160 KiB)", lds_c);
    a.image = get_image(ctx, a, 1, nlr, layout, plc.image_end, sva);
    const long long items_c = a.n * R;
    const int grid_c = (int)items_c;                                       // one item per workgroup, always (vk_kernel_cells.h)
    if (fused) *fused = a.fuse != 0;
    switch (nlr) {
      case 1: return launch_cells_nl<1>(ctx, a, grid_c, lds_c);
      case 2: return launch_cells_nl<2>(ctx, a, grid_c, lds_c);
      case 3: return launch_cells_nl<3>(ctx, a, grid_c, lds_c);
    }
  }
  if (kais || sva_disp) fast = false;   // grids the cells kernel cannot take (n_mu < 64): the generic kernel
  ctx->last_kernel = fast ? "vk_theory_fast_kernel" : "vk_theory_kernel";
  if (fast) {
    const long long groups = (a.n_s + a.sbins_per_item - 1) / a.sbins_per_item;
    if (a.parts > kMaxParts) a.parts = kMaxParts;
    if (a.parts > 1 && ((size_t)a.n * a.n_s * kMaxParts * kMaxEll > ctx->partial_doubles || a.n > kCounterCap)) a.parts = 1;
    const bool need_counters = groups * a.parts > 1;
    a.fuse = want_fuse && (!need_counters || a.n <= kCounterCap) ? 1 : 0;   // counters[point] exists for point < kCounterCap only
    const bool tail = a.fuse || a.parts > 1;
    const FastPlan plf = make_fast_plan(a.n_mu, a.n_x, a.uni_n, nlr, a.n_beta_r, a.uni_lut_n, layout, tail ? N : 0, n_sva);
    const size_t lds = (size_t)plf.total * sizeof(double);
    if (lds > 160 * 1024) return fail(ctx, VK_E_ARG, "tables need %zu bytes of LDS (> 160 KiB)", lds);
    a.image = get_image(ctx, a, 0, nlr, layout, plf.image_end, sva);
    const long long items = a.n * groups * a.parts;
    const int grid = (int)items;                                            // one item per workgroup, always (vk_kernel_fast.h)
    // Hand-off by polling instead of the completion counters (vk_common.h: kPollEmpty): launches of a few points whose
    // workgroups are all resident at once - one point per call and the mailbox server's launches.  Same partial sums, added in the same order: the results do not change by a bit.
    // (resident at once: the kernel's launch bounds give the streaming instantiations three workgroups per CU, the others two,
    // if their LDS fits as often)
    const long long per_cu = std::min<long long>((a.rsd == VK_RSD_STREAMING && !sva && !a.from_data) ? 3 : 2, (160 * 1024) / (long long)(lds ? lds : 1));
    if (a.parts > 1 && ctx->d_poll && !ctx->knobs.no_poll && (size_t)a.n * a.n_s * kMaxParts * kMaxEll <= ctx->poll_doubles &&
        a.n <= kPollPoints) {
      // The context's reservation grows on demand, as far as the process's budget and the device-wide ledger allow (released in
      // vk_destroy) - asked for only by a launch that WOULD poll with it (the rule first: a context whose launches can never
      // poll takes nothing from the 63 waiters of the device), and after a refusal only once the ledger has moved (somebody
      // returned a reservation, a slot changed hands) or every kPollRetryLaunches launches: a refused context must not walk the
      // ledger on every launch of a 20 us path.
      if (ctx->poll_reserved < a.n && vk_poll_rule(a.n, a.parts, items, (int32_t)per_cu, ctx->n_cu, (int32_t)a.n)) {
        int led_status = vkl::kUnavailable;
        vkl::Ledger* led = vkl::for_device(ctx->bus, &led_status);
        // no ledger to be had (no /dev/shm): the process budget alone; a ledger that is there but not ours to trust, or full: no polling
        const bool may_ask = led_status == vkl::kOpened || led_status == vkl::kUnavailable;
        const uint32_t gen = vkl::generation(led);
        const bool refused_before = ctx->poll_refused_want > 0 && ctx->poll_refused_want <= a.n;
        if (may_ask && (!refused_before || gen != ctx->poll_refused_gen || ++ctx->poll_refused_launches >= kPollRetryLaunches)) {
          const int grant = vkl::grant(led, &g_poll_reserved, ctx->poll_reserved, (int)a.n);
          ctx->poll_reserved += grant;
          ctx->poll_refused_want = grant > 0 ? 0 : (int)a.n;
          ctx->poll_refused_gen = gen;
          ctx->poll_refused_launches = 0;
        }
      }
      if (vk_poll_rule(a.n, a.parts, items, (int32_t)per_cu, ctx->n_cu, ctx->poll_reserved)) {
        a.poll = 1;
        a.partial = ctx->d_poll;
        ctx->last_polled = true;
      }
    }
    if (fused) *fused = a.fuse != 0;
    switch (nlr) {
      case 1: return launch_fast_nl<1>(ctx, a, grid, lds);
      case 2: return launch_fast_nl<2>(ctx, a, grid, lds);
      case 3: return launch_fast_nl<3>(ctx, a, grid, lds);
    }
    return fail(ctx, VK_E_ARG, "bad number of real-space multipoles %d", nlr);
  }
  a.parts = 1;
  const size_t lds = (size_t)make_plan(a.n_mu, a.n_x, a.n_ell, a.sv.n_int, a.vr.n_int, a.xi.n_int, nlr, a.n_beta_r).total *
                     sizeof(double);
  if (lds > 160 * 1024) return fail(ctx, VK_E_ARG, "tables need %zu bytes of LDS (> 160 KiB)", lds);
  const long long groups = (a.n_s + a.sbins_per_item - 1) / a.sbins_per_item;
  const long long items = a.n * groups;
  const int grid = (int)(items < cap ? items : cap);
  switch (a.rsd) {
    case VK_RSD_STREAMING: return launch_generic<VK_RSD_STREAMING>(ctx, a, nlr, grid, lds);
    case VK_RSD_DISPERSION: return launch_generic<VK_RSD_DISPERSION>(ctx, a, nlr, grid, lds);
    case VK_RSD_KAISER: return launch_generic<VK_RSD_KAISER>(ctx, a, nlr, grid, lds);
    case VK_RSD_EUCLID: return launch_generic<VK_RSD_EUCLID>(ctx, a, nlr, grid, lds);
  }
  return fail(ctx, VK_E_ARG, "unknown rsd_model %d", a.rsd);
}

void fill_like_args(const vk_ctx* ctx, const vk_eval_opts* o, const double* d_params, const double* d_theory, long long n,
                    double* d_lnl, double* d_chi2, LikeArgs* a) {
  *a = LikeArgs{};
  a->params = d_params;
  a->theory = d_theory;
  a->n = n;
  a->N = ctx->N;
  a->n_beta_d = ctx->n_beta_d;
  a->beta_d = ctx->d_beta_d;
  a->data = ctx->d_data;
  a->n_beta_c = ctx->n_beta_c;
  a->beta_c = ctx->d_beta_c;
  a->prec = ctx->d_prec;
  a->grids_in_lds = ctx->grids_in_lds;
  a->tri = ctx->d_tri;
  a->logdet = ctx->d_logdet;
  a->eig = ctx->d_eig;
  a->like_form = o->like_form;
  a->nmocks = o->nmocks;
  a->nparams = o->nparams;
  a->lnl = d_lnl;
  a->chi2 = d_chi2;
}

int launch_like(vk_ctx* ctx, const LikeArgs& a) {
  const long long n = a.n;
  if (n <= 0) return VK_OK;
  const long long cap = 16LL * ctx->n_cu;
  // small batches: one workgroup per point (the wave-per-point kernels below need >= 1024 points to fill the chip; a
  // single point took 27 us in one wave against the ~3 us of 256 threads)
  const size_t lds_wide = (size_t)like_lds_doubles(ctx->N) * sizeof(double);
  const bool wide = ctx->knobs.like_wide >= 0 ? ctx->knobs.like_wide == 1 : n <= 2048;
  if (wide && lds_wide <= 160 * 1024) {
    return launch_on_stream(ctx, vk_like_wide_kernel, (int)n, lds_wide, a);      // one point per workgroup
  }
  // fixed covariance: 8 points per wave share the loads of the precision matrix (LDS: 4 waves x 8 x N doubles)
  constexpr int kTile = 8;
  const size_t lds_tiled = (size_t)kWaves * kTile * ctx->N * sizeof(double);
  if (ctx->n_beta_c == 0 && n >= 4 * kTile * kWaves && lds_tiled <= 64 * 1024 && !ctx->knobs.like_untiled) {
    const long long tiles = (n + kTile - 1) / kTile;
    const long long blocks = (tiles + kWaves - 1) / kWaves;
    const int grid = (int)(blocks < cap ? blocks : cap);
    return launch_on_stream(ctx, vk_like_tiled_kernel<kTile>, grid, lds_tiled, a);
  }
  const long long blocks = (n + kWaves - 1) / kWaves;
  const int grid = (int)(blocks < cap ? blocks : cap);
  const size_t lds = (size_t)kWaves * ctx->N * sizeof(double);
  if (lds > 160 * 1024) return fail(ctx, VK_E_ARG, "data vector of %d bins needs %zu bytes of LDS (> 160 KiB)", ctx->N, lds);
  return launch_on_stream(ctx, vk_like_kernel, grid, lds, a);
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

}  // namespace

// ---- the internal entry points the host-compiled units share with this file (vk_host.h) -----------------------------------
void vkh::sync_knobs(vk_ctx* ctx) {
  if (ctx->knob_gen != g_knob_gen.load(std::memory_order_relaxed)) load_knobs(ctx);
}

int vkh::fail(vk_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

int vkh::check_opts(vk_ctx* ctx, const vk_eval_opts* o) {
  if (!o) return fail(ctx, VK_E_ARG, "opts is NULL");
  if (o->rsd_model < VK_RSD_STREAMING || o->rsd_model > VK_RSD_EUCLID)
    return fail(ctx, VK_E_ARG, "unknown rsd_model %d", o->rsd_model);
  if (o->niter < 0 || o->niter > 64) return fail(ctx, VK_E_ARG, "niter must be in 0..64");
  if (o->like_form < VK_LIKE_GAUSSIAN || o->like_form > VK_LIKE_PERCIVAL)
    return fail(ctx, VK_E_ARG, "unknown likelihood form %d", o->like_form);
  return VK_OK;
}


extern "C" {

int vk_abi_version(void) { return VK_ABI_VERSION; }

#ifdef VK_PHASES
// profiling build only: wall_clock64() marks (100 MHz) of the last point-major launch, [4096][16]
int vk_debug_read_stamps(long long* out) {
  if (!g_stamps) return VK_E_ARG;
  return hipMemcpy(out, g_stamps, 4096 * 16 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? VK_OK : VK_E_HIP;
}
#endif

void vk_knobs_refresh(void) { g_knob_gen.fetch_add(1, std::memory_order_relaxed); }

int32_t vk_poll_device_reserved(const vk_ctx* ctx, int32_t* others, int32_t* mine) {
  if (!ctx) return VK_E_ARG;
  vkl::Ledger* led = vkl::for_device(ctx->bus, nullptr);
  if (others) *others = led ? vkl::others(led) : -1;
  if (mine) *mine = g_poll_reserved.load(std::memory_order_relaxed);
  return led ? VK_OK : VK_E_ARG;
}

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

int vk_last_fused(const vk_ctx* ctx) { return ctx && ctx->last_fused ? 1 : 0; }

int vk_last_polled(const vk_ctx* ctx) { return ctx && ctx->last_polled ? 1 : 0; }

static int check_pp(const vk_pp* p, const char* name, std::string* err) {
  char buf[256];
  if (p->n_int < 1 || !p->knots || !p->coef || p->lead < 0 || p->lead > 1 || p->lead >= p->n_int + (p->inv_h != 0 ? 0 : 1)) {
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
  {
    // the host-compiled units (vk_walk.cpp, vk_serve.cpp, vk_rccl.cpp) read the context's fields: one layout in every unit
    size_t off[3] = {0, 0, 0};
    const size_t sz[3] = {vkh::ctx_layout_walk(&off[0]), vkh::ctx_layout_serve(&off[1]), vkh::ctx_layout_rccl(&off[2])};
    for (int u = 0; u < 3; ++u)
      if (sz[u] != sizeof(vk_ctx) || off[u] != offsetof(vk_ctx, spin_timeouts))
        return bail("internal: the translation units of this library disagree about the context's layout (a broken build)");
  }
  if (t->n_s < 1 || t->n_mu < 2 || t->n_x < 3 || t->n_ell < 1 || t->n_ell > kMaxEll || t->n_ell_r < 1 ||
      t->n_ell_r > kMaxEll)
    return bail("bad grid sizes (need n_s>=1, n_mu>=2, n_x>=3, 1<=n_ell<=3, 1<=n_ell_r<=3)");
  if (!t->s || !t->mu || !t->w_ell || !t->x || !t->w_x) return bail("grid arrays missing");
  for (int k = 0; k < t->n_x; ++k) {
    // a group of equal weights is closed by the non-zero high word of its weight (vk_common.h: load_node): zero weights are
    // dropped below, anything else must be a normal number
    if (!std::isfinite(t->x[k]) || !std::isfinite(t->w_x[k]) || (t->w_x[k] != 0.0 && std::fabs(t->w_x[k]) < std::numeric_limits<double>::min()))
      return bail("velocity nodes x and quadrature weights w_x must be finite (weights: zero or normal numbers)");
  }
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
  {
    char busid[64] = {0};
    if (hipDeviceGetPCIBusId(busid, (int)sizeof busid, device) != hipSuccess) {
      (void)hipGetLastError();
      snprintf(busid, sizeof busid, "device%d", device);
    }
    ctx->bus = busid;
  }
  if ((rc = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
    return hip_bail(rc, "hipStreamCreate");
  for (auto& evt : ctx->ev)
    if ((rc = hipEventCreate(&evt)) != hipSuccess) return hip_bail(rc, "hipEventCreate");

  load_knobs(ctx);
  ctx->n_s = t->n_s; ctx->n_mu = t->n_mu; ctx->n_x = t->n_x; ctx->n_ell = t->n_ell; ctx->n_ell_r = t->n_ell_r;
  ctx->n_beta_r = t->n_beta_r; ctx->N = N; ctx->iaH = t->iaH; ctx->template_sigma8 = t->template_sigma8;
  ctx->n_beta_d = t->data ? t->n_beta_d : 0;
  ctx->n_beta_c = t->data ? t->n_beta_c : 0;

  {
    // the fast theory kernels need the unified tables: either the uniform-lattice form (uniform, commensurate knot
    // sets sharing the r grid between xi and V) or the union-grid form (any knots) located through a look-up table
    // an anisotropic sigma_v(r, mu) template rides along as bicubic patches when its r grid is uniform, the unified tables are in
    // lattice form and the patches fit beside them in LDS (48 KB)
    const bool sva_fits = t->sv_n_mu == 0 || (t->sv.inv_h > 0 && t->sv.lead == 0 && t->uni_lut_n == 0 &&
                                              (size_t)t->sv.n_int * (t->sv_n_mu - 1) * 16 + t->sv_n_mu <= 6144);
    bool ok = (!t->vr_beta_dep || t->uni_vb) && sva_fits && t->uni_n > 0 && t->uni_sv_v && t->uni_xi && t->uni_xic &&
              t->xi.knots[0] > t->vr.knots[0] && t->sv.knots[0] >= t->vr.knots[0];
    if (ok && t->uni_lut_n > 0) {
      ok = t->uni_lut && t->uni_knots && t->uni_lut_n <= 4097 && t->uni_n < 4096 && t->uni_lut_inv_g > 0 &&
           t->uni_knots[0] == t->vr.knots[0];
      for (int q = 0; q < t->uni_n && ok; ++q) ok = t->uni_knots[q + 1] > t->uni_knots[q];
      for (int c = 0; c < t->uni_lut_n && ok; ++c) ok = t->uni_lut[c] < t->uni_n;
      // the kernels index lut[(int)(u * inv_g)] for every u in [knots[0], knots[n]) and look one and two records ahead of
      // the entry they read: the table must cover the whole clamp range, start at a non-negative radius and each cell's
      // entry must be the interval of the cell's left edge (a cell may hold at most two further knots)
      if (ok) {
        ok = t->uni_knots[0] >= 0.0 && (long long)(t->uni_knots[t->uni_n] * t->uni_lut_inv_g) < (long long)t->uni_lut_n;
        for (int c = 0; c < t->uni_lut_n && ok; ++c) {
          const double left = c / t->uni_lut_inv_g, right = (c + 1) / t->uni_lut_inv_g;
          const int q = t->uni_lut[c];
          if (left >= t->uni_knots[t->uni_n]) break;                       // cells beyond the last knot are never read
          ok = q == 0 ? t->uni_knots[1] > left : t->uni_knots[q] <= left * (1.0 + 1e-12);
          if (ok && q + 3 <= t->uni_n) ok = t->uni_knots[q + 3] >= right * (1.0 - 1e-12);   // at most two knots inside the cell
        }
        if (!ok) return bail("union-grid look-up table is inconsistent with its knots (uni_lut / uni_lut_inv_g / uni_knots)");
      }
    } else if (ok) {
      ok = t->xi.inv_h > 0 && t->xi.lead == 0 && t->sv.inv_h > 0 && t->sv.lead == 0 && t->vr.inv_h > 0 &&
           t->vr.lead == 1 && t->vr.n_int == t->xi.n_int + 1 && t->uni_inv_h > 0 && t->uni_u0 <= t->vr.knots[0];
    }
    ctx->fast_ok = ok;
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
  // {kExpScale x_k, w_k} pairs for the kernels that read the velocity nodes through the scalar cache; one pad pair
  std::vector<double> xw_scaled(2 * ((size_t)t->n_x + 1), 0.0);
  for (int k = 0; k < t->n_x; ++k) {
    xw_scaled[2 * k] = t->x[k] * vkm::kExpScale;
    xw_scaled[2 * k + 1] = t->w_x[k];
    ctx->xw_max = std::max(ctx->xw_max, std::fabs(xw_scaled[2 * k]));
  }
  const size_t o_xws = up.add(xw_scaled.data(), xw_scaled.size());
  // the same nodes in groups of equal weight, first occurrence first, original order inside a group; the weight travels with
  // the group's last node (the lanes and cells kernels multiply a group's sum by it once)
  std::vector<double> xgw;
  {
    std::vector<char> taken(t->n_x, 0);
    for (int k0 = 0; k0 < t->n_x; ++k0) {
      if (taken[k0] || t->w_x[k0] == 0.0) continue;       // a node of weight zero adds nothing to the integral
      for (int k = k0; k < t->n_x; ++k)
        if (!taken[k] && t->w_x[k] == t->w_x[k0]) {
          taken[k] = 1;
          xgw.push_back(t->x[k] * vkm::kExpScale);
          xgw.push_back(0.0);
        }
      xgw.back() = t->w_x[k0];
    }
    ctx->n_xg = (int)(xgw.size() / 2);
    xgw.resize(xgw.size() + 2, 0.0);      // pad pair
  }
  const size_t o_xgw = up.add(xgw.data(), xgw.size());
  const size_t o_br = t->n_beta_r > 0 ? up.add(t->beta_r, t->n_beta_r) : 0;
  const size_t xi_coef_n = t->n_beta_r > 0 ? (size_t)t->n_ell_r * (t->n_beta_r - 1) * t->xi.n_int * 16
                                           : (size_t)t->n_ell_r * t->xi.n_int * 4;
  const size_t o_xik = up.add(t->xi.knots, t->xi.n_int + 1), o_xic = up.add(t->xi.coef, xi_coef_n);
  const size_t vr_coef_n = t->vr_beta_dep ? (size_t)2 * (t->n_beta_r - 1) * t->vr.n_int * 16 : (size_t)kVrVars * t->vr.n_int * 4;
  const size_t o_vrk = up.add(t->vr.knots, t->vr.n_int + 1), o_vrc = up.add(t->vr.coef, vr_coef_n);
  const bool have_vr_emp = t->vr_beta_dep && t->vr_emp;
  const size_t o_vre = have_vr_emp ? up.add(t->vr_emp, (size_t)3 * (t->n_beta_r - 1) * t->vr.n_int * 28) : 0;
  const size_t o_svk = up.add(t->sv.knots, t->sv.n_int + 1),
               o_svc = up.add(t->sv.coef, t->sv_n_mu ? 4 : (size_t)t->sv.n_int * 4);   // 1-D coefficients unused with sv2d
  size_t o_svmu = 0, o_sv2d = 0, o_sva = 0;
  if (t->sv_n_mu) {
    o_svmu = up.add(t->sv_mu, t->sv_n_mu);
    o_sv2d = up.add(t->sv2d, (size_t)t->sv.n_int * (t->sv_n_mu - 1) * 16);
    if (ctx->fast_ok) {
      // the same patches for the fast kernels: powers of the r interval's local coordinate in [0, 1) instead of (u - knot),
      // followed by the mu knots - one contiguous block that the kernels copy into LDS
      const int nm = t->sv_n_mu - 1;
      const double h = 1.0 / t->sv.inv_h;
      std::vector<double> blk((size_t)t->sv.n_int * nm * 16 + t->sv_n_mu);
      for (size_t e = 0; e < (size_t)t->sv.n_int * nm * 16; ++e) {
        const int pw = (int)((e >> 2) & 3);                        // [i][j][p][q]: coefficient of (u - x_i)^p (mu - mu_j)^q
        blk[e] = t->sv2d[e] * (pw == 0 ? 1.0 : pw == 1 ? h : pw == 2 ? h * h : h * h * h);
      }
      for (int k = 0; k < t->sv_n_mu; ++k) blk[(size_t)t->sv.n_int * nm * 16 + k] = t->sv_mu[k];
      o_sva = up.add(blk.data(), blk.size());
      ctx->sva_doubles = (int)blk.size();
    }
  }
  size_t o_usv = 0, o_uxi = 0, o_uxc = 0, o_ulut = 0, o_uk = 0, o_uvb = 0, o_uv2 = 0, o_uda = 0, o_uge = 0, o_udab = 0, o_uempb = 0;
  const bool have_lut = t->uni_n > 0 && t->uni_lut_n > 0 && t->uni_lut && t->uni_knots;
  if (have_lut) {
    std::vector<double> packed(((size_t)t->uni_lut_n + 3) / 4, 0.0);        // u16 cells travel inside the double arena
    memcpy(packed.data(), t->uni_lut, (size_t)t->uni_lut_n * sizeof(uint16_t));
    o_ulut = up.add(packed.data(), packed.size());
    o_uk = up.add(t->uni_knots, (size_t)t->uni_n + 1);
  }
  if (t->uni_n > 0 && t->uni_sv_v && t->uni_xi && t->uni_xic) {
    o_usv = up.add(t->uni_sv_v, (size_t)t->uni_n * 8);
    o_uxi = up.add(t->uni_xi, t->n_beta_r > 0 ? (size_t)t->n_ell_r * (t->n_beta_r - 1) * t->uni_n * 16
                                              : (size_t)t->n_ell_r * t->uni_n * 4);
    o_uxc = up.add(t->uni_xic, t->n_beta_r > 0 ? (size_t)t->n_ell_r * (t->n_beta_r - 1) * t->uni_n * 16
                                               : (size_t)t->n_ell_r * t->uni_n * 4);
    // What the fast kernels read is 1 + xi^r as a polynomial in (2 mu_r)^2 (vk_kernel_fast.h: uni_point): the "+1" of
    // ccf_model.py:690 goes into the constant coefficient of the l = 0 set (for beta polynomials: the beta-constant one), and
    // the sets of mu_r^2 and mu_r^4 are divided by 4 and 16 - exact rescalings, done here so that vk_tables keeps its meaning
    {
      const size_t per_l = t->n_beta_r > 0 ? (size_t)(t->n_beta_r - 1) * t->uni_n * 16 : (size_t)t->uni_n * 4;
      const size_t step = t->n_beta_r > 0 ? 16 : 4;          // doubles per (interval): [4 powers of tau] x [4 powers of d beta] or [4]
      for (size_t o : {o_uxi, o_uxc}) {
        double* x0 = up.host.data() + o;
        for (size_t e = 0; e < per_l; e += step) x0[e] += 1.0;
      }
      double* xc = up.host.data() + o_uxc;
      for (int l = 1; l < t->n_ell_r; ++l) {
        const double f = l == 1 ? 0.25 : 0.0625;
        for (size_t e = 0; e < per_l; ++e) xc[(size_t)l * per_l + e] *= f;
      }
    }
    if (t->vr_beta_dep && t->uni_vb) o_uvb = up.add(t->uni_vb, (size_t)(t->n_beta_r - 1) * t->uni_n * 16);
    if (!t->vr_beta_dep && t->uni_v2) o_uv2 = up.add(t->uni_v2, (size_t)t->uni_n * 4);
    if (!t->vr_beta_dep && t->uni_da) o_uda = up.add(t->uni_da, (size_t)t->uni_n * 4);
    if (!t->vr_beta_dep && t->uni_ge) o_uge = up.add(t->uni_ge, (size_t)t->uni_n * 8);
    if (t->vr_beta_dep && t->uni_dab) o_udab = up.add(t->uni_dab, (size_t)(t->n_beta_r - 1) * t->uni_n * 16);
    if (t->vr_beta_dep && t->uni_empb) o_uempb = up.add(t->uni_empb, (size_t)3 * (t->n_beta_r - 1) * t->uni_n * 28);
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
  // the quadratic form of every precision slice folded onto its upper triangle and stored by circular diagonals, M/2 + 1 rows
  // of M entries (+ 2 of padding), M = N rounded up to even (a zero row and column for an odd N): what the fused tail and the
  // wide K2 read (vk_kernel_like.h: LikePrefetch), half the bytes of the slice
  size_t o_tri = 0;
  const bool have_tri = t->data != nullptr;
  if (have_tri) {
    const int slices = t->n_beta_c > 0 ? t->n_beta_c : 1;
    const int M = (N + 1) & ~1, half = M / 2, W = M + 2;
    const size_t slice = like_slice_doubles(N);
    std::vector<double> tri((size_t)slices * slice, 0.0);
    for (int sl = 0; sl < slices; ++sl) {
      const double* P = t->prec + (size_t)sl * N * N;
      auto fold = [&](int i, int j) {
        if (i >= N || j >= N) return 0.0;
        return i == j ? P[(size_t)i * N + i] : P[(size_t)i * N + j] + P[(size_t)j * N + i];
      };
      for (int k = 0; k <= half; ++k) {
        double* row = tri.data() + (size_t)sl * slice + (size_t)k * W;
        const int n_i = k == half ? half : M;            // the diagonal half-way round pairs every i with i + M/2 once
        for (int i = 0; i < n_i; ++i) row[i] = fold(i, (i + k) % M);
      }
    }
    o_tri = up.add(tri.data(), tri.size());
  }
  const size_t bytes = up.host.size() * sizeof(double);
  if ((rc = hipMalloc((void**)&ctx->d_tables, bytes)) != hipSuccess) return hip_bail(rc, "hipMalloc(tables)");
  if ((rc = hipMemcpy(ctx->d_tables, up.host.data(), bytes, hipMemcpyHostToDevice)) != hipSuccess)
    return hip_bail(rc, "hipMemcpy(tables)");
  const double* base = ctx->d_tables;
  ctx->d_x1 = base + o_x1; ctx->d_w1 = base + o_x1 + 1;
  ctx->d_xws = base + o_xws;
  ctx->d_xgw = base + o_xgw;
  ctx->d_s = base + o_s; ctx->d_mu = base + o_mu; ctx->d_w = base + o_w; ctx->d_x = base + o_x; ctx->d_wx = base + o_wx;
  ctx->d_beta_r = t->n_beta_r > 0 ? base + o_br : nullptr;
  auto view = [&](const vk_pp& p, size_t ok, size_t oc) {
    PPView v;
    v.n_int = p.n_int; v.lead = p.lead; v.inv_h = p.inv_h; v.knots = base + ok; v.coef = base + oc;
    return v;
  };
  ctx->xi = view(t->xi, o_xik, o_xic);
  ctx->vr = view(t->vr, o_vrk, o_vrc);
  ctx->d_vr_emp = have_vr_emp ? base + o_vre : nullptr;
  ctx->sv = view(t->sv, o_svk, o_svc);
  if (t->uni_n > 0 && t->uni_sv_v && t->uni_xi && t->uni_xic) {
    ctx->uni_n = t->uni_n;
    ctx->uni_u0 = t->uni_u0;
    ctx->uni_inv_h = t->uni_inv_h;
    ctx->d_uni_sv_v = base + o_usv;
    ctx->d_uni_xi = base + o_uxi;
    ctx->d_uni_xic = base + o_uxc;
    if (t->vr_beta_dep && t->uni_vb) ctx->d_uni_vb = base + o_uvb;
    if (!t->vr_beta_dep && t->uni_v2) ctx->d_uni_v2 = base + o_uv2;
    if (!t->vr_beta_dep && t->uni_da) ctx->d_uni_da = base + o_uda;
    if (!t->vr_beta_dep && t->uni_ge) ctx->d_uni_ge = base + o_uge;
    if (t->vr_beta_dep && t->uni_dab) ctx->d_uni_dab = base + o_udab;
    if (t->vr_beta_dep && t->uni_empb) ctx->d_uni_empb = base + o_uempb;
    if (have_lut) {
      ctx->uni_lut_n = t->uni_lut_n;
      ctx->uni_lut_inv_g = t->uni_lut_inv_g;
      ctx->d_uni_lut = reinterpret_cast<const unsigned short*>(base + o_ulut);
      ctx->d_uni_knots = base + o_uk;
    }
  }
  ctx->sv_n_mu = t->sv_n_mu;
  ctx->sv_mu_inv_h = t->sv_mu_inv_h;
  if (t->sv_n_mu) {
    ctx->d_sv_mu = base + o_svmu;
    ctx->d_sv2d = base + o_sv2d;
    if (ctx->sva_doubles) ctx->d_sva = base + o_sva;
  }
  {
    // batch-independent staging tables + bookkeeping of the split / fused launches
    const size_t n_stage = (size_t)t->n_mu * kMuRec;
    ctx->partial_doubles = (size_t)kPartialPoints * t->n_s * kMaxParts * kMaxEll;
    const size_t n_exp = vkm::ExpCfg<0>::kDoubles + vkm::ExpCfg<1>::kDoubles;
    ctx->poll_doubles = (size_t)kPollPoints * t->n_s * kMaxParts * kMaxEll;
    const size_t counter_doubles = (kCounterCap * sizeof(unsigned) + 7) / 8 + 2;
    const size_t aux_doubles = n_exp + n_stage + ctx->partial_doubles + counter_doubles + ctx->poll_doubles;
    if ((rc = hipMalloc((void**)&ctx->d_aux, aux_doubles * sizeof(double))) != hipSuccess) return hip_bail(rc, "hipMalloc(aux)");
    if ((rc = hipMemsetAsync(ctx->d_aux, 0, aux_doubles * sizeof(double), ctx->stream)) != hipSuccess)
      return hip_bail(rc, "hipMemset(aux)");
    // the polling area (behind the counters) starts out empty; without the pinned word for its alarm there is no polling
    {
      void* dev = nullptr;
      if (hipHostMalloc((void**)&ctx->h_poll_failed, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
          hipHostGetDevicePointer(&dev, ctx->h_poll_failed, 0) == hipSuccess) {
        *ctx->h_poll_failed = 0;
        ctx->d_poll_failed = static_cast<int*>(dev);
        ctx->d_poll = ctx->d_aux + (aux_doubles - ctx->poll_doubles);
        if ((rc = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ctx->d_poll), (int)(kPollEmpty & 0xffffffffu), ctx->poll_doubles * 2,
                                    ctx->stream)) != hipSuccess)
          return hip_bail(rc, "hipMemsetD32(polling area)");
      } else {
        (void)hipGetLastError();
        if (ctx->h_poll_failed) (void)hipHostFree(ctx->h_poll_failed);
        ctx->h_poll_failed = nullptr;
      }
    }
    double* exp_tab = ctx->d_aux;
    double* exp_tab_rep = exp_tab + vkm::ExpCfg<0>::kDoubles;
    double* stage_mu = exp_tab + n_exp;
    ctx->d_partial = stage_mu + n_stage;
    ctx->d_counters = reinterpret_cast<unsigned*>(ctx->d_partial + ctx->partial_doubles);
    hipLaunchKernelGGL(vk_init_stage_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->d_mu, ctx->d_w, t->n_mu, t->n_ell, exp_tab,
                       exp_tab_rep, stage_mu);
    if ((rc = hipGetLastError()) != hipSuccess) return hip_bail(rc, "vk_init_stage_kernel");
    if ((rc = hipStreamSynchronize(ctx->stream)) != hipSuccess) return hip_bail(rc, "vk_init_stage_kernel");
    ctx->d_exp_tab = exp_tab;
    ctx->d_exp_tab_rep = exp_tab_rep;
    ctx->d_stage_mu = stage_mu;
    for (int l = 0; l < t->n_ell; ++l) {
      double ws = 0.0;
      for (int i = 0; i < t->n_mu; ++i) ws += t->w_ell[(size_t)l * t->n_mu + i];
      ctx->wsum[l] = ws;
    }
  }
  if (t->data) {
    ctx->d_beta_d = t->n_beta_d > 0 ? base + o_bd : nullptr;
    ctx->d_data = base + o_data;
    ctx->d_beta_c = t->n_beta_c > 0 ? base + o_bc : nullptr;
    ctx->d_prec = base + o_prec;
    ctx->d_tri = have_tri ? base + o_tri : nullptr;
    auto same_grid = [&](const double* g, int n) { return n > 0 && n == t->n_beta_r && memcmp(g, t->beta_r, (size_t)n * sizeof(double)) == 0; };
    ctx->grids_in_lds = (same_grid(t->beta_d, t->n_beta_d) ? 1 : 0) | (same_grid(t->beta_c, t->n_beta_c) ? 2 : 0);
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
  if (ctx->poll_reserved)              // (nothing of it is in flight any more)
    vkl::release(vkl::for_device(ctx->bus, nullptr), &g_poll_reserved, ctx->poll_reserved);
  ctx->poll_reserved = 0;
  drop_graphs(ctx);
  if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
  if (ctx->h_zc) (void)hipHostFree(ctx->h_zc);
  if (ctx->d_tables) (void)hipFree(ctx->d_tables);
  if (ctx->d_aux) (void)hipFree(ctx->d_aux);
  if (ctx->h_poll_failed) (void)hipHostFree(ctx->h_poll_failed);
  if (ctx->h_comm) (void)hipHostFree(ctx->h_comm);
  if (ctx->d_comm) (void)hipFree(ctx->d_comm);
  if (ctx->ev_comm) (void)hipEventDestroy(ctx->ev_comm);
  for (auto& kv : ctx->images) (void)hipFree(kv.second);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  for (auto& evt : ctx->ev)
    if (evt) (void)hipEventDestroy(evt);
  if (ctx->ev_joint) (void)hipEventDestroy(ctx->ev_joint);
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
  if (ctx->h_poll_failed && *ctx->h_poll_failed)
    return fail(ctx, VK_E_HIP, "a workgroup waited %.0f s for partial sums that never arrived; the context is unusable", (double)kPollTicks * 1e-8);
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
  sync_knobs(ctx);
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
  a.want_theory = (!want_like || ctx->theory_wanted) ? 1 : 0;   // with lnL / chi2 requested the workspace is scratch (victor_hip.h)
  if (ctx->inline_params && n == 1) {       // single point from host buffers: the row travels in the kernel arguments (TheoryArgs::row0)
    memcpy(a.row0, ctx->inline_params, VK_NPAR * sizeof(double));
    a.inline_row = 1;
  }
  a.n_s = ctx->n_s; a.n_mu = ctx->n_mu; a.n_ell = ctx->n_ell;
  a.s = ctx->d_s; a.mu = ctx->d_mu; a.w_ell = ctx->d_w;
  a.stage_mu = ctx->d_stage_mu;
  for (int l = 0; l < 3; ++l) a.wsum[l] = ctx->wsum[l];
  // One launch addresses its work items with 32 bits (n * n_s * kMaxParts < 2^31): larger batches - 6.7 M points at 40 s bins -
  // are cut into chunks of whole 65536-point blocks here, each with its own pair of launches on the same stream.  Points are
  // independent and every kernel's arithmetic per point is independent of its neighbours, so the results are bit-identical to
  // calls the caller chunks himself.
  const long long chunk_max = std::max<long long>(65536, ((1LL << 31) / ((long long)ctx->n_s * kMaxParts) - 1) & ~65535LL);
  if (n > chunk_max && ctx->timing) return fail(ctx, VK_E_ARG, "kernel timing is per launch pair: disable it for batches above %lld points", chunk_max);
  const bool timed = ctx->timing && want_like;
  if (timed) {
    harvest_timing(ctx);
    VK_HIP(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  }
  for (long long off = 0; off < n; off += chunk_max) {
    const long long m = std::min<long long>(chunk_max, n - off);
    a.params = d_params + off * VK_NPAR;
    a.n = m;
    a.out = d_theory_ws + off * ctx->N;
    LikeArgs la;
    if (want_like)
      fill_like_args(ctx, opts, a.params, a.out, m, d_lnl ? d_lnl + off : nullptr, d_chi2 ? d_chi2 + off : nullptr, &la);
    bool fused = false;
    rc = launch_theory(ctx, a, nlr, want_like ? &la : nullptr, &fused);
    if (rc) return rc;
    ctx->last_fused = fused;
    if (timed) VK_HIP(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    if (want_like && !fused) {
      rc = launch_like(ctx, la);
      if (rc) return rc;
    }
  }
  if (timed) {
    VK_HIP(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    ctx->pending = true;
  }
  return VK_OK;
}

size_t vk_joint_workspace_doubles(vk_ctx* const* ctxs, int32_t n_ctx, int64_t n) {
  if (!ctxs || n_ctx < 1 || n < 0) return 0;
  int n_max = 0;
  for (int q = 0; q < n_ctx; ++q)
    if (ctxs[q] && ctxs[q]->N > n_max) n_max = ctxs[q]->N;
  return (size_t)n_ctx * (size_t)n * (n_max + 2);
}

int vk_joint_eval_device_async(vk_ctx* const* ctxs, int32_t n_ctx, const vk_eval_opts* opts, const double* d_params, int64_t n,
                               double* d_lnl, double* d_chi2, double* d_ws) {
  if (!ctxs || n_ctx < 1 || !ctxs[0]) return VK_E_ARG;
  vk_ctx* lead = ctxs[0];
  for (int q = 0; q < n_ctx; ++q) {
    if (!ctxs[q]) return fail(lead, VK_E_ARG, "context %d is NULL", q);
    if (ctxs[q]->device != lead->device) return fail(lead, VK_E_ARG, "joint fit: every context must live on the same device");
    if (!ctxs[q]->d_data) return fail(lead, VK_E_ARG, "joint fit: context %d was created without a data vector", q);
  }
  if (n < 0 || (n > 0 && (!d_params || !d_ws || !(d_lnl || d_chi2)))) return fail(lead, VK_E_ARG, "bad device buffers");
  if (n == 0) return VK_OK;
  VK_HIP(lead, hipSetDevice(lead->device));
  int n_max = 0;
  for (int q = 0; q < n_ctx; ++q) n_max = std::max(n_max, ctxs[q]->N);
  const long long block_stride = (long long)n * (n_max + 2);    // per block: lnl[n] | chi2[n] | theory workspace [n][N]
  for (int q = 0; q < n_ctx; ++q)
    if (!ctxs[q]->ev_joint) VK_HIP(lead, hipEventCreateWithFlags(&ctxs[q]->ev_joint, hipEventDisableTiming));
  // the blocks run on their own streams, all behind what the lead stream has enqueued so far (e.g. the parameter upload)
  VK_HIP(lead, hipEventRecord(lead->ev_joint, lead->stream));
  for (int q = 0; q < n_ctx; ++q) {
    vk_ctx* c = ctxs[q];
    if (q > 0) VK_HIP(lead, hipStreamWaitEvent(c->stream, lead->ev_joint, 0));
    double* blk = d_ws + q * block_stride;
    c->depth_mult = n_ctx;       // the launches overlap on the GPU: the lanes kernel's depth rule sees all of them
    const int rc = vk_eval_batch_device_async(c, opts, d_params, n, blk, blk + n, blk + 2 * n);
    c->depth_mult = 1;
    if (rc) {
      if (c != lead) lead->err = c->err;
      return rc;
    }
  }
  for (int q = 1; q < n_ctx; ++q) {
    VK_HIP(lead, hipEventRecord(ctxs[q]->ev_joint, ctxs[q]->stream));
    VK_HIP(lead, hipStreamWaitEvent(lead->stream, ctxs[q]->ev_joint, 0));
  }
  hipLaunchKernelGGL(vk_joint_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, lead->stream, d_ws, (long long)n, n_ctx,
                     block_stride, d_lnl, d_chi2);
  VK_HIP(lead, hipGetLastError());
  return VK_OK;
}

// The launch-bound small-batch case of vk_eval_batch: one hipGraph launch per call (see vk_ctx::graphs).
// Returns 1 when it handled the call, 0 when the caller should take the eager path, < 0 on error.
static int eval_batch_graph(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n, double* lnl,
                            double* chi2, double* d_par, double* d_th, double* d_lnl, double* d_chi) {
  if (ctx->graphs_off || n > kGraphMaxN || ctx->timing || !(lnl || chi2) || ctx->knobs.no_graph || ctx->knobs.mapping ||
      ctx->knobs.force_generic || ctx->knobs.no_fuse || ctx->split_as_single)
    return 0;
  std::string key(reinterpret_cast<const char*>(opts), sizeof *opts);
  key.append(reinterpret_cast<const char*>(&n), sizeof n);
  key.push_back(lnl ? 1 : 0);
  key.push_back(chi2 ? 1 : 0);
  auto hit = ctx->graphs.find(key);
  if (hit == ctx->graphs.end()) {
    int& seen = ctx->graph_seen[key];
    if (seen < 0) return 0;                            // capturing this shape failed before: stay eager
    if (seen++ == 0) return 0;                         // first use of this shape: eager (also warms attributes)
    if (ctx->graphs.size() >= 32) drop_graphs(ctx);
    if (!ctx->h_pin) {
      if (hipHostMalloc((void**)&ctx->h_pin, (size_t)kGraphMaxN * (VK_NPAR + 2) * sizeof(double), hipHostMallocDefault) !=
          hipSuccess) {
        ctx->graphs_off = true;
        return 0;
      }
    }
    double* h_in = ctx->h_pin;
    double* h_out = ctx->h_pin + (size_t)kGraphMaxN * VK_NPAR;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool ok = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed) == hipSuccess;
    if (ok) {
      ok = hipMemcpyAsync(d_par, h_in, (size_t)n * VK_NPAR * sizeof(double), hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
      ok = ok && vk_eval_batch_device_async(ctx, opts, d_par, n, d_lnl, d_chi, d_th) == VK_OK;
      // d_lnl and d_chi are adjacent in the scratch layout: one copy brings both back
      ok = ok && hipMemcpyAsync(h_out, d_lnl, (size_t)2 * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
      ok = (hipStreamEndCapture(ctx->stream, &graph) == hipSuccess) && ok && graph;
    }
    if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
    if (!ok) {
      (void)hipGetLastError();
      ctx->graph_seen[key] = -1;                       // e.g. an unsupported option combination: the eager path reports it
      return 0;
    }
    ctx->graph_kernel[key] = ctx->last_kernel;
    hit = ctx->graphs.emplace(key, exec).first;
  }
  double* h_in = ctx->h_pin;
  double* h_out = ctx->h_pin + (size_t)kGraphMaxN * VK_NPAR;
  memcpy(h_in, params, (size_t)n * VK_NPAR * sizeof(double));
  VK_HIP(ctx, hipGraphLaunch(hit->second, ctx->stream));
  VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->last_kernel = ctx->graph_kernel[key];
  if (ctx->h_poll_failed && *ctx->h_poll_failed)      // (set before the results were written, see finish_point)
    return fail(ctx, VK_E_HIP, "a workgroup waited %.0f s for partial sums that never arrived; the context is unusable", (double)kPollTicks * 1e-8);
  if (lnl) memcpy(lnl, h_out, (size_t)n * sizeof(double));
  if (chi2) memcpy(chi2, h_out + n, (size_t)n * sizeof(double));
  return 1;
}

constexpr int64_t kSpinMaxDefault = 256;         // in-place batches up to this size poll for their results (eval_batch_zero_copy)
constexpr uint64_t kSpinSentinel = 0x7ff8dead5ca1ab1eULL;   // a quiet NaN with a payload no arithmetic produces
constexpr int64_t kZeroCopyMaxDefault = 4096;   // host-buffer batches up to this size: parameters read in place, results written in place

// Launch-bound host-buffer batches (the reference calls the likelihood with ONE point, CCFLikelihood.py:32-39): no copies at
// all.  The parameter rows are placed in pinned host memory that the GPU reads in place (96 bytes per point over the link,
// overlapped with the staging of the tables) and lnL / chi2 are written straight back into pinned host memory, so the call
// is one kernel launch and one stream synchronisation instead of a graph of (H2D copy, kernel, D2H copy).  Measured per call,
// config 3, against the captured graph: 1 point 41 -> 30 us, 33: 66 -> 54, 64: 75 -> 63, 256: 172 -> 154, 1024: 493 -> 483,
// 4096: 1855 -> 1819 us; the graph path remains for contexts whose in-place buffers cannot be mapped (VICTOR_HIP_NO_ZERO_COPY).
// The two halves of such a call, separable so that the mailbox server (vk_serve_mailboxes) can have launches of several
// contexts in flight at once.  zc_begin: 1 = launched (zc_finish brings the results), 0 = this batch cannot go in place (the
// caller takes another path), < 0 = error.
// the pinned, device-mapped buffers of the in-place paths (kZeroCopyCap x (VK_NPAR + 2) doubles); false: not available here
static bool ensure_zero_copy(vk_ctx* ctx) {
  if (ctx->knobs.no_zero_copy || ctx->zero_copy_off) return false;
  if (!ctx->h_zc) {
    void* dev = nullptr;
    if (hipHostMalloc((void**)&ctx->h_zc, (size_t)kZeroCopyCap * (VK_NPAR + 2) * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostGetDevicePointer(&dev, ctx->h_zc, 0) != hipSuccess) {
      (void)hipGetLastError();
      ctx->zero_copy_off = true;
      return false;
    }
    ctx->d_zc = static_cast<double*>(dev);
  }
  return true;
}

extern "C++" int vkh::zc_begin(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n, bool want_out, double* d_th) {
  const int64_t zc_max = kZeroCopyMaxDefault;
  if (n > zc_max || ctx->timing || !want_out || !ensure_zero_copy(ctx)) return 0;
  double* h_out = ctx->h_zc + (size_t)kZeroCopyCap * VK_NPAR;
  double* d_out = ctx->d_zc + (size_t)kZeroCopyCap * VK_NPAR;
  memcpy(ctx->h_zc, params, (size_t)n * VK_NPAR * sizeof(double));
  // A handful of points: the host does not wait for the end of the launch (the command processor's end-of-kernel release and
  // completion signal, then the runtime's wake-up) but for the results themselves - the slots are pre-set to a NaN pattern no
  // kernel writes (failed rows report -inf / +inf) and polled in place; each slot is one 8-byte store of the thread that
  // finishes its point.  The launch stays queued on the stream, which orders the next call behind it, and every workgroup
  // has read its parameter row before the last result can appear, so the buffers may be reused at once.  Falls back to
  // the stream synchronisation if nothing arrives within 2 ms (a failed launch reports its error there).  A result that
  // happens to carry the sentinel's bit pattern - only a caller's NaN parameter with exactly that payload, propagated
  // into lnL, can - looks like "not arrived": the call then returns the correct values after the 2 ms timeout.
  const int64_t spin_max = ctx->knobs.spin_max >= 0 ? ctx->knobs.spin_max : kSpinMaxDefault;
  ctx->zc_spin = n <= spin_max && !ctx->spin_off;
  volatile uint64_t* slots = reinterpret_cast<volatile uint64_t*>(h_out);
  if (ctx->zc_spin)
    for (int64_t i = 0; i < 2 * n; ++i) slots[i] = kSpinSentinel;
  ctx->inline_params = (n == 1 && !ctx->knobs.no_inline_row) ? ctx->h_zc : nullptr;
  const int rc = vk_eval_batch_device_async(ctx, opts, ctx->d_zc, n, d_out, d_out + n, d_th);
  ctx->inline_params = nullptr;
  if (rc) return rc;
  ctx->zc_t0 = std::chrono::steady_clock::now();
  return 1;
}

// 1 = the results are in lnl / chi2, 0 = not yet (only with block == false), < 0 = error
extern "C++" int vkh::zc_finish(vk_ctx* ctx, int64_t n, double* lnl, double* chi2, bool block) {
  double* h_out = ctx->h_zc + (size_t)kZeroCopyCap * VK_NPAR;
  volatile uint64_t* slots = reinterpret_cast<volatile uint64_t*>(h_out);
  bool arrived = false;
  if (ctx->zc_spin) {
    for (unsigned it = 1; !arrived; ++it) {
      arrived = true;
      for (int64_t i = 2 * n - 1; i >= 0; --i)
        if (slots[i] == kSpinSentinel) { arrived = false; break; }
      if (arrived) break;
      const bool late = ((it & 255u) == 0 || !block) && std::chrono::steady_clock::now() - ctx->zc_t0 > std::chrono::milliseconds(2);
      if (!late) {
        if (!block) return 0;
        cpu_relax();
        continue;
      }
      // Nothing after 2 ms.  Either the launch is simply still running (a busy GPU: other contexts, other processes, a slow
      // kernel variant) - then a non-blocking caller is told "not yet" and asks again, a blocking one waits for the stream -
      // or the stream HAS completed and the slots still hold the sentinel: the stores do not reach this memory before the
      // launch ends on this system (or a result carries the sentinel's bit pattern).  Only the latter counts against polling.
      const hipError_t q = hipStreamQuery(ctx->stream);
      if (q == hipErrorNotReady) {
        (void)hipGetLastError();
        if (!block) return 0;
        VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
      }
      // the stream is complete: whatever the launch wrote is visible now (the last results may have landed between the poll
      // above and the query) - look once more before calling it a time-out of the polling itself
      arrived = true;
      for (int64_t i = 2 * n - 1; i >= 0; --i)
        if (slots[i] == kSpinSentinel) { arrived = false; break; }
      break;                                  // arrived: a slow launch or the race - nothing to count; not arrived: counted below
    }
  } else if (!block && hipStreamQuery(ctx->stream) == hipErrorNotReady) {
    (void)hipGetLastError();
    return 0;
  }
  if (!arrived) {
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // three small launches in a row had completed before their results showed in the polled slots (the first call of a context,
    // which also builds the LDS image, may): the stores evidently do not reach this memory before the launch ends - stop polling it
    if (ctx->zc_spin && n <= 64 && ++ctx->spin_timeouts >= 3) ctx->spin_off = true;
  } else {
    ctx->spin_timeouts = 0;
  }
  if (ctx->h_poll_failed && *ctx->h_poll_failed)      // (set before the results were written, see finish_point)
    return fail(ctx, VK_E_HIP, "a workgroup waited %.0f s for partial sums that never arrived; the context is unusable", (double)kPollTicks * 1e-8);
  if (lnl) memcpy(lnl, h_out, (size_t)n * sizeof(double));
  if (chi2) memcpy(chi2, h_out + n, (size_t)n * sizeof(double));
  return 1;
}

static int eval_batch_zero_copy(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n, double* lnl,
                                double* chi2, double* d_th) {
  const int rc = zc_begin(ctx, opts, params, n, lnl || chi2, d_th);
  if (rc != 1) return rc;
  return zc_finish(ctx, n, lnl, chi2, true);
}

// scratch of the host-buffer entry points: parameters | theory workspace | lnl | chi2 (vk_host.h: HostScratch)
extern "C++" int vkh::host_scratch(vk_ctx* ctx, int64_t n, HostScratch* sc) {
  // small batches share one scratch layout sized for kGraphMaxN so that captured graphs stay valid across sizes
  const int64_t n_lay = n <= kGraphMaxN ? kGraphMaxN : n;
  const int rc = ensure_scratch(ctx, (size_t)n_lay * (VK_NPAR + ctx->N + 2) * sizeof(double));
  if (rc) return rc;
  sc->d_par = ctx->d_scratch;
  sc->d_th = sc->d_par + (size_t)n_lay * VK_NPAR;
  sc->d_lnl = sc->d_th + (size_t)n_lay * ctx->N;
  sc->d_chi = sc->d_lnl + n;
  return VK_OK;
}

int vk_eval_batch(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n, double* lnl, double* chi2,
                  double* theory) {
  if (!ctx) return VK_E_ARG;
  sync_knobs(ctx);
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (n < 0 || (n > 0 && !params)) return fail(ctx, VK_E_ARG, "params is NULL");
  if (n == 0) return VK_OK;
  if (ctx->begun_n != 0) return fail(ctx, VK_E_ARG, "a batch begun with vk_eval_batch_begin is awaiting vk_eval_batch_finish on this context");
  if ((lnl || chi2) && !ctx->d_data) return fail(ctx, VK_E_ARG, "context was created without a data vector");
  VK_HIP(ctx, hipSetDevice(ctx->device));
  const size_t nb_par = (size_t)n * VK_NPAR * sizeof(double);
  const size_t nb_th = (size_t)n * ctx->N * sizeof(double);
  const size_t nb_out = (size_t)n * sizeof(double);
  HostScratch sc;
  rc = host_scratch(ctx, n, &sc);
  if (rc) return rc;
  double *d_par = sc.d_par, *d_th = sc.d_th, *d_lnl = sc.d_lnl, *d_chi = sc.d_chi;
  if (!theory) {
    rc = eval_batch_zero_copy(ctx, opts, params, n, lnl, chi2, d_th);
    if (rc < 0) return rc;
    if (rc == 1) return VK_OK;
    rc = eval_batch_graph(ctx, opts, params, n, lnl, chi2, d_par, d_th, d_lnl, d_chi);
    if (rc < 0) return rc;
    if (rc == 1) return VK_OK;
  }
  VK_HIP(ctx, hipMemcpyAsync(d_par, params, nb_par, hipMemcpyHostToDevice, ctx->stream));
  ctx->theory_wanted = theory != nullptr;
  rc = vk_eval_batch_device_async(ctx, opts, d_par, n, lnl ? d_lnl : nullptr, chi2 ? d_chi : nullptr, d_th);
  ctx->theory_wanted = false;
  if (rc) return rc;
  if (lnl) VK_HIP(ctx, hipMemcpyAsync(lnl, d_lnl, nb_out, hipMemcpyDeviceToHost, ctx->stream));
  if (chi2) VK_HIP(ctx, hipMemcpyAsync(chi2, d_chi, nb_out, hipMemcpyDeviceToHost, ctx->stream));
  if (theory) VK_HIP(ctx, hipMemcpyAsync(theory, d_th, nb_th, hipMemcpyDeviceToHost, ctx->stream));
  return vk_sync(ctx);
}

// ---- a small host-buffer batch in two halves: enqueue now, collect later (include/victor_hip.h) ------------------------
int vk_eval_batch_begin(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n) {
  if (!ctx) return VK_E_ARG;
  sync_knobs(ctx);
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (ctx->begun_n != 0) return fail(ctx, VK_E_ARG, "vk_eval_batch_begin: the previous batch of this context has not been collected");
  if (n < 1 || n > kZeroCopyCap || !params) return fail(ctx, VK_E_ARG, "vk_eval_batch_begin: 1 <= n <= %lld rows", (long long)kZeroCopyCap);
  if (!ctx->d_data) return fail(ctx, VK_E_ARG, "context was created without a data vector");
  VK_HIP(ctx, hipSetDevice(ctx->device));
  HostScratch sc;
  rc = host_scratch(ctx, n, &sc);
  if (rc) return rc;
  rc = zc_begin(ctx, opts, params, n, true, sc.d_th);
  if (rc < 0) return rc;
  if (rc == 0) {          // no in-place buffers on this system: evaluate now, hand the results over in finish
    ctx->begun_sync.resize((size_t)2 * n);
    rc = vk_eval_batch(ctx, opts, params, n, ctx->begun_sync.data(), ctx->begun_sync.data() + n, nullptr);
    if (rc) return rc;
    ctx->begun_n = -n;
    return VK_OK;
  }
  ctx->begun_n = n;
  return VK_OK;
}

int vk_eval_batch_finish(vk_ctx* ctx, double* lnl, double* chi2) {
  if (!ctx) return VK_E_ARG;
  const int64_t n = ctx->begun_n;
  if (n == 0) return fail(ctx, VK_E_ARG, "vk_eval_batch_finish: nothing was begun on this context");
  ctx->begun_n = 0;
  if (n < 0) {
    if (lnl) memcpy(lnl, ctx->begun_sync.data(), (size_t)(-n) * sizeof(double));
    if (chi2) memcpy(chi2, ctx->begun_sync.data() + (-n), (size_t)(-n) * sizeof(double));
    return VK_OK;
  }
  VK_HIP(ctx, hipSetDevice(ctx->device));
  const int rc = zc_finish(ctx, n, lnl, chi2, true);
  return rc < 0 ? rc : VK_OK;
}

static int general_grid(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n, const double* s,
                        int32_t n_s, const double* mu, int32_t n_mu, const double* w_ell, int32_t n_ell, double* out,
                        bool project) {
  if (!ctx) return VK_E_ARG;
  sync_knobs(ctx);
  int rc = check_opts(ctx, opts);
  if (rc) return rc;
  if (n < 0 || n_s < 1 || n_mu < 2 || !params || !s || !mu || !out) return fail(ctx, VK_E_ARG, "bad arguments");
  if (ctx->begun_n != 0) return fail(ctx, VK_E_ARG, "a batch begun with vk_eval_batch_begin is awaiting vk_eval_batch_finish on this context");
  if (project && (n_ell < 1 || n_ell > kMaxEll || !w_ell)) return fail(ctx, VK_E_ARG, "n_ell must be 1..3");
  if (n == 0) return VK_OK;
  VK_HIP(ctx, hipSetDevice(ctx->device));
  const int ne = project ? n_ell : 1;
  const size_t out_n = project ? (size_t)n * n_ell * n_s : (size_t)n * n_mu * n_s;
  const size_t grid_n = (size_t)n_s + n_mu + (size_t)ne * n_mu + 4;
  const size_t total = ((size_t)n * VK_NPAR + grid_n + out_n) * sizeof(double);
  // A call that fits into the context's pinned, device-mapped buffers - theory_xi or theory_multipoles for a point or a few,
  // what a notebook asks for - goes through them: rows and grids are gathered there and reach the device in ONE copy from pinned
  // memory (instead of four from pageable memory), and the kernel stores its results straight into the pinned buffer (instead
  // of a download into pageable memory behind it).  theory_xi, one point on a 40 x 100 grid: DESIGN.md section 5.
  const size_t in_n = (size_t)n * VK_NPAR + grid_n;
  const bool in_place = total + 32 <= (size_t)kZeroCopyCap * (VK_NPAR + 2) * sizeof(double) && !ctx->timing && ensure_zero_copy(ctx);
  rc = ensure_scratch(ctx, in_place ? (in_n + 2) * sizeof(double) : total);
  if (rc) return rc;
  double* d_par = ctx->d_scratch;
  double* d_s = d_par + (size_t)n * VK_NPAR;
  double* d_mu = d_s + n_s;
  double* d_w = d_mu + n_mu;
  double* d_out = d_w + (size_t)ne * n_mu;
  d_out = (double*)(((uintptr_t)d_out + 15) & ~(uintptr_t)15);
  double* h_out = nullptr;
  if (in_place) {
    double* h = ctx->h_zc;
    memcpy(h, params, (size_t)n * VK_NPAR * sizeof(double));
    memcpy(h + (d_s - d_par), s, n_s * sizeof(double));
    memcpy(h + (d_mu - d_par), mu, n_mu * sizeof(double));
    if (project) memcpy(h + (d_w - d_par), w_ell, (size_t)n_ell * n_mu * sizeof(double));
    else memset(h + (d_w - d_par), 0, (size_t)n_mu * sizeof(double));
    VK_HIP(ctx, hipMemcpyAsync(d_par, h, ((size_t)(d_w - d_par) + (size_t)ne * n_mu) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const size_t out_off = (in_n + 1) & ~(size_t)1;                      // 16-byte aligned behind the inputs
    h_out = h + out_off;
    d_out = ctx->d_zc + out_off;
  } else {
    VK_HIP(ctx, hipMemcpyAsync(d_par, params, (size_t)n * VK_NPAR * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipMemcpyAsync(d_s, s, n_s * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    VK_HIP(ctx, hipMemcpyAsync(d_mu, mu, n_mu * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (project)
      VK_HIP(ctx, hipMemcpyAsync(d_w, w_ell, (size_t)n_ell * n_mu * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    else
      VK_HIP(ctx, hipMemsetAsync(d_w, 0, (size_t)n_mu * sizeof(double), ctx->stream));
  }
  TheoryArgs a{};
  int nlr = 1;
  rc = theory_args(ctx, opts, &a, &nlr);
  if (rc) return rc;
  a.params = d_par;
  a.n = n;
  a.n_s = n_s; a.n_mu = n_mu; a.n_ell = ne;
  a.s = d_s; a.mu = d_mu; a.w_ell = d_w;
  a.out = d_out;
  a.want_theory = 1;
  a.stage_mu = nullptr;                          // the caller's own (mu, W) grid: staged inside the kernel
  for (int l = 0; l < 3; ++l) {
    a.wsum[l] = 0.0;
    if (project && l < n_ell)
      for (int i = 0; i < n_mu; ++i) a.wsum[l] += w_ell[(size_t)l * n_mu + i];
  }
  a.xi_out = project ? 0 : 1;            // theory_xi: the cells kernel stores every cell, the generic kernel where it cannot go
  rc = launch_theory(ctx, a, nlr, nullptr, nullptr);
  if (rc) return rc;
  if (in_place) {
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, h_out, out_n * sizeof(double));
  } else {
    VK_HIP(ctx, hipMemcpyAsync(out, d_out, out_n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    VK_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (ctx->h_poll_failed && *ctx->h_poll_failed)
    return fail(ctx, VK_E_HIP, "a workgroup waited %.0f s for partial sums that never arrived; the context is unusable", (double)kPollTicks * 1e-8);
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

}  // extern "C"
