// vk_host.h - what the host-side translation units of libvictor_hip.so share: the context, the development knobs and the
// internal entry points that cross files.  No device code: victor_hip.hip (hipcc; the ABI's launch side and every kernel)
// includes it next to the kernel headers, vk_walk.cpp / vk_serve.cpp / vk_rccl.cpp (host compiler) include nothing else of
// the library.  The layout of vk_ctx must be the same in every unit: one definition, no conditional members, standard-library
// members only (hipcc and the host compiler use the same libstdc++); vk_create checks the sizes the units report against each
// other once (ctx_layout_*).
#pragma once

#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "victor_hip.h"
#include "vk_views.h"

using vk::PPView;

// Development-only tuning / A-B knobs from the environment (VICTOR_HIP_*).  They are honoured ONLY when VICTOR_HIP_DEV=1 is set
// as well (tests/ and tools/ set it, victor_amd._native.set_knob does): a variable inherited from somebody's shell must never
// change the kernel mapping or switch the fused path off in a production run.  Read once per context - not once per launch -
// and again after vk_knobs_refresh().
struct Knobs {
  int split_s = 0, split_t = 0;        // VICTOR_HIP_SPLIT "spi,team"
  bool force_generic = false;          // VICTOR_HIP_FORCE_GENERIC
  long long point_cap = 0;             // VICTOR_HIP_POINT_CAP   (workgroups per CU, generic theory kernel; 0 = default)
  int mapping = 0;                     // VICTOR_HIP_MAPPING: 0 auto, 1 point, 2 cells, 3 lanes, -1 unknown name
  bool like_untiled = false;           // VICTOR_HIP_LIKE_UNTILED
  bool no_graph = false;               // VICTOR_HIP_NO_GRAPH
  bool no_fuse = false;                // VICTOR_HIP_NO_FUSE: keep chi2 in its own launch (A/B of the fused path)
  bool no_inline_row = false;          // VICTOR_HIP_NO_INLINE_ROW: single-point host calls read their row from the pinned buffer (A/B)
  long long fuse_max = -1;             // VICTOR_HIP_FUSE_MAX: largest batch whose chi2 is taken inside the theory kernel (-1 = default)
  int split_q = 0;                     // third field of VICTOR_HIP_SPLIT "spi,team,parts": workgroups per (mu, v) plane
  int cells_parts = 0;                 // VICTOR_HIP_CELLS_PARTS: workgroups per point in the cells kernel (0 = choose)
  int like_wide = -1;                  // VICTOR_HIP_LIKE_WIDE: 1 / 0 force the workgroup-per-point chi2 kernel on / off (the parity tests
                                       // reach the wave-per-point and tiled K2 at small batches through it)
  bool no_zero_copy = false;           // VICTOR_HIP_NO_ZERO_COPY: small host-buffer batches through the copy / graph path
  long long spin_max = -1;             // VICTOR_HIP_SPIN_MAX: largest in-place batch whose results are polled for (-1 = default, 0 = never)
  bool no_poll = false;                // VICTOR_HIP_NO_POLL: split single-point launches hand over through the completion counters (A/B)
};

struct vk_ctx {
  int device = -1;
  Knobs knobs;
  unsigned knob_gen = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int n_cu = 256;
  // host copy of sizes
  int n_s = 0, n_mu = 0, n_x = 0, n_ell = 0, n_ell_r = 0, n_beta_r = 0, n_beta_d = 0, n_beta_c = 0, N = 0;
  double iaH = 0, template_sigma8 = 0;
  double* d_tables = nullptr;  // one allocation holding every table
  // device pointers into d_tables
  const double *d_x1 = nullptr, *d_w1 = nullptr;  // single velocity node for the Kaiser-type models
  const double* d_vr_emp = nullptr;               // beta-dependent V2, Ge1, Ge2 (degree 6 in beta), see vk_tables.vr_emp
  const double* d_xws = nullptr;                  // [n_x + 1][2]: {kExpScale x_k, w_k} (point-major fast kernel)
  const double* d_sva = nullptr;                  // anisotropic sigma_v block for the fast kernels (TheoryArgs::sva)
  int sva_doubles = 0;
  const double* d_xgw = nullptr;                  // velocity nodes grouped by quadrature weight (TheoryArgs::xgw)
  int n_xg = 0;                                   // ... how many of them (nodes of weight zero are left out)
  double xw_max = 0.0;                            // max |kExpScale x_k|
  const double *d_s = nullptr, *d_mu = nullptr, *d_w = nullptr, *d_x = nullptr, *d_wx = nullptr, *d_beta_r = nullptr,
               *d_beta_d = nullptr, *d_data = nullptr, *d_beta_c = nullptr, *d_prec = nullptr, *d_tri = nullptr, *d_logdet = nullptr,
               *d_eig = nullptr;
  PPView xi{}, vr{}, sv{};
  bool fast_ok = false;      // tables qualify for vk_theory_fast_kernel
  int matter_lb = 0, vr_beta_dep = 0, matter_vt = 0, sv_n_mu = 0;
  double vt_amp = 0, sv_mu_inv_h = 0;
  const double *d_sv_mu = nullptr, *d_sv2d = nullptr;
  int uni_n = 0;             // unified refined grid (fast kernels need it)
  double uni_u0 = 0, uni_inv_h = 0;
  const double *d_uni_sv_v = nullptr, *d_uni_xi = nullptr, *d_uni_xic = nullptr, *d_uni_vb = nullptr, *d_uni_v2 = nullptr, *d_uni_da = nullptr, *d_uni_ge = nullptr, *d_uni_dab = nullptr, *d_uni_empb = nullptr;
  int uni_lut_n = 0;         // > 0: union-grid form of the unified tables
  double uni_lut_inv_g = 0;
  const unsigned short* d_uni_lut = nullptr;
  const double* d_uni_knots = nullptr;
  // batch-independent staging tables and the bookkeeping of the fused / split launches (one device allocation)
  double* d_aux = nullptr;
  const double* d_exp_tab = nullptr;   // [ExpCfg<0>::kDoubles]
  const double* d_exp_tab_rep = nullptr;   // [ExpCfg<1>::kDoubles]
  const double* d_stage_mu = nullptr;  // [n_mu][kMuRec]
  int grids_in_lds = 0;                    // LikeArgs::grids_in_lds
  const double* inline_params = nullptr;   // set around a single-point host-buffer call: the row goes into the kernel arguments
  bool theory_wanted = false;              // set around a host-buffer call that returns the theory vectors (TheoryArgs::want_theory)
  bool split_as_single = false;            // set around the launches of vk_serve_mailboxes: every point is evaluated with the work
                                           // split of a single-point call, whatever else shares its launch (kServeMaxBatch)
  unsigned* d_counters = nullptr;      // [kCounterCap], zero between launches
  double* d_partial = nullptr;         // [partial_doubles]
  size_t partial_doubles = 0;
  double* d_poll = nullptr;            // [poll_doubles] polling area of the launches that hand over without counters (kPollEmpty
  size_t poll_doubles = 0;             // between launches; vk_common.h), or NULL
  int* h_poll_failed = nullptr;        // pinned, device-mapped word a polling workgroup sets when it gives up; d_poll_failed: the
  int* d_poll_failed = nullptr;        // same word through the device's eyes
  int poll_reserved = 0;               // waiters this context may have resident at once (its share of kPollBudget; vk_poll_grant)
  int poll_refused_want = 0;           // > 0: a reservation of this many waiters was refused ...
  uint32_t poll_refused_gen = 0;       // ... at this generation of the ledger (vkl::generation): asked again only once that has moved
  int poll_refused_launches = 0;       // ... or every kPollRetryLaunches launches (a reservation may have died with its process)
  std::string bus;                     // PCI bus id of the device (the device-wide ledger of reserved waiters is kept per GPU)
  double wsum[3] = {0, 0, 0};
  int depth_mult = 1;                // joint fits: launches of this many contexts share the GPU (vk_joint_eval_device_async)
  hipEvent_t ev_joint = nullptr;
  std::map<int, double*> images;     // LDS images per (kernel kind, real-space multipoles, dispersion tables), built on first use
  std::map<const void*, int> lds_opt_in;   // dynamic LDS above 64 KiB a kernel has been opted in for (launch_on_stream)
  const char* last_kernel = "none";  // theory kernel variant of the most recent launch
  bool last_fused = false;           // ... and whether it took the chi-square as well
  bool last_polled = false;          // ... and whether its split planes were handed over by polling (vk_poll_rule)
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
  int comm_nranks = 0;
  // vk_comm_allgather_host_begin / _finish: pinned host staging [1 + nranks][cap], device buffers likewise, the event behind the download
  double* h_comm = nullptr;
  double* d_comm = nullptr;
  int64_t comm_cap = 0, comm_begun = 0;
  hipEvent_t ev_comm = nullptr;
  // small host-buffer batches are launch-bound: (H2D, theory kernel, likelihood kernel, D2H) is captured once per
  // (n, options) into a hipGraph over pinned staging buffers and replayed with a single launch
  double* h_pin = nullptr;                      // pinned: params[kGraphMaxN][VK_NPAR] | lnl, chi2 [2 kGraphMaxN]
  std::map<std::string, hipGraphExec_t> graphs;  // key: n + option bytes + requested outputs
  std::map<std::string, int> graph_seen;         // a key is captured on its second use (the first one runs eagerly)
  std::map<std::string, const char*> graph_kernel;
  bool graphs_off = false;
  double* h_zc = nullptr;                       // pinned, device-mapped: params[kZeroCopyCap][VK_NPAR] | lnl, chi2 [2 kZeroCopyCap]
  double* d_zc = nullptr;                       // the same memory through the device's eyes
  bool zero_copy_off = false;
  bool spin_off = false;               // results did not become visible to polling on this system (eval_batch_zero_copy)
  int64_t begun_n = 0;                 // vk_eval_batch_begin: rows of the batch awaiting vk_eval_batch_finish (< 0: evaluated already)
  std::vector<double> begun_sync;      // ... their results in that case
  bool zc_spin = false;                // the in-place launch in flight polls for its results (zc_begin / zc_finish)
  std::chrono::steady_clock::time_point zc_t0;
  int spin_timeouts = 0;
};

constexpr int64_t kGraphMaxN = 4096;
constexpr int64_t kZeroCopyCap = 4096;     // capacity of the in-place buffers (points)
constexpr long long kCounterCap = 16384;   // points per launch that may share work between workgroups (completion counters)
constexpr long long kPartialPoints = 2048;  // batches up to this many points may split a point's work over workgroups (partial sums)
constexpr int kServeMaxBatch = 32;          // requests one launch of the mailbox server carries (vk_ctx::split_as_single)

// ---- internal entry points that cross translation units (defined in victor_hip.hip unless noted) ---------------------------
namespace vkh __attribute__((visibility("hidden"))) {       // (internal: not part of the library's exported symbols)

int fail(vk_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
int check_opts(vk_ctx* ctx, const vk_eval_opts* o);
void sync_knobs(vk_ctx* ctx);

// scratch of the host-buffer entry points: parameters | theory workspace | lnl | chi2
struct HostScratch { double *d_par, *d_th, *d_lnl, *d_chi; };
int host_scratch(vk_ctx* ctx, int64_t n, HostScratch* sc);

// The two halves of a launch-bound host-buffer batch on the in-place (pinned, device-mapped) buffers.  zc_begin: 1 = launched
// (zc_finish brings the results), 0 = this batch cannot go in place (the caller takes another path), < 0 = error.  zc_finish:
// 1 = the results are in lnl / chi2, 0 = not yet (only with block == false), < 0 = error.
int zc_begin(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n, bool want_out, double* d_th);
int zc_finish(vk_ctx* ctx, int64_t n, double* lnl, double* chi2, bool block);

// sizeof(vk_ctx) and the offset of its last member as each host-compiled unit sees them (vk_create compares them with its own)
size_t ctx_layout_walk(size_t* last_offset);
size_t ctx_layout_serve(size_t* last_offset);
size_t ctx_layout_rccl(size_t* last_offset);

// spin-wait hint of the polling loops
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  asm volatile("" ::: "memory");
#endif
}

}  // namespace vkh

#define VK_HIP(ctx, call)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) return vkh::fail((ctx), VK_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)
