/*
 * victor_hip.h - C ABI of libvictor_hip.so, the MI355X (gfx950) implementation of
 * victor's per-step likelihood hot path.
 *
 * The reference (seshnadathur/victor, pure Python) has no native boundary of its
 * own; the replaceable interface is its Python class API.  Each entry point below
 * names the reference method whose work it performs for a whole BATCH of parameter
 * points (paths relative to the reference root):
 *
 *   vk_eval_batch        CCFFit.log_likelihood        victor/ccf_fit.py:356-483
 *                        CCFFit.chi_squared           victor/ccf_fit.py:325-354
 *                        CCFModel.theory_multipole_vector  victor/ccf_model.py:829-860
 *   vk_theory_batch      CCFModel.theory_multipoles   victor/ccf_model.py:791-827
 *                        (+ utils.multipoles_from_fn  victor/utils.py:9-58)
 *   vk_xi_smu_batch      CCFModel.theory_xi           victor/ccf_model.py:538-789
 *   vk_create            CCFModel.__init__ / CCFFit.__init__ table set-up
 *                        victor/ccf_model.py:33-97, victor/ccf_fit.py:15-42
 *                        (the host has already turned every spline into explicit
 *                        piecewise-cubic coefficient tables; vk_create uploads them)
 *
 * Conventions
 *   - plain C, no C++/torch types; all arrays are contiguous row-major doubles
 *   - the caller owns every host buffer; the context owns device memory and one
 *     HIP stream; a context is bound to one device and is not thread-safe
 *   - return value 0 = success, negative = VK_E_* code; vk_last_error() gives text
 *   - calls are synchronous unless the name ends in _async (then use vk_sync)
 *   - there is no CPU fallback anywhere in the library
 */
#ifndef VICTOR_HIP_H
#define VICTOR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VK_ABI_VERSION 18

/* error codes */
#define VK_OK 0
#define VK_E_ARG (-1)      /* bad argument / unsupported option combination */
#define VK_E_HIP (-2)      /* a HIP runtime call failed */
#define VK_E_NODEVICE (-3) /* no usable GPU */
#define VK_E_RCCL (-4)     /* an RCCL call failed */

/* one row of the parameter batch: VK_NPAR doubles (ccf_model.py:583-613,638,695-696) */
#define VK_NPAR 12
#define VK_P_FSIGMA8 0 /* params['fsigma8']                                   */
#define VK_P_SIGMAV 1  /* params.get('sigma_v', 380)                          */
#define VK_P_APERP 2   /* alpha_perp  (from aperp, or epsilon*apar)           */
#define VK_P_APAR 3    /* alpha_par   (from apar, or alpha*epsilon^(-2/3))    */
#define VK_P_EPSILON 4 /* epsilon     (given, or aperp/apar)                  */
#define VK_P_BETA 5    /* reconstruction beta (0.40 dummy when irrelevant)    */
#define VK_P_ASTAR 6   /* params.get('astar', 1)                              */
#define VK_P_M 7       /* Kaiser nuisance M (default 1)                       */
#define VK_P_Q 8       /* Kaiser nuisance Q (default 1)                       */
#define VK_P_BIAS 9    /* params.get('bias', model['bias']) - linear_bias matter model (ccf_model.py:359,430) */
#define VK_P_AV 10     /* params.get('Av', 0) - empirical velocity correction (ccf_model.py:453) */
#define VK_P_SPARE 11

/* vk_eval_opts.rsd_model (ccf_model.py:646-787) */
#define VK_RSD_STREAMING 0
#define VK_RSD_DISPERSION 1
#define VK_RSD_KAISER 2
#define VK_RSD_EUCLID 3

/* vk_tables.matter_model (ccf_model.py:71,358-372) */
#define VK_MATTER_TEMPLATE 0
#define VK_MATTER_LINEAR_BIAS 1
#define VK_MATTER_VELOCITY_TEMPLATE 2 /* velocity_pdf.mean.model == 'template' (ccf_model.py:439-443,483-490): the
                                         tables hold the template v_r(r) itself, amplitude = fsigma8 * vt_amp */

/* vk_eval_opts.like_form (ccf_fit.py:455-473) */
#define VK_LIKE_GAUSSIAN 0
#define VK_LIKE_SELLENTIN 1
#define VK_LIKE_HARTLAP 2
#define VK_LIKE_PERCIVAL 3

/*
 * A clamped piecewise-cubic table  f(u) = sum_p coef[i][p] * (u - knot_i)^p  on interval i.
 * Evaluation clamps u to [knots[0], knots[n_int]] first (FITPACK ext=3 / box clamp).
 * Interval search: the first `lead` intervals are irregular, the rest start at knots[lead].
 * inv_h > 0: those knots are exactly uniform with spacing 1/inv_h; inv_h < 0: nearly uniform (every knot within
 * 0.4 spacings of the uniform position, mean spacing 1/|inv_h|; the estimate is corrected against the knots);
 * inv_h == 0: arbitrary knots, binary search.
 */
typedef struct vk_pp {
  int32_t n_int;        /* number of intervals; knots has n_int+1 entries      */
  int32_t lead;         /* 0 or 1 leading irregular interval                   */
  double inv_h;         /* see above                                           */
  const double* knots;  /* [n_int+1]                                           */
  const double* coef;   /* [n_var][n_int][4]; n_var is table specific          */
} vk_pp;

typedef struct vk_tables {
  /* ---- redshift-space grid of the data vector and its projection ---------- */
  int32_t n_s;          /* s bins of the data vector (ccf_fit.py:90)           */
  int32_t n_mu;         /* mu nodes, 100 (ccf_model.py:819,822)                */
  int32_t n_x;          /* velocity nodes, 50 (ccf_model.py:570)               */
  int32_t n_ell;        /* multipoles in the data vector (ccf_fit.py:92)       */
  const double* s;      /* [n_s]                                               */
  const double* mu;     /* [n_mu]                                              */
  const double* w_ell;  /* [n_ell][n_mu] projection weights (utils.py:45-56 composed with
                           the cubic interp2d of ccf_model.py:824)             */
  const double* x;      /* [n_x] v/sigma_v nodes, linspace(-6,6,n_x)           */
  const double* w_x;    /* [n_x] Simpson weights * dx / sqrt(2 pi) (ccf_model.py:690).  For the reference's 50 (even) nodes
                           the caller chooses which SciPy's default `simps` rule the weights restate: SciPy >= 1.11
                           (1/3,4/3,2/3,...,4/3 - 1/12, 1/3 + 2/3, 5/12) or SciPy < 1.11, even='avg'
                           (5/12, 13/12, 1, ..., 1, 13/12, 5/12); victor_amd.tables.simpson_weights builds both */

  /* ---- real-space CCF multipoles xi^r_l(r) (ccf_model.py:615-621) ---------- */
  int32_t n_ell_r;      /* real-space multipoles available: 1..3 (l = 0,2,4)   */
  int32_t n_beta_r;     /* 0: tables fixed; else nodes of the reconstruction-beta grid */
  const double* beta_r; /* [n_beta_r]                                          */
  vk_pp xi;             /* fixed: coef[n_ell_r][n_int][4]
                           beta-dependent: coef[n_ell_r][n_beta_r-1][n_int][4][4], the last
                           index being the power of (beta - beta_r[k]) (PCHIP in beta,
                           ccf_model.py:323-326, composed with the not-a-knot spline in r) */

  /* ---- matter profile -> velocity profile (ccf_model.py:421-459,625-636) --- */
  int32_t matter_model; /* VK_MATTER_TEMPLATE or VK_MATTER_LINEAR_BIAS: selects the per-point amplitude
                           (growth term and powers of 1/bias, ccf_model.py:426-435)                */
  int32_t vr_beta_dep;  /* 0: velocity tables fixed; 1: they depend on the reconstruction beta through
                           xi^r_0 (linear_bias with reconstruction): PCHIP-in-beta form on beta_r  */
  vk_pp vr;             /* splined on r_ext = [0.01, r...]:
                           fixed:  coef[5][n_int][4] = V1 = r*Delta, Da = delta - 2 Delta/3,
                                   V2 = r*Delta*delta, Ge1, Ge2 (numerical-gradient tables of the
                                   empirical_corr branch, ccf_model.py:455-459; /3 folded in)
                           beta-dependent: coef[2][n_beta_r-1][n_int][4][4] = V1, Da              */
  const double* vr_emp; /* beta-dependent tables only, may be NULL (then empirical_corr is refused):
                           [3][n_beta_r-1][vr.n_int][4][7] = V2, Ge1, Ge2 as polynomials of degree 6 in
                           (beta - beta_r[k]) - V2 and Ge2 are products of two PCHIP cubics              */
  double vt_amp;        /* VK_MATTER_VELOCITY_TEMPLATE only: template_hubble_ratio * (1+z_sim)/(1+z_eff) /
                           template_fsigma8 (ccf_model.py:442-443); apar cancels against aH_true       */
  /* ---- velocity dispersion template (ccf_model.py:654-655) ----------------- */
  vk_pp sv;             /* isotropic template: coef[1][n_int][4], normalised sigma_v(r) shape;
                           anisotropic template: knots = r_for_sv only (coef unused, see sv2d)          */
  int32_t sv_n_mu;      /* 0: isotropic.  else number of mu nodes of the sigma_v(r, mu) template        */
  double sv_mu_inv_h;   /* 1/spacing of the mu nodes if uniform, else 0                                 */
  const double* sv_mu;  /* [sv_n_mu] mu nodes                                                           */
  const double* sv2d;   /* [sv.n_int][sv_n_mu-1][4][4] bicubic patches: coefficient of du^p dmu^q of the
                           tensor-product not-a-knot spline (RectBivariateSpline, ccf_model.py:654)     */

  /* ---- optional unified grid: one interval index for sigma_v, V1 and every xi^r_l -------------------- */
  /* When the r grid and the sigma_v grid are uniform and commensurate, the host re-expresses all tables on
   * their common refinement (u0 + q/inv_h, q < uni_n) in interval units; the fast theory kernels need it.
   * The grid starts at or below the first knot of vr (u = 0.01): refined intervals below a table's first knot hold
   * its boundary value, the interval that contains 0.01 holds the leading V cubic (the kernels never evaluate
   * below u = 0.01: V is clamped there, ccf_model.py:625 with ext=3, and every other table is constant).       */
  int32_t uni_n;        /* 0: not available (only the generic theory kernel is used)                    */
  double uni_u0;        /* left end of the refined grid, <= vr.knots[0]                                  */
  double uni_inv_h;     /* 1/spacing of the refined grid                                                 */
  const double* uni_sv_v; /* [uni_n][2][4]: sigma_v shape and V1 = r*Delta, coefficients of tau^p        */
  const double* uni_xi; /* fixed: [n_ell_r][uni_n][4]; beta-dependent: [n_ell_r][n_beta_r-1][uni_n][4][4]
                           (last index = power of beta - beta_r[k])                                       */
  const double* uni_xic;/* same shape, the Legendre sum regrouped in powers of m = mu_r^2 (n_ell_r >= 2 only):
                           sum_l xi_l P_l(mu_r) = A + m B + m^2 C with A = xi_0 - xi_2/2 + 3 xi_4/8,
                           B = 3 xi_2/2 - 15 xi_4/4, C = 35 xi_4/8; used when the anisotropic sum is asked for  */
  const double* uni_vb; /* vr_beta_dep only (else NULL): V1 on the unified grid as beta polynomials,
                           [n_beta_r-1][uni_n][4][4]; the V half of uni_sv_v is then rebuilt per point            */
  const double* uni_v2; /* fixed velocity tables only (else NULL): V2 = r*Delta*delta on the unified grid, [uni_n][4];
                           with empirical_corr the V half of the records becomes V1 + Av V2 per point            */
  const double* uni_da; /* fixed velocity tables only (else NULL): Da = delta - 2 Delta/3 on the unified grid,
                           [uni_n][4]; lets the dispersion model run on the fast kernels                        */
  const double* uni_ge; /* fixed velocity tables only (else NULL): the numerical-gradient tables Ge1, Ge2 of the
                           empirical_corr branch on the unified grid, [2][uni_n][4]: the dispersion model then uses
                           Ge1 + Av Ge2 in place of Da (ccf_model.py:455-459)                                    */
  const double* uni_dab;/* vr_beta_dep only (else NULL): Da on the unified grid as beta polynomials,
                           [n_beta_r-1][uni_n][4][4]: the dispersion model on the fast kernels with linear_bias on a
                           reconstructed real-space ccf (ccf_model.py:358-370, 658-671)                           */
  const double* uni_empb;/* vr_beta_dep only (else NULL): V2, Ge1, Ge2 of the empirical_corr branch on the unified grid,
                           degree 6 in beta (vr_emp refined): [3][n_beta_r-1][uni_n][4][7]                        */
  /* Union-grid form of the same tables for knots that are not uniform or not commensurate: uni_n intervals between
   * the sorted distinct knots uni_knots[0..uni_n] of vr (uni_knots[0] = vr.knots[0] = 0.01), xi and sv; the
   * coefficient arrays above are then in units of each interval's own width.  A uniform look-up table of
   * uni_lut_n - 1 cells over [0, uni_knots[uni_n]) plus one guard cell locates the interval: every cell holds at most
   * TWO interior knots and uni_lut[c] is the interval that contains the cell's left edge (guard cell: the last
   * interval); the kernels add (u >= next knot) + (u >= the one after).  uni_u0 / uni_inv_h are unused here.     */
  int32_t uni_lut_n;      /* 0: uniform-lattice form.  > 0: union-grid form, entries of the table (<= 4097)     */
  double uni_lut_inv_g;   /* cells per unit length: (uni_lut_n - 1) / uni_knots[uni_n]                          */
  const uint16_t* uni_lut;/* [uni_lut_n]                                                                        */
  const double* uni_knots;/* [uni_n + 1]                                                                        */

  double iaH;           /* (1+z)/(100 E(z)) (ccf_model.py:43-45)               */
  double template_sigma8; /* ccf_model.py:432-435                              */

  /* ---- data vector (ccf_fit.py:166-193,306-323) ---------------------------- */
  int32_t n_beta_d;     /* 0: fixed data vector; else beta nodes               */
  const double* beta_d; /* [n_beta_d]                                          */
  const double* data;   /* fixed: [N]; else PCHIP pieces [n_beta_d-1][N][4], N = n_ell*n_s */

  /* ---- covariance / precision (ccf_fit.py:195-260, 445-453) ---------------- */
  int32_t n_beta_c;     /* 0: fixed; else beta nodes of the covariance stack   */
  const double* beta_c; /* [n_beta_c]                                          */
  const double* prec;   /* fixed: [N][N]; else [n_beta_c][N][N] (inverse of each slice) */
  const double* logdet; /* [n_beta_c] log det of each covariance slice (NULL if fixed) */
  const double* eig;    /* [n_beta_c][N] generalised eigenvalues of (cov[last], cov[k]) so that
                           logdet((1-t) cov[k] + t cov[last]) = logdet[k] + sum log(1-t+t*eig) */
} vk_tables;

typedef struct vk_eval_opts {
  int32_t rsd_model;         /* VK_RSD_*                                        */
  int32_t assume_isotropic;  /* use xi^r_0 only (ccf_model.py:681-687)          */
  int32_t rescale_from_ap;   /* 0: c = astar; 1: c = AP integral (ccf_model.py:606-611) */
  int32_t like_form;         /* VK_LIKE_*                                       */
  double nmocks;             /* ccf_fit.py:456                                  */
  double nparams;            /* ccf_fit.py:465                                  */
  int32_t kaiser_approx;     /* ccf_model.py:730-738                            */
  int32_t kaiser_coord_shift;/* ccf_model.py:698-707                            */
  int32_t niter;             /* fixed-point iterations (ccf_model.py:661,701), default 5 */
  int32_t from_data;         /* realspace_ccf_from_data: xi^r evaluated at fiducial coordinates, un-rescaled
                                abscissae (ccf_model.py:618-619,675-679); growth term beta*bias for linear_bias */
  int32_t empirical_corr;    /* velocity multiplied by (1 + Av delta(r)) (ccf_model.py:451-459)          */
  int32_t reserved;
} vk_eval_opts;

typedef struct vk_ctx vk_ctx;

int vk_abi_version(void);
int vk_device_count(void);
/* The VICTOR_HIP_* tuning / A-B environment knobs are read when a context is created, not per launch; call this after
 * changing one in a running process and every context re-reads them at its next entry point. */
void vk_knobs_refresh(void);

/* ---- the polling hand-off's launch rule (pure functions: no context, no GPU) -----------------------------------------------
 * Launches of at most 8 points whose (mu, v) planes are split over workgroups - one point per call, the reference's calling
 * convention (CCFLikelihood.py:32-39), the mailbox server's launches, the half-ensembles of victor_amd/sampler.py - may hand
 * their partial sums over by POLLING: the workgroup of a point's last work item waits, resident, for the other workgroups'
 * sums.  Waiting workgroups are harmless as long as they can never hold ALL workgroup slots of an XCD while the workgroups
 * they wait for are not placed yet.  The rule that guarantees it:
 *   - a launch polls only if at least two of its workgroups fit on a CU, so an XCD (32 CUs) offers 64 slots at least, if it
 *     fits on the chip at once, and if its context holds a reservation for its points (one waiter per point);
 *   - a context reserves on demand, up to 8 waiters, out of a budget of 32 per PROCESS (enforced here; a context's launches
 *     are stream-ordered, so its reservation bounds what it has resident); without reservation the launch hands over through
 *     completion counters - the same sums in the same order, bit-identical results, ~1.2 us slower;
 *   - across processes the reservations of one GPU must stay below 64 in all - one process with the full budget (the GPU
 *     owner, victor_amd/broker.py: 4 contexts x 8 requests) plus up to 31 processes that evaluate one point per call in one
 *     context, or up to 63 such processes (vk_poll_budget returns 1: the number of full-budget processes per device).  The
 *     processes of one user on one host enforce it among themselves through a ledger in /dev/shm, one file per GPU
 *     (victor_hip_poll2_<uid>_<PCI bus id>, 5 KB: a slot per owner, written by that owner only).  An owner is {pid, start time
 *     of that pid, inode of its pid namespace, library instance} - a recycled pid does not inherit a dead process's
 *     reservation, two copies of the library in one process keep a slot each; slots of owners that are gone (dead, a zombie,
 *     the pid now another process) are ignored and reused; a slot written in ANOTHER pid namespace (containers sharing
 *     /dev/shm) cannot be judged from here and counts as living, for good: the bound errs towards the counters.  The file is
 *     opened without following symbolic links and must be a regular file of this user, closed to group and others, of the
 *     expected size and version: otherwise the process does not poll at all.  Without /dev/shm (nothing to create the file
 *     in) the process budget alone applies.  A process beyond the bound gets no reservation and hands over through the
 *     counters; a launch asks for a reservation only if it would poll with it, and after a refusal only once the ledger has
 *     changed.  Processes that do not share that file (other users, other /dev/shm mounts) are the operator's to keep within
 *     the bound.
 * Failure mode if the rule is broken (or a launch is lost): a waiting workgroup gives up after 5 s of wall clock, the call -
 * or, for enqueued work, the next call / vk_sync on that context - returns VK_E_HIP ("waited ... for partial sums that never
 * arrived") and the context stays unusable: destroy it and create a new one.  No result of such a launch is delivered.
 * vk_poll_rule: 1 when a launch of `n_points` points, `parts` workgroups per plane and `workgroups` workgroups in all, of
 * which `workgroups_per_cu` fit on a CU of a device with `n_cu` CUs, polls under a context reservation of `reserved` waiters.
 * vk_poll_grant: how many MORE waiters a context holding `ctx_reserved` is granted when it wants `want`, the process has
 * `process_reserved` reserved in all and the other living processes of the ledger `others_reserved` (0 or want - ctx_reserved:
 * all or nothing).  vk_poll_device_reserved: what the ledger of `ctx`'s GPU holds for the other processes and for this one
 * (VK_E_ARG, *others = -1, when no ledger could be mapped). */
int32_t vk_poll_rule(int64_t n_points, int32_t parts, int64_t workgroups, int32_t workgroups_per_cu, int32_t n_cu, int32_t reserved);
int32_t vk_poll_grant(int32_t others_reserved, int32_t process_reserved, int32_t ctx_reserved, int32_t want);
int32_t vk_poll_budget(int32_t* per_process, int32_t* xcd_slots);
int32_t vk_poll_device_reserved(const vk_ctx* ctx, int32_t* others, int32_t* mine);

/* ---- the ledger's operations on a file of the caller's choosing, for an owner of the caller's making ------------------------
 * DEVELOPMENT entry points (tests/test_ledger.py drives them without a GPU): like every development switch of the library they
 * answer only with VICTOR_HIP_DEV=1 in the environment (otherwise: NULL / -1 / 0, nothing is touched).
 * vk_ledger_open_at: opens (creates) the ledger file at `path` and claims a slot for the owner {pid, start, ns, lib}; pid <= 0:
 *   the calling process itself (lib != 0 then stands for another copy of the library in it).  *status: 0 opened, 1 unavailable
 *   (cannot be created), 2 untrusted (a symbolic link, another owner or mode, wrong size / magic / version, a lock nobody
 *   releases), 3 full.  vk_ledger_close(led, keep_slot): keep_slot != 0 leaves the slot behind as a killed process would.
 * vk_ledger_others: waiters reserved by the other owners that are living or cannot be judged.  vk_ledger_grant / _release:
 *   the library's own grant / release protocol with *process_reserved standing for the owner's process-wide count.
 * vk_ledger_self(what, pid): 0 pid, 1 start time, 2 pid-namespace inode, 3 library instance of the caller; 4 start time of
 *   process `pid` (0: gone), 5: 1 if process `pid` is there (not gone, not a zombie).
 * vk_ledger_layout: bytes of the file's header and of a slot {int64 pid, uint64 start, uint64 ns, uint64 lib, int32 reserved,
 *   int32 pad}, number of slots, format version (header: uint32 magic "VKPL", uint32 version, uint32 generation, uint32 pad). */
void vk_ledger_layout(int32_t* header_bytes, int32_t* slot_bytes, int32_t* slots, int32_t* version);
uint64_t vk_ledger_self(int32_t what, int64_t pid);
void* vk_ledger_open_at(const char* path, int64_t pid, uint64_t start, uint64_t ns, uint64_t lib, int32_t* status);
int32_t vk_ledger_slot(const void* led);
int32_t vk_ledger_others(const void* led);
uint32_t vk_ledger_generation(const void* led);
int32_t vk_ledger_grant(void* led, int32_t* process_reserved, int32_t ctx_reserved, int32_t want);
void vk_ledger_release(void* led, int32_t* process_reserved, int32_t n);
void vk_ledger_close(void* led, int32_t keep_slot);

/* Copies every table to `device`.  On failure returns NULL and writes a message to err. */
vk_ctx* vk_create(const vk_tables* tables, int device, char* err, size_t errlen);
void vk_destroy(vk_ctx* ctx);
const char* vk_last_error(const vk_ctx* ctx);
/* name of the theory-kernel variant the most recent evaluation launched (diagnostics / benchmarks) */
const char* vk_last_kernel(const vk_ctx* ctx);
/* 1 when that launch also took the chi-square / log-likelihood (fused tail), 0 when K2 ran as its own launch */
int vk_last_fused(const vk_ctx* ctx);
/* 1 when that launch handed the partial sums of its split planes over by polling (vk_poll_rule below), 0 otherwise */
int vk_last_polled(const vk_ctx* ctx);
void vk_default_opts(vk_eval_opts* opts);

/* Full likelihood for n parameter rows (host buffers).  Any of lnl/chi2/theory may be NULL.
 * lnl[i] = -inf, chi2[i] = +inf where the reference returns (-inf, inf) (ccf_fit.py:447-450,477-481).
 * theory is [n][n_ell*n_s].  Synchronous for the caller: the results are in lnl / chi2 on return.  (Calls of a few
 * hundred points that do not ask for `theory` return as soon as the results have arrived in pinned host memory, with
 * the launch still retiring on the context's stream; every later call, vk_sync and vk_destroy are ordered behind it.) */
int vk_eval_batch(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n,
                  double* lnl, double* chi2, double* theory);

/* The same call in two halves for callers that have host work to overlap with the launch (the walker ensembles of
 * victor_amd/sampler.py advance one half of their walkers while the other half is on the GPU): vk_eval_batch_begin copies the
 * rows and enqueues the launch (1 <= n <= 4096), vk_eval_batch_finish waits for lnl / chi2 (either may be NULL).  One batch
 * per context at a time - several contexts may each have one in flight; nothing else may be called on a context between
 * its begin and its finish. */
int vk_eval_batch_begin(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n);
int vk_eval_batch_finish(vk_ctx* ctx, double* lnl, double* chi2);

/* ---- lock-step Metropolis walkers, advanced natively --------------------------------------------------------------------
 * The reference is sampled by cobaya, one likelihood per call (victor/likelihoods/CCFLikelihood.py:32-39);
 * victor_amd/sampler.py: EnsembleMetropolis advances W random-walk Metropolis chains together, one likelihood batch per step,
 * from the priors / proposal widths of the same `params:` block (config/boss_cobaya_config.yaml:50-97 of the reference).
 * vk_walk_run is that sampler's step loop in the library: the ensemble as two halves on two contexts (vk_eval_batch_begin /
 * _finish: half A of step t + 1 is on the GPU while half B of step t is accepted / rejected), rows formed as
 * CCFModel._param_rows forms them (one routine for the Alcock-Paczynski factors: vk_epsilon_to_ap), the caller's pre-drawn
 * random numbers - without `speculate` the same launches as the Python loop, hence the same chain bit for bit
 * (tests/test_gpu_workloads.py), without the host's ~20 NumPy calls per step.
 *   columns[j]   row column (VK_P_*) the j-th sampled parameter is written to, or VK_WALK_EPSILON: the parameter is epsilon and
 *                the columns APERP, APAR, EPSILON follow from it (apar = alpha eps^(-2/3), aperp = eps apar, ccf_model.py:589-592)
 *   lo, hi       uniform prior box; a proposal outside it is evaluated at the walker's position, discarded, and reads -inf
 *   base_rows    [n_walkers][VK_NPAR]: the fixed parameters and defaults of every row
 *   speculate    != 0: TWO steps per launch.  The proposal of step t + 1 starts from the proposal of step t (accepted) or from
 *                the old position (rejected) - both are known when step t is proposed, so a walker's three points travel in one
 *                launch and both decisions are taken when the results arrive: two steps per round trip host -> GPU -> host
 *                (what bounds a small ensemble: the GPU is far from full) for three evaluations instead of two.  The same
 *                proposals, the same acceptance levels as the step-by-step loop; n_evals counts the evaluations that loop
 *                would have made.  PARITY WITH THE STEP-BY-STEP LOOP IS TO ROUNDING, NOT BIT FOR BIT: a launch of three rows
 *                per walker takes another work split than one of one row, so the log-likelihoods agree to ~1e-13 relative
 *                and an acceptance `logu < lnL' - lnL` decided within that margin may fall the other way; positions and
 *                decisions were identical over every chain compared so far (thousands of steps), the stored lnL differ in
 *                their last bits.  A caller who needs the step-by-step chain bit for bit passes speculate = 0.  What does NOT
 *                vary: the chain for a given `speculate` is independent of how a run is cut into vk_walk_run calls - a left-
 *                over single step travels in a launch of the same shape (its two candidate rows idle).  Worth it while three
 *                times the ensemble still fits the idle part of the GPU (victor_amd/sampler.py chooses; ignored when a launch
 *                would exceed 4096 rows)
 * vk_walk_run: x [W][P] and lnl [W] are the ensemble's state (in / out); dz [n_steps][W][P] the proposal increments, logu
 * [n_steps][W] the log acceptance levels; chain [n_steps][W][P] and lnl_hist [n_steps][W] receive the state after every step
 * (either may be NULL); *n_accept and *n_evals are incremented.  One or two contexts holding the same tables (two: the halves
 * overlap); nothing else may use those contexts during a run.  On error the code is returned, vk_walk_last_error gives the
 * text, no batch stays begun on the contexts and x / lnl hold the state after the last completed half-step. */
#define VK_WALK_EPSILON (-1)
typedef struct vk_walk vk_walk;
vk_walk* vk_walk_create(vk_ctx* const* ctxs, int32_t n_ctx, const vk_eval_opts* opts, int32_t n_walkers, int32_t n_params,
                        const int32_t* columns, const double* lo, const double* hi, const double* base_rows, double alpha,
                        int32_t speculate, char* err, size_t errlen);
int vk_walk_run(vk_walk* w, int64_t n_steps, double* x, double* lnl, const double* dz, const double* logu, double* chain,
                double* lnl_hist, int64_t* n_accept, int64_t* n_evals);
const char* vk_walk_last_error(const vk_walk* w);
void vk_walk_destroy(vk_walk* w);
/* apar[i] = alpha * eps[i]^(-2/3) (alpha == 1: no multiply), aperp[i] = eps[i] * apar[i] with libm's pow, as the reference
 * evaluates them on Python floats (ccf_model.py:589-592): the one routine behind every row the package forms from an epsilon.
 * Pure host arithmetic: no context, no GPU. */
void vk_epsilon_to_ap(const double* eps, int64_t n, double alpha, double* aperp, double* apar);

/* Theory multipoles on a caller-supplied s grid: out[n][n_ell][n_s] with the caller's own
 * projection weights w_ell[n_ell][n_mu] on mu[n_mu] (host buffers). */
int vk_theory_batch(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n,
                    const double* s, int32_t n_s, const double* mu, int32_t n_mu,
                    const double* w_ell, int32_t n_ell, double* out);

/* xi^s(mu_i, s_j) for every point: out[n][n_mu][n_s] (host buffers). */
int vk_xi_smu_batch(vk_ctx* ctx, const vk_eval_opts* opts, const double* params, int64_t n,
                    const double* s, int32_t n_s, const double* mu, int32_t n_mu, double* out);

/* ---- device-resident variants (inputs/outputs already in HBM) ----------------------- */
void* vk_device_alloc(vk_ctx* ctx, size_t bytes);
void vk_device_free(vk_ctx* ctx, void* ptr);
int vk_memcpy_h2d(vk_ctx* ctx, void* dst, const void* src, size_t bytes);
int vk_memcpy_d2h(vk_ctx* ctx, void* dst, const void* src, size_t bytes);
/* enqueue on the context's stream; d_theory_ws is a device workspace of n*N doubles.  With d_lnl or d_chi2 given it is SCRATCH:
 * its contents after the call are unspecified (a launch that takes the chi-square inside the theory kernel never writes the
 * theory vectors to HBM).  With d_lnl = d_chi2 = NULL the theory vectors [n][N] are the result and are left in d_theory_ws. */
int vk_eval_batch_device_async(vk_ctx* ctx, const vk_eval_opts* opts, const double* d_params,
                               int64_t n, double* d_lnl, double* d_chi2, double* d_theory_ws);
int vk_sync(vk_ctx* ctx);

/* ---- joint fit of several data vectors sharing one parameter batch (block-diagonal covariance) ----------------------
 * BASELINE config 5 (density-split quantiles: five CCFFit-equivalents, N = 5 x 120): the reference has no joint class
 * (density-split centres are only mentioned, victor/ccf_model.py:28-30); with a block-diagonal covariance chi2 and lnL of
 * the blocks add.  One context per block, all on one device; ONE parameter upload, every block's two kernels enqueued on its
 * own stream behind the lead context's stream (the launches overlap on the GPU), lnL and chi2 summed on the device in block
 * order; a block that fails its guards (-inf, inf) fails the point.  Enqueued; vk_sync(ctxs[0]) waits for the result.
 * d_ws: vk_joint_workspace_doubles(...) doubles of device memory. */
size_t vk_joint_workspace_doubles(vk_ctx* const* ctxs, int32_t n_ctx, int64_t n);
int vk_joint_eval_device_async(vk_ctx* const* ctxs, int32_t n_ctx, const vk_eval_opts* opts, const double* d_params,
                               int64_t n, double* d_lnl, double* d_chi2, double* d_ws);

/* ---- many one-point callers sharing one GPU: mailboxes in shared memory -----------------------------------------------
 * The reference is sampled by cobaya, which asks for ONE likelihood per call (victor/likelihoods/CCFLikelihood.py:32-39); more
 * throughput comes from several chains, one process each, under mpirun (README.md:30).  P processes with a context each
 * serialise P small launches on the GPU.  Instead ONE owner process holds the context and serves an array of mailboxes that
 * lives in memory shared with the chains (a file in /dev/shm, victor_amd/broker.py): a chain writes its parameter row and
 * bumps `req_seq`; the owner evaluates everything that is pending as ONE launch and answers each mailbox by
 * writing lnl / chi2 / status and then `resp_seq = req_seq`.  The chains never touch the GPU (or this library).
 *
 * Protocol of one mailbox (all fields naturally aligned; release / acquire accesses on the two sequence words - this library's
 * side uses them on every architecture; the Python client of victor_amd/broker.py writes plain stores and relies on x86-64's
 * total store order: on another architecture use a native client):
 *   client   state = VK_BOX_ATTACHED once, with its pid;  per call: row[] <- parameters, then req_seq <- req_seq + 1 (release);
 *            wait until resp_seq == req_seq (acquire), then read lnl, chi2, status.  One call in flight per mailbox.
 *   server   sees req_seq != resp_seq (acquire), copies row[], evaluates, writes lnl, chi2, status, then resp_seq <- the
 *            req_seq it served (release).
 * status is the VK_* code of the batch the request was part of (0: lnl / chi2 valid, failed rows report -inf / +inf as
 * vk_eval_batch does). */
#define VK_BOX_FREE 0
#define VK_BOX_ATTACHED 1
typedef struct vk_mailbox {          /* 256 bytes: the client's and the server's words on different cache lines */
  volatile uint64_t req_seq;         /* client -> server                                                      */
  volatile uint32_t state;           /* VK_BOX_*; written by the client (attach / detach) and by vk_serve_mailboxes, which frees
                                        the boxes of dead clients at the start of a call (no launch of its own in flight then) */
  uint32_t reserved0;
  volatile int64_t client_pid;
  uint64_t reserved1[5];
  double row[VK_NPAR];               /* 96 bytes, the parameter row in the column order above                 */
  uint64_t reserved2[4];
  volatile uint64_t resp_seq;        /* server -> client                                                      */
  double lnl, chi2;
  int32_t status;
  uint32_t reserved3;
  uint64_t reserved4[4];
} vk_mailbox;

typedef struct vk_serve_stats {      /* accumulated over calls until the caller zeroes it */
  uint64_t batches, evals, max_batch, windows_timed_out;
  double busy_seconds;               /* inside vk_eval_batch */
} vk_serve_stats;

/* Serve `n_boxes` mailboxes until *stop != 0 or `max_seconds` have passed (then returns VK_OK once nothing is in flight; call
 * it again - the owner looks after its clients between calls).  `ctxs`: 1..8 contexts holding the SAME tables on one device
 * (the owner creates them from one vk_tables): one launch per context may be in flight, so a round's requests start at once
 * on a free context while earlier rounds are still on the GPU - the launches overlap there like those of separate processes -
 * and requests that arrive together share a launch.  `gather_window_us`: after the first pending request of a round the
 * server waits up to this long for the requests of the other attached clients that are not being served already (chains in
 * lock-step then share one launch instead of splitting into ever smaller ones; 0: launch at once).  When nothing is pending
 * or in flight it polls for ~0.2 ms, then naps in steps of 50 us (1 ms after 50 ms of silence).  n_boxes <= 1024.  Returns a
 * VK_E_* code only for bad arguments: an evaluation error is reported to the requesting mailboxes (status) and the loop goes
 * on.  A launch carries at most `max_batch` requests (1..32; 0 = 32: smaller launches finish sooner and leave the other contexts
 * something to overlap with), and every request is evaluated with the work split of a single-point
 * vk_eval_batch call whatever shares its launch: lnl / chi2 of a row are bit-identical to vk_eval_batch(ctx, opts, row, 1, ...)
 * and do not depend on what the other clients were doing. */
int vk_serve_mailboxes(vk_ctx* const* ctxs, int32_t n_ctx, const vk_eval_opts* opts, vk_mailbox* boxes, int32_t n_boxes,
                       const volatile uint32_t* stop, double gather_window_us, int32_t max_batch, double max_seconds,
                       vk_serve_stats* stats);

/* ---- timing on the context's stream (HIP events) ------------------------------------ */
/* Marks: 0 = before theory kernel, 1 = between kernels, 2 = after likelihood kernel, recorded by
 * the next vk_eval_batch_device_async when enabled.  Times accumulate until reset. */
int vk_timing_enable(vk_ctx* ctx, int on);
int vk_timing_read(vk_ctx* ctx, double* theory_ms, double* like_ms, int64_t* launches, int reset);

/* ---- multi-GPU gather of log-likelihoods over RCCL/xGMI ------------------------------ */
#define VK_COMM_ID_BYTES 128
int vk_comm_unique_id(char* id_out /* [VK_COMM_ID_BYTES] */);
int vk_comm_init(vk_ctx* ctx, const char* id, int rank, int nranks);
/* all ranks contribute count doubles at d_send, every rank receives nranks*count at d_recv */
int vk_comm_allgather_async(vk_ctx* ctx, const double* d_send, double* d_recv, int64_t count);
int vk_comm_destroy(vk_ctx* ctx);
/* PCI bus id of the context's GPU ("0000:c1:00.0"): ranks compare these (with their host names) before building a
 * communicator - RCCL needs one device per rank, and a launcher may give two ranks the same GPU under different ordinals. */
int vk_device_bus_id(const vk_ctx* ctx, char* buf, size_t len);
/* One process driving several GPUs (one context per device, any host thread): communicators for all n contexts at once
 * (ncclCommInitAll - no unique id, no rendezvous), context i being rank i.  Fails with VK_E_RCCL when two contexts share a
 * device (RCCL refuses that; callers then gather through the host). */
int vk_comm_init_all(vk_ctx* const* ctxs, int32_t n);
/* An all-gather of HOST data that is collected later (one process per GPU; after vk_comm_init): `send` (count doubles) is copied
 * to pinned memory; upload, ncclAllGather and download are enqueued on the context's stream and nothing waits.
 * vk_comm_allgather_host_finish waits for the download and writes recv[nranks][count], rank-major.  One gather per context at a
 * time.  For monitoring traffic that must not sit in a caller's step: victor_amd/sampler.py exchanges the walkers'
 * log-likelihoods of a block of steps this way, collecting one block late, on a context of the gather's own (an all-gather
 * enqueued on a stream the walkers launch on would hold their launches back until the slowest rank has arrived). */
int vk_comm_allgather_host_begin(vk_ctx* ctx, const double* send, int64_t count);
int vk_comm_allgather_host_finish(vk_ctx* ctx, double* recv);

/* The all-gather of such a group in one call: ncclGroupStart, one ncclAllGather per context on that context's stream
 * (d_send[i]: count doubles on device i, d_recv[i]: n*count doubles on device i), ncclGroupEnd.  Enqueued. */
int vk_comm_allgather_group_async(vk_ctx* const* ctxs, int32_t n, const double* const* d_send, double* const* d_recv,
                                  int64_t count);
/* Writes a JSON object naming the HIP runtime this process mapped (path, runtime/driver version, the HIP version the
 * library was built with) and the RCCL that vk_comm_* uses (path, ncclGetVersion).  RCCL is looked up next to the mapped
 * HIP runtime first, so both come from one ROCm install (PyTorch's bundled pair when torch was imported before this
 * library, /opt/rocm's otherwise); VICTOR_HIP_RCCL_LIB names another one in development runs only (VICTOR_HIP_DEV=1).  Returns VK_E_RCCL (buffer still filled) if no RCCL
 * could be loaded. */
int vk_comm_info(char* buf, size_t len);
/* What the LIVE communicator of `ctx` says about itself: ncclCommCount, ncclCommUserRank, ncclCommCuDevice (each -1 when the
 * loaded library does not export the call or it fails).  VK_E_RCCL without a communicator.  A multi-GPU record carries these per
 * rank (bench.py: config.rccl.ranks) next to the PCI bus ids the ranks compared before building it. */
int vk_comm_rank_info(const vk_ctx* ctx, int32_t* count, int32_t* user_rank, int32_t* device);

#ifdef __cplusplus
}
#endif
#endif /* VICTOR_HIP_H */
