// vk_walk.cpp - lock-step Metropolis walkers advanced natively (include/victor_hip.h: vk_walk_*).  Host code only: the step loop
// drives the library's own public entry points (vk_eval_batch_begin / _finish) on the contexts it was given; compiled by the
// host compiler (vk_host.h for the context's fields it reads: begun_n, err, N, d_data).

#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "vk_host.h"

using vkh::check_opts;

size_t vkh::ctx_layout_walk(size_t* last_offset) {
  if (last_offset) *last_offset = offsetof(vk_ctx, spin_timeouts);
  return sizeof(vk_ctx);
}

// ---- lock-step Metropolis walkers advanced natively (include/victor_hip.h: vk_walk_*) ------------------------------------
// The step of victor_amd/sampler.py: EnsembleMetropolis.run restated in C++: the same two half-ensembles on two contexts, the
// same pipelining (half A of step t + 1 is on the GPU while the host accepts / rejects half B of step t), the same rows, the
// same launches - so the same chain -, without the ~20 NumPy calls per step that made the host the limit of an 8-walker step.
struct vk_walk {
  vk_ctx* ctx[2] = {nullptr, nullptr};
  int n_half = 0;                       // 1 or 2
  int lo[2] = {0, 0}, hi[2] = {0, 0};   // walkers of each half
  vk_eval_opts opts{};
  int W = 0, P = 0;
  std::vector<int> col;                 // row column per sampled parameter, or VK_WALK_EPSILON
  int eps = -1;                         // index of the parameter that is epsilon, or -1
  double alpha = 1.0;
  std::vector<double> box_lo, box_hi, rows, prop, lnl_prop, chi_prop;
  std::vector<char> inside;
  // two steps per launch (vk_walk_create: speculate): per walker the proposal of step t and BOTH proposals of step t + 1 - from
  // the accepted and from the rejected position - travel in one launch; three rows, proposals, results per walker
  bool speculate = false;
  std::vector<double> rows3, prop3, lnl3, chi3;
  std::vector<char> in3;
  std::string err;
};

// The Alcock-Paczynski factors from epsilon (ccf_model.py:589-592 of the reference: apar = alpha * epsilon**(-2/3), aperp =
// epsilon * apar, on Python floats: libm's pow) - ONE routine for every row the package forms from an epsilon: the walkers'
// rows below and, through vk_epsilon_to_ap, CCFModel._param_rows and the Python step loop (NumPy's vectorised power may
// differ from libm's in the last bit on hosts where it runs through SVML).
static inline void eps_to_ap(double e, double alpha, double* aperp, double* apar) {
  double a = pow(e, -2.0 / 3.0);
  if (alpha != 1.0) a = alpha * a;
  *apar = a;
  *aperp = e * a;
}

// the sampled columns of one row from the walker's coordinates `xs` (CCFModel._param_rows)
static inline void walk_fill_row(const vk_walk* w, const double* xs, double* row) {
  for (int j = 0; j < w->P; ++j)
    if (w->col[j] >= 0) row[w->col[j]] = xs[j];
  if (w->eps >= 0) {
    const double e = xs[w->eps];
    eps_to_ap(e, w->alpha, &row[VK_P_APERP], &row[VK_P_APAR]);
    row[VK_P_EPSILON] = e;
  }
}

static inline bool walk_in_box(const vk_walk* w, const double* p) {
  bool in = true;
  for (int j = 0; j < w->P; ++j) in = in && p[j] >= w->box_lo[j] && p[j] <= w->box_hi[j];      // (a NaN proposal is outside)
  return in;
}

static void walk_begin(vk_walk* w, int k, const double* x, const double* dz_t, int* rc) {
  const int P = w->P;
  for (int i = w->lo[k]; i < w->hi[k]; ++i) {
    double* pr = &w->prop[(size_t)i * P];
    for (int j = 0; j < P; ++j) pr[j] = x[(size_t)i * P + j] + dz_t[(size_t)i * P + j];
    const bool in = walk_in_box(w, pr);
    w->inside[i] = in ? 1 : 0;
    // a proposal outside the prior: its row keeps the walker's position (a valid point; the result is discarded)
    walk_fill_row(w, in ? pr : &x[(size_t)i * P], &w->rows[(size_t)i * VK_NPAR]);
  }
  const int r = vk_eval_batch_begin(w->ctx[k], &w->opts, &w->rows[(size_t)w->lo[k] * VK_NPAR], w->hi[k] - w->lo[k]);
  if (r != VK_OK && *rc == VK_OK) {
    *rc = r;
    w->err = w->ctx[k]->err;
  }
}

// Two steps in one launch.  Step t + 1's proposal is x_(t+1) + dz_(t+1) with x_(t+1) either the proposal of step t (accepted)
// or the old position (rejected): both candidates are known when step t is proposed, so all three points of a walker are
// evaluated together and the two decisions are taken when the results arrive - the ensemble advances two steps per round
// trip host -> GPU -> host, the limit of a small ensemble, for three evaluations instead of two.  The decisions are those of
// the step-by-step loop: the same proposals (the same additions), the same acceptance levels.
// dz_t1 == NULL: the LAST step of a run of odd length, sent in the shape of a two-step launch (the two candidate rows hold the
// walker's position, their results are discarded): a launch's work split follows its number of rows, so every log-likelihood of
// a run comes from launches of one shape and its last bits do not depend on how the caller cuts the run into pieces.
static void walk_begin2(vk_walk* w, int k, const double* x, const double* dz_t, const double* dz_t1, int* rc) {
  const int P = w->P;
  for (int i = w->lo[k]; i < w->hi[k]; ++i) {
    const double* xi = &x[(size_t)i * P];
    double* p0 = &w->prop3[(size_t)i * 3 * P];
    double* pa = p0 + P;
    double* pr = pa + P;
    for (int j = 0; j < P; ++j) {
      p0[j] = xi[j] + dz_t[(size_t)i * P + j];
      pa[j] = dz_t1 ? p0[j] + dz_t1[(size_t)i * P + j] : std::numeric_limits<double>::quiet_NaN();      // (NaN: outside the box)
      pr[j] = dz_t1 ? xi[j] + dz_t1[(size_t)i * P + j] : std::numeric_limits<double>::quiet_NaN();
    }
    const bool in0 = walk_in_box(w, p0), ina = in0 && walk_in_box(w, pa), inr = walk_in_box(w, pr);
    w->in3[(size_t)i * 3] = in0;
    w->in3[(size_t)i * 3 + 1] = ina;
    w->in3[(size_t)i * 3 + 2] = inr;
    double* row = &w->rows3[(size_t)i * 3 * VK_NPAR];
    walk_fill_row(w, in0 ? p0 : xi, row);                                   // (outside the prior: a valid point, result discarded)
    walk_fill_row(w, ina ? pa : (in0 ? p0 : xi), row + VK_NPAR);
    walk_fill_row(w, inr ? pr : xi, row + 2 * VK_NPAR);
  }
  const int r = vk_eval_batch_begin(w->ctx[k], &w->opts, &w->rows3[(size_t)w->lo[k] * 3 * VK_NPAR], 3 * (w->hi[k] - w->lo[k]));
  if (r != VK_OK && *rc == VK_OK) {
    *rc = r;
    w->err = w->ctx[k]->err;
  }
}

// logu_t1 == NULL: the launch carried one step only (walk_begin2 with dz_t1 == NULL)
static void walk_finish_accept2(vk_walk* w, int k, double* x, double* lnl, const double* logu_t, const double* logu_t1, double* chain_t,
                                double* hist_t, int64_t* n_accept, int64_t* n_evals, int* rc) {
  const int P = w->P;
  const size_t step_x = (size_t)w->W * P, step_u = (size_t)w->W;
  if (w->ctx[k]->begun_n == 0) return;                                 // its begin failed
  const int r = vk_eval_batch_finish(w->ctx[k], &w->lnl3[(size_t)w->lo[k] * 3], &w->chi3[(size_t)w->lo[k] * 3]);
  if (r != VK_OK) {
    if (*rc == VK_OK) {
      *rc = r;
      w->err = w->ctx[k]->err;
    }
    return;
  }
  const double minus_inf = -std::numeric_limits<double>::infinity();
  for (int i = w->lo[k]; i < w->hi[k]; ++i) {
    double* xi = &x[(size_t)i * P];
    const double* p0 = &w->prop3[(size_t)i * 3 * P];
    const char* in = &w->in3[(size_t)i * 3];
    const double* l3 = &w->lnl3[(size_t)i * 3];
    // step t
    const double lp0 = in[0] ? l3[0] : minus_inf;
    if (in[0]) *n_evals += 1;
    const bool acc0 = logu_t[i] < lp0 - lnl[i];                         // (false for NaN)
    if (acc0) {
      memcpy(xi, p0, (size_t)P * sizeof(double));
      lnl[i] = lp0;
      *n_accept += 1;
    }
    if (chain_t) memcpy(chain_t + (size_t)i * P, xi, (size_t)P * sizeof(double));
    if (hist_t) hist_t[i] = lnl[i];
    if (!logu_t1) continue;
    // step t + 1: the candidate that belongs to the position step t left
    const int c = acc0 ? 1 : 2;
    const double lp1 = in[c] ? l3[c] : minus_inf;
    if (in[c]) *n_evals += 1;
    if (logu_t1[i] < lp1 - lnl[i]) {
      memcpy(xi, p0 + (size_t)c * P, (size_t)P * sizeof(double));
      lnl[i] = lp1;
      *n_accept += 1;
    }
    if (chain_t) memcpy(chain_t + step_x + (size_t)i * P, xi, (size_t)P * sizeof(double));
    if (hist_t) hist_t[step_u + i] = lnl[i];
  }
}

static void walk_finish_accept(vk_walk* w, int k, double* x, double* lnl, const double* logu_t, int64_t* n_accept, int64_t* n_evals, int* rc) {
  const int P = w->P;
  if (w->ctx[k]->begun_n == 0) return;                                 // its begin failed
  const int r = vk_eval_batch_finish(w->ctx[k], &w->lnl_prop[w->lo[k]], &w->chi_prop[w->lo[k]]);
  if (r != VK_OK) {
    if (*rc == VK_OK) {
      *rc = r;
      w->err = w->ctx[k]->err;
    }
    return;
  }
  for (int i = w->lo[k]; i < w->hi[k]; ++i) {
    double lp = w->lnl_prop[i];
    if (w->inside[i]) *n_evals += 1; else lp = -std::numeric_limits<double>::infinity();
    if (logu_t[i] < lp - lnl[i]) {                                     // (false for NaN)
      memcpy(&x[(size_t)i * P], &w->prop[(size_t)i * P], (size_t)P * sizeof(double));
      lnl[i] = lp;
      *n_accept += 1;
    }
  }
}


extern "C" {

vk_walk* vk_walk_create(vk_ctx* const* ctxs, int32_t n_ctx, const vk_eval_opts* opts, int32_t n_walkers, int32_t n_params,
                        const int32_t* columns, const double* lo, const double* hi, const double* base_rows, double alpha,
                        int32_t speculate, char* err, size_t errlen) {
  auto bail = [&](const char* msg) -> vk_walk* {
    if (err && errlen) {
      strncpy(err, msg, errlen - 1);
      err[errlen - 1] = 0;
    }
    return nullptr;
  };
  if (!ctxs || n_ctx < 1 || n_ctx > 2 || !ctxs[0] || (n_ctx == 2 && !ctxs[1]) || !opts || !columns || !lo || !hi || !base_rows)
    return bail("vk_walk_create: NULL argument");
  if (n_walkers < 1 || n_params < 1 || n_params > VK_NPAR) return bail("vk_walk_create: need 1 <= walkers, 1 <= parameters <= VK_NPAR");
  if (n_ctx == 2 && (ctxs[0] == ctxs[1] || ctxs[0]->N != ctxs[1]->N)) return bail("vk_walk_create: the two contexts must be different and hold the same tables");
  for (int c = 0; c < n_ctx; ++c)
    if (!ctxs[c]->d_data) return bail("vk_walk_create: context was created without a data vector");
  if (check_opts(ctxs[0], opts) != VK_OK) return bail(ctxs[0]->err.c_str());
  vk_walk* w = new (std::nothrow) vk_walk();
  if (!w) return bail("out of memory");
  w->W = n_walkers;
  w->P = n_params;
  w->opts = *opts;
  w->alpha = alpha;
  const int h = (n_ctx == 2 && n_walkers >= 2) ? n_walkers / 2 : n_walkers;
  w->ctx[0] = ctxs[0];
  w->lo[0] = 0;
  w->hi[0] = h;
  w->n_half = 1;
  if (h < n_walkers) {
    w->ctx[1] = ctxs[1];
    w->lo[1] = h;
    w->hi[1] = n_walkers;
    w->n_half = 2;
  }
  for (int k = 0; k < w->n_half; ++k)
    if (w->hi[k] - w->lo[k] > kZeroCopyCap) {
      delete w;
      return bail("vk_walk_create: at most 4096 walkers per half-ensemble");
    }
  int n_eps = 0;
  for (int j = 0; j < n_params; ++j) {
    const int c = columns[j];
    if (c == VK_WALK_EPSILON) {
      w->eps = j;
      ++n_eps;
    } else if (c < 0 || c >= VK_NPAR || (c >= VK_P_APERP && c <= VK_P_EPSILON)) {
      delete w;
      return bail("vk_walk_create: a sampled parameter must name a row column other than aperp / apar / epsilon, or VK_WALK_EPSILON");
    }
    if (!(hi[j] > lo[j])) {
      delete w;
      return bail("vk_walk_create: the prior box needs hi > lo");
    }
    w->col.push_back(c);
  }
  if (n_eps > 1) {
    delete w;
    return bail("vk_walk_create: epsilon sampled twice");
  }
  w->box_lo.assign(lo, lo + n_params);
  w->box_hi.assign(hi, hi + n_params);
  w->rows.assign(base_rows, base_rows + (size_t)n_walkers * VK_NPAR);
  w->prop.resize((size_t)n_walkers * n_params);
  w->lnl_prop.resize(n_walkers);
  w->chi_prop.resize(n_walkers);
  w->inside.resize(n_walkers);
  // two steps per launch: three rows per walker, as long as a launch stays within the in-place buffers
  w->speculate = speculate != 0;
  for (int k = 0; k < w->n_half; ++k)
    if (3 * (w->hi[k] - w->lo[k]) > kZeroCopyCap) w->speculate = false;
  if (w->speculate) {
    w->rows3.resize((size_t)n_walkers * 3 * VK_NPAR);
    for (int i = 0; i < n_walkers; ++i)
      for (int c = 0; c < 3; ++c) memcpy(&w->rows3[((size_t)i * 3 + c) * VK_NPAR], base_rows + (size_t)i * VK_NPAR, VK_NPAR * sizeof(double));
    w->prop3.resize((size_t)n_walkers * 3 * n_params);
    w->lnl3.resize((size_t)n_walkers * 3);
    w->chi3.resize((size_t)n_walkers * 3);
    w->in3.resize((size_t)n_walkers * 3);
  }
  return w;
}

void vk_walk_destroy(vk_walk* w) { delete w; }

void vk_epsilon_to_ap(const double* eps, int64_t n, double alpha, double* aperp, double* apar) {
  if (!eps || !aperp || !apar) return;
  for (int64_t i = 0; i < n; ++i) eps_to_ap(eps[i], alpha, &aperp[i], &apar[i]);
}

const char* vk_walk_last_error(const vk_walk* w) { return w ? w->err.c_str() : ""; }

int vk_walk_run(vk_walk* w, int64_t n_steps, double* x, double* lnl, const double* dz, const double* logu, double* chain,
                double* lnl_hist, int64_t* n_accept, int64_t* n_evals) {
  if (!w || n_steps < 0 || !x || !lnl || (n_steps > 0 && (!dz || !logu))) return VK_E_ARG;
  for (int k = 0; k < w->n_half; ++k)
    if (w->ctx[k]->begun_n != 0) {
      w->err = "vk_walk_run: a batch begun on one of the contexts has not been collected";
      return VK_E_ARG;
    }
  int64_t acc = 0, ev = 0;
  int rc = VK_OK;
  const size_t step_x = (size_t)w->W * w->P, step_u = (size_t)w->W;
  int64_t t0 = 0;                        // steps taken by the two-steps-per-launch loop; the rest (one step at most) below
  if (w->speculate && n_steps >= 2) {
    const int64_t pairs = n_steps / 2;
    walk_begin2(w, 0, x, dz, dz + step_x, &rc);
    for (int64_t q = 0; q < pairs && rc == VK_OK; ++q) {
      const int64_t t = 2 * q;
      const double* dz_t = dz + (size_t)t * step_x;
      const double* lu_t = logu + (size_t)t * step_u;
      double* ch = chain ? chain + (size_t)t * step_x : nullptr;
      double* hi = lnl_hist ? lnl_hist + (size_t)t * step_u : nullptr;
      if (w->n_half == 2) walk_begin2(w, 1, x, dz_t, dz_t + step_x, &rc);
      walk_finish_accept2(w, 0, x, lnl, lu_t, lu_t + step_u, ch, hi, &acc, &ev, &rc);
      if (q + 1 < pairs && rc == VK_OK) walk_begin2(w, 0, x, dz_t + 2 * step_x, dz_t + 3 * step_x, &rc);   // the next two steps go out now
      if (w->n_half == 2) walk_finish_accept2(w, 1, x, lnl, lu_t, lu_t + step_u, ch, hi, &acc, &ev, &rc);
    }
    t0 = 2 * pairs;
  }
  if (w->speculate && t0 < n_steps && rc == VK_OK) {
    // one step left over: the same three-rows-per-walker launch with the candidate rows idle (walk_begin2)
    const double* dz_t = dz + (size_t)t0 * step_x;
    const double* lu_t = logu + (size_t)t0 * step_u;
    double* ch = chain ? chain + (size_t)t0 * step_x : nullptr;
    double* hi = lnl_hist ? lnl_hist + (size_t)t0 * step_u : nullptr;
    walk_begin2(w, 0, x, dz_t, nullptr, &rc);
    if (w->n_half == 2) walk_begin2(w, 1, x, dz_t, nullptr, &rc);
    walk_finish_accept2(w, 0, x, lnl, lu_t, nullptr, ch, hi, &acc, &ev, &rc);
    if (w->n_half == 2) walk_finish_accept2(w, 1, x, lnl, lu_t, nullptr, ch, hi, &acc, &ev, &rc);
    t0 = n_steps;
  }
  if (t0 < n_steps && rc == VK_OK) walk_begin(w, 0, x, dz + (size_t)t0 * step_x, &rc);
  for (int64_t t = t0; t < n_steps && rc == VK_OK; ++t) {
    const double* dz_t = dz + (size_t)t * step_x;
    const double* lu_t = logu + (size_t)t * step_u;
    if (w->n_half == 2) walk_begin(w, 1, x, dz_t, &rc);
    walk_finish_accept(w, 0, x, lnl, lu_t, &acc, &ev, &rc);
    if (t + 1 < n_steps && rc == VK_OK) walk_begin(w, 0, x, dz_t + step_x, &rc);      // half A of the next step goes out now
    if (w->n_half == 2) walk_finish_accept(w, 1, x, lnl, lu_t, &acc, &ev, &rc);
    if (chain) memcpy(chain + (size_t)t * step_x, x, step_x * sizeof(double));
    if (lnl_hist) memcpy(lnl_hist + (size_t)t * step_u, lnl, step_u * sizeof(double));
  }
  if (rc != VK_OK)                       // nothing may stay begun on the contexts: collect (and drop) what is in flight
    for (int k = 0; k < w->n_half; ++k)
      if (w->ctx[k]->begun_n != 0) (void)vk_eval_batch_finish(w->ctx[k], nullptr, nullptr);
  if (n_accept) *n_accept += acc;
  if (n_evals) *n_evals += ev;
  return rc;
}

}  // extern "C"
