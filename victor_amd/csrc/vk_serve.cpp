// vk_serve.cpp - mailboxes: many one-point callers, one launch (include/victor_hip.h: vk_serve_mailboxes).  Host code only: the
// serving loop of the GPU owner process (victor_amd/broker.py) on the in-place launch halves of victor_hip.hip (vkh::zc_begin /
// zc_finish); compiled by the host compiler.

#include <time.h>

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <cstring>
#include <limits>
#include <vector>

#include "vk_host.h"
#include "vk_ledger.h"

using vkh::check_opts;
using vkh::cpu_relax;
using vkh::fail;
using vkh::host_scratch;
using vkh::HostScratch;
using vkh::sync_knobs;
using vkh::zc_begin;
using vkh::zc_finish;

size_t vkh::ctx_layout_serve(size_t* last_offset) {
  if (last_offset) *last_offset = offsetof(vk_ctx, spin_timeouts);
  return sizeof(vk_ctx);
}

// Is process `pid` still there?  kill(pid, 0) answers for zombies as well (a dead child nobody has waited for yet), so the
// state letter of /proc/<pid>/stat decides for those (vk_ledger.cpp: process_state; victor_amd/broker.py: _pid_alive is the same test).
static bool process_alive(long long pid) { return vkl::process_state(pid, nullptr) != 0; }

extern "C" {

// ---- the mailbox layout is mirrored field by field in victor_amd/broker.py ----
static_assert(sizeof(vk_mailbox) == 256, "vk_mailbox is mirrored field by field in victor_amd/broker.py");
static_assert(offsetof(vk_mailbox, row) == 64 && offsetof(vk_mailbox, resp_seq) == 192, "vk_mailbox layout");

int vk_serve_mailboxes(vk_ctx* const* ctxs, int32_t n_ctx, const vk_eval_opts* opts, vk_mailbox* boxes, int32_t n_boxes,
                       const volatile uint32_t* stop, double gather_window_us, int32_t max_batch, double max_seconds,
                       vk_serve_stats* stats) {
  if (!ctxs || n_ctx < 1 || n_ctx > 8 || !ctxs[0]) return VK_E_ARG;
  vk_ctx* lead = ctxs[0];
  int rc = check_opts(lead, opts);
  if (rc) return rc;
  if (!boxes || n_boxes < 1 || n_boxes > 1024 || !stop || !(max_seconds > 0)) return fail(lead, VK_E_ARG, "vk_serve_mailboxes: bad arguments");
  for (int c = 0; c < n_ctx; ++c) {
    if (!ctxs[c] || !ctxs[c]->d_data) return fail(lead, VK_E_ARG, "vk_serve_mailboxes: context %d is NULL or was created without a data vector", c);
    if (ctxs[c]->N != lead->N) return fail(lead, VK_E_ARG, "vk_serve_mailboxes: the contexts must hold the same tables");
  }
  using clock = std::chrono::steady_clock;
  const auto t_start = clock::now();
  const auto window = std::chrono::nanoseconds((long long)(std::max(gather_window_us, 0.0) * 1e3));
  const int cap = (max_batch >= 1 && max_batch <= kServeMaxBatch) ? max_batch : kServeMaxBatch;
  // Mailboxes of clients that died without detaching are handed on HERE, at the start of a slice, when no launch of this loop
  // carries anybody's request: a box freed while a flight still held its dead owner's request could be claimed by a new client
  // whose first sequence number equals the one in flight - and would then be answered with the dead client's result.  The
  // sequence words are zeroed before FREE is published (release); clients only ever claim FREE boxes (under their file lock)
  // and this only touches ATTACHED boxes of dead processes, so the two never write the same box.
  for (int b = 0; b < n_boxes; ++b) {
    vk_mailbox& box = boxes[b];
    if (box.state == VK_BOX_ATTACHED && !process_alive((long long)box.client_pid)) {
      box.req_seq = 0;
      box.resp_seq = 0;
      __atomic_store_n(&box.state, (uint32_t)VK_BOX_FREE, __ATOMIC_RELEASE);
    }
  }
  // One launch per context may be in flight: a round's requests go to a free context at once and its results are handed back
  // when they have arrived, while the requests that come in meanwhile take the next context - the launches overlap on the GPU
  // like those of separate processes (each context has its own stream), and chains that post together still share one.
  struct Flight {
    bool active = false;
    int n = 0;
    clock::time_point t0;
    std::vector<int> idx;
    std::vector<uint64_t> seq;
    std::vector<double> rows, lnl, chi2;
  };
  std::vector<Flight> fl(n_ctx);
  for (auto& f : fl) {
    f.idx.resize(n_boxes);
    f.seq.resize(n_boxes);
    f.rows.resize((size_t)n_boxes * VK_NPAR);
    f.lnl.resize(n_boxes);
    f.chi2.resize(n_boxes);
  }
  std::vector<uint64_t> taken(n_boxes, 0);   // req_seq of the request of this mailbox that is in flight (0: none)
  std::vector<int> pend(n_boxes);
  std::vector<uint64_t> pend_seq(n_boxes);
  auto deliver = [&](Flight& f, int code) {
    for (int k = 0; k < f.n; ++k) {
      vk_mailbox& box = boxes[f.idx[k]];
      box.lnl = code == VK_OK ? f.lnl[k] : -std::numeric_limits<double>::infinity();
      box.chi2 = code == VK_OK ? f.chi2[k] : std::numeric_limits<double>::infinity();
      box.status = code;
      __atomic_store_n(&box.resp_seq, f.seq[k], __ATOMIC_RELEASE);
      taken[f.idx[k]] = 0;
    }
    if (stats) {
      stats->batches += 1;
      stats->evals += (uint64_t)f.n;
      if ((uint64_t)f.n > stats->max_batch) stats->max_batch = (uint64_t)f.n;
      stats->busy_seconds += std::chrono::duration<double>(clock::now() - f.t0).count();
    }
    f.active = false;
  };
  auto t_last_work = t_start;
  auto t_first_pending = t_start;
  bool waiting = false;
  int last_batch = 0;
  for (;;) {
    // results that have arrived
    int in_flight = 0, served = 0;       // launches in flight, requests they carry
    for (int c = 0; c < n_ctx; ++c) {
      Flight& f = fl[c];
      if (!f.active) continue;
      const int done = zc_finish(ctxs[c], f.n, f.lnl.data(), f.chi2.data(), false);
      if (done != 0) {
        deliver(f, done < 0 ? done : VK_OK);
        t_last_work = clock::now();
      } else {
        ++in_flight;
        served += f.n;
      }
    }
    // one scan: who is attached, who has a new request
    int n = 0, attached = 0;
    for (int b = 0; b < n_boxes; ++b) {
      vk_mailbox& box = boxes[b];
      if (box.state != VK_BOX_ATTACHED) continue;
      ++attached;
      const uint64_t r = __atomic_load_n(&box.req_seq, __ATOMIC_ACQUIRE);
      if (r != box.resp_seq && r != taken[b]) {
        pend[n] = b;
        pend_seq[n] = r;
        ++n;
      }
    }
    const auto now = clock::now();
    // the slice is over (or the owner is leaving): nothing new is started, what is in flight is brought home, then back to
    // the caller - also under a load that never leaves a quiet moment
    const bool expired = *stop || std::chrono::duration<double>(now - t_start).count() >= max_seconds;
    if (expired) {
      if (in_flight == 0) return VK_OK;
      cpu_relax();
      continue;
    }
    if (n == 0) {
      waiting = false;
      if (in_flight) {
        cpu_relax();
        continue;
      }
      const auto idle = now - t_last_work;
      if (idle > std::chrono::milliseconds(50)) {
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, nullptr);
      } else if (idle > std::chrono::microseconds(200)) {
        struct timespec ts = {0, 50000};
        nanosleep(&ts, nullptr);
      } else {
        cpu_relax();
      }
      continue;
    }
    int free_ctx = -1;
    for (int c = 0; c < n_ctx && free_ctx < 0; ++c)
      if (!fl[c].active) free_ctx = c;
    if (free_ctx < 0) {                 // every context is busy: the requests wait (and gather) until one comes back
      cpu_relax();
      continue;
    }
    // chains in lock-step post within a few microseconds of each other: give the ones that were part of the previous round
    // (and one more) the window to arrive, so that they share a launch instead of splitting into ever smaller batches -
    // but only among the chains that are not being served already
    // (with no more clients than contexts every request simply takes a context of its own, at once: measured, 4 chains on 4
    // contexts 160 k evaluations/s without the window against 144 k with it - tools/gpu_broker_sweep.py, profiles/r04)
    const int expect = std::min(std::min(attached - served, last_batch + 1), cap);
    if (n < expect && window.count() > 0 && attached > n_ctx) {
      if (!waiting) {
        waiting = true;
        t_first_pending = now;
      }
      if (now - t_first_pending < window) {
        cpu_relax();
        continue;
      }
      if (stats) stats->windows_timed_out += 1;
    }
    waiting = false;
    Flight& f = fl[free_ctx];
    vk_ctx* ctx = ctxs[free_ctx];
    if (n > cap) n = cap;                              // the others stay pending: the next free context takes them
    f.n = n;
    for (int k = 0; k < n; ++k) {
      f.idx[k] = pend[k];
      f.seq[k] = pend_seq[k];
      taken[pend[k]] = pend_seq[k];
      memcpy(&f.rows[(size_t)k * VK_NPAR], boxes[pend[k]].row, VK_NPAR * sizeof(double));
    }
    f.t0 = clock::now();
    f.active = true;
    last_batch = n;
    HostScratch sc;
    rc = hipSetDevice(ctx->device) == hipSuccess ? host_scratch(ctx, n, &sc) : VK_E_HIP;
    if (rc == VK_OK) {
      sync_knobs(ctx);
      ctx->split_as_single = true;
      rc = zc_begin(ctx, opts, f.rows.data(), n, true, sc.d_th);
      if (rc == 0) {                    // no in-place buffers on this system (or a development knob): the blocking call
        rc = vk_eval_batch(ctx, opts, f.rows.data(), n, f.lnl.data(), f.chi2.data(), nullptr);
        ctx->split_as_single = false;
        deliver(f, rc);
        t_last_work = clock::now();
        rc = 1;
      }
    }
    ctx->split_as_single = false;
    if (rc < 0) {
      deliver(f, rc);                   // the requesting mailboxes learn about it; the loop goes on
      t_last_work = clock::now();
    }
  }
}

}  // extern "C"
