"""Small-batch regime (cobaya calls the likelihood with ONE point per step, CCFLikelihood.py:32-39): a loop of calls for
rocprofv3 / wall-clock timing.  Usage: small_batch_loop.py {3|2|boss} BATCH {api|resident} [ITERS]

  api       fit.log_likelihood(dict) for BATCH == 1, fit.log_likelihood_batch(rows) otherwise (host buffers, PCIe inclusive)
  resident  vk_eval_batch_device_async on buffers already in HBM, one sync at the end
Prints one JSON line with the wall time per call.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd  # noqa: E402
from tests import cases  # noqa: E402

which, batch, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 500
if which == "boss":
    fit = victor_amd.CCFFit(*cases.boss_options("config"))
    hp = cases.halton_params(max(batch, 2), with_beta=True)
else:
    fit = victor_amd.CCFFit(*cases.synth_options(int(which)))
    hp = cases.halton_params(max(batch, 2))
hp = {k: v[:batch] for k, v in hp.items()}
eng = fit._get_engine()
opts = eng.make_opts(fit.model, fit.fit_options)
rows = fit._fit_rows(hp, fit.model)
if mode == "api":
    p = cases.point(hp, 0)
    call = (lambda: fit.log_likelihood(p)) if batch == 1 else (lambda: fit.log_likelihood_batch(rows))
    for _ in range(20):
        call()
    t0 = time.perf_counter()
    for _ in range(iters):
        call()
    dt = (time.perf_counter() - t0) / iters
else:
    bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
    eng.upload(bufs[0], rows)
    for _ in range(20):
        eng.eval_device_async(opts, bufs[0], batch, bufs[1], bufs[2], bufs[3])
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        eng.eval_device_async(opts, bufs[0], batch, bufs[1], bufs[2], bufs[3])
    eng.sync()
    dt = (time.perf_counter() - t0) / iters
print(json.dumps({"config": which, "batch": batch, "mode": mode, "iters": iters, "us_per_call": dt * 1e6,
                  "evals_per_s": batch / dt, "kernel": eng.last_kernel()}))
