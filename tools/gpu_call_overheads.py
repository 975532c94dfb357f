"""Where a single-point CCFFit.log_likelihood call spends its time on the host side: the full call, the engine call with a
ready-made row (no dict handling), the bare ctypes call of vk_eval_batch, and back-to-back resident launches (the kernel with
its launch pipelined).  Microseconds per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from tests import cases
from victor_amd import _native as N

def per_call(f, n=4000):
    for _ in range(300):
        f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e6

for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    p = cases.point(cases.halton_params(8, with_beta=beta), 3)
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        fit.log_likelihood(p)
    eng, o, need_beta, need_fs8, o_struct = fit._single_point_plan()
    row = fit._scalar_row(p, need_beta, need_fs8)
    buf = np.array([row]); out = np.empty(2)
    p_rows, p_lnl, p_chi = N.as_dp(buf), N.as_dp(out[0:1]), N.as_dp(out[1:2])
    lib, ctx = eng._lib, eng._ctx
    rows = fit._fit_rows({k: np.array([v]) for k, v in p.items()}, fit.model)
    d = [eng.alloc(rows.size), eng.alloc(1), eng.alloc(1), eng.alloc(eng.n_data)]
    eng.upload(d[0], rows)
    def resident():
        eng.eval_device_async(o_struct, d[0], 1, d[1], d[2], d[3])
    t_res = per_call(resident, 4000); eng.sync()
    def resident_sync():
        eng.eval_device_async(o_struct, d[0], 1, d[1], d[2], d[3]); eng.sync()
    print(f"{name}: log_likelihood(dict) {per_call(lambda: fit.log_likelihood(p)):.2f} | _scalar_row alone {per_call(lambda: fit._scalar_row(p, need_beta, need_fs8)):.2f} | "
          f"eval_point(row) {per_call(lambda: eng.eval_point(o, row)):.2f} | bare vk_eval_batch {per_call(lambda: lib.vk_eval_batch(ctx, o, p_rows, 1, p_lnl, p_chi, None)):.2f} | "
          f"resident launch + stream sync {per_call(resident_sync):.2f} | resident launches back to back {t_res:.2f}", flush=True)
