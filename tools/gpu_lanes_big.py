"""Lanes kernel at batch sizes beyond the launch cap (64 workgroups per CU): resident evals/s vs VICTOR_HIP_LANES_CAP."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _devlib import use_dev_library
use_dev_library()          # the lanes kernel lives in the development build of the library only
import victor_amd
from tests import cases
from victor_amd import _native
fit = victor_amd.CCFFit(*cases.synth_options(3))
eng = fit._get_engine()
o = eng.make_opts(fit.model, fit.fit_options)
for batch in (65536, 131072, 262144, 1048576):
    rows = fit._fit_rows(cases.halton_params(batch), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
    eng.upload(bufs[0], rows)
    for cap in ("64", "100000"):
        _native.set_knob("VICTOR_HIP_LANES_CAP", cap)
        for _ in range(2):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3]); eng.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        dt = (time.perf_counter() - t0) / 3
        print(f"batch {batch} cap {cap}: {dt*1e3:.2f} ms -> {batch/dt:.0f} evals/s ({eng.last_kernel()})", flush=True)
    for b in bufs: eng.free(b)
