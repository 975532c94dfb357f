"""Cells kernel on the fixed config-3 tables: evals/s vs VICTOR_HIP_POINT_CAP (the staging per workgroup is cheap: 64 and 256 per CU tie)."""
import os, sys, time
sys.path.insert(0, '/root/repo')
import victor_amd
from tests import cases
from victor_amd import _native
fit = victor_amd.CCFFit(*cases.synth_options(3))
eng = fit._get_engine()
o = eng.make_opts(fit.model, fit.fit_options)
_native.set_knob("VICTOR_HIP_MAPPING", "cells")
for batch in (2000, 10000, 20000, 30000):
    rows = fit._fit_rows(cases.halton_params(batch), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
    eng.upload(bufs[0], rows)
    line = f"config3 cells batch {batch}:"
    for cap in ("4", "8", "16", "32", "64", "256"):
        _native.set_knob("VICTOR_HIP_POINT_CAP", cap)
        for _ in range(3):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3]); eng.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        dt = (time.perf_counter() - t0) / 10
        line += f"  cap{cap}: {batch/dt/1e6:.3f}M"
    print(line, flush=True)
    for b in bufs: eng.free(b)
