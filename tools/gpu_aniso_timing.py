"""Throughput of the anisotropic sigma_v(r, mu) template (config-3 tables with a 3-key template, batch 16384): bicubic patches in
LDS on the cells kernel (SVA instantiations) against the generic kernel (patches from global memory)."""
import os, sys, time, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from tests import cases
from tests.test_host import _aniso_inputs
from victor_amd import _native

for non_uniform in (False, True):
    tmp = pathlib.Path(tempfile.mkdtemp())
    model, data = _aniso_inputs(tmp, non_uniform)
    fit = victor_amd.CCFFit(model, data)
    eng = fit._get_engine()
    o = eng.make_opts(fit.model, fit.fit_options)
    batch = 16384
    rows = fit._fit_rows(cases.halton_params(batch), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
    eng.upload(bufs[0], rows)
    for knob in (None, "1"):
        _native.set_knob("VICTOR_HIP_FORCE_GENERIC", knob)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3]); eng.sync()
        t0 = time.perf_counter()
        for _ in range(4):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        dt = (time.perf_counter() - t0) / 4
        print(f"anisotropic sigma_v(r,mu), {'non-uniform' if non_uniform else 'uniform'} mu knots: {dt*1e3:.2f} ms/batch "
              f"{batch/dt:.0f} evals/s ({eng.last_kernel()})", flush=True)
    _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
