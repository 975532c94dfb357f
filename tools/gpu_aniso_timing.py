"""Throughput of the anisotropic sigma_v(r, mu) template (bicubic patches from global memory, generic kernel)."""
import os, sys, time, tempfile, pathlib
sys.path.insert(0, '/root/repo')
import numpy as np
import victor_amd
from tests import cases
from tests.test_host import _aniso_inputs
tmp = pathlib.Path(tempfile.mkdtemp())
model, data = _aniso_inputs(tmp)
fit = victor_amd.CCFFit(model, data)
eng = fit._get_engine()
o = eng.make_opts(fit.model, fit.fit_options)
batch = 16384
rows = fit._fit_rows(cases.halton_params(batch), fit.model)
bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
eng.upload(bufs[0], rows)
for _ in range(3):
    eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3]); eng.sync()
t0 = time.perf_counter()
for _ in range(4):
    eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
eng.sync()
dt = (time.perf_counter() - t0) / 4
print(f"anisotropic sigma_v(r,mu): {dt*1e3:.2f} ms/batch {batch/dt:.0f} evals/s ({eng.last_kernel()})")
