"""Throughput of the generic kernel (non-uniform / nearly uniform grids, every option) next to the fast paths."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from tests import cases
from victor_amd import _native

def run(fit, label, batch=32768):
    eng = fit._get_engine()
    o = eng.make_opts(fit.model, fit.fit_options)
    rows = fit._fit_rows(cases.halton_params(batch), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
    eng.upload(bufs[0], rows)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3]); eng.sync()
    t0 = time.perf_counter()
    for _ in range(4):
        eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
    eng.sync()
    dt = (time.perf_counter() - t0) / 4
    print(f"{label}: {dt*1e3:.2f} ms/batch {batch/dt:.0f} evals/s ({eng.last_kernel()})")
    return eng.download(bufs[2], batch)

fit = victor_amd.CCFFit(*cases.synth_options(3))
a = run(fit, "config 3, uniform grids, default")
_native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
b = run(fit, "config 3, uniform grids, generic kernel")
_native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
print("max rel diff", np.max(np.abs(a / b - 1)))
# jitter the r grid (bin centres as mean separations): nearly uniform -> generic kernel with estimate + correction
import tempfile
src = np.load(os.path.join(cases.GOLDEN, "synth", "model.npy"), allow_pickle=True).item()
d = dict(src)
rng = np.random.default_rng(0)
d["r"] = src["r"] + rng.uniform(-0.2, 0.2, len(src["r"]))
tmp = tempfile.mkdtemp()
np.save(os.path.join(tmp, "jitter.npy"), d, allow_pickle=True)
m, dd = cases.synth_options(3)
for k in ("data_file",):
    dd["redshift_space_ccf"][k] = os.path.join(cases.GOLDEN, dd["redshift_space_ccf"][k]); dd["covariance_matrix"][k] = os.path.join(cases.GOLDEN, dd["covariance_matrix"][k])
dd["dir"] = ""
m["dir"] = tmp; m["input_model_data_file"] = "jitter.npy"
fitj = victor_amd.CCFFit(m, dd)
a = run(fitj, "config 3, jittered r grid: union-grid fast path")
_native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
b = run(fitj, "config 3, jittered r grid: generic kernel")
_native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
print("max rel diff", np.max(np.abs(a / b - 1)))
