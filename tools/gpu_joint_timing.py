"""BASELINE config 5 (5-quantile joint fit, batch 16384) resident on one GPU: joint evals/s against the config-3 rate / 5."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from victor_amd.joint import JointFit
from tests import cases

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
joint = JointFit([victor_amd.CCFFit(*cases.dsplit_options(q)) for q in range(5)])
engines, opts = joint._plan({})
rows = joint.fits[0]._fit_rows(cases.halton_params(n), joint.fits[0].model)
ctxs, (d_rows, d_out, d_ws) = joint._device_buffers(engines, n)
lead = engines[0]
lead.upload(d_rows, rows)
d_chi = d_out + 8 * n


def run(k):
    for _ in range(k):
        joint.eval_device_async(engines, opts, d_rows, n, d_out, d_chi, d_ws)
    lead.sync()


t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end:
    run(1)
t0 = time.perf_counter(); run(10); dt = (time.perf_counter() - t0) / 10
print(f"joint 5 x {n}: {dt*1e3:.3f} ms per batch, {n/dt/1e6:.3f} M joint evals/s ({5*n/dt/1e6:.3f} M block evals/s), kernel {engines[0].last_kernel()}")
# one block alone, same batch, for reference
eng = engines[0]
bufs = [eng.alloc(n), eng.alloc(n), eng.alloc(n * eng.n_data)]
for _ in range(5):
    eng.eval_device_async(opts, d_rows, n, bufs[0], bufs[1], bufs[2])
eng.sync()
t0 = time.perf_counter()
for _ in range(10):
    eng.eval_device_async(opts, d_rows, n, bufs[0], bufs[1], bufs[2])
eng.sync()
d1 = (time.perf_counter() - t0) / 10
print(f"one block x {n}: {d1*1e3:.3f} ms, {n/d1/1e6:.3f} M evals/s, kernel {eng.last_kernel()}; joint / (5 x one block) = {dt/(5*d1):.3f}")
