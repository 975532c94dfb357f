"""Small and medium batches, resident: point-major against cells (ranges per point swept) - us per call.
Usage: gpu_small_batch_ab.py [config3|config2|boss] [batches...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases
from victor_amd import _native

which = sys.argv[1] if len(sys.argv) > 1 else "config3"
batches = [int(b) for b in sys.argv[2:]] or [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096]
opts, beta = (cases.boss_options("config"), True) if which == "boss" else (cases.synth_options(int(which[-1])), False)
fit = victor_amd.CCFFit(*opts)
eng = fit._get_engine()
o = eng.make_opts(fit.model, fit.fit_options)
nmax = max(batches)
bufs = [eng.alloc(nmax * 12), eng.alloc(nmax), eng.alloc(nmax), eng.alloc(nmax * eng.n_data)]


def timed(batch):
    reps = 300 if batch <= 256 else 60
    for _ in range(20):
        eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
    eng.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


t_end = time.perf_counter() + 0.4
while time.perf_counter() < t_end:
    timed(1)
for batch in batches:
    rows = fit._fit_rows(cases.halton_params(max(batch, 2), with_beta=beta), fit.model)[:batch]
    eng.upload(bufs[0], rows)
    line = f"{which} batch {batch:5d}: default {timed(batch):7.1f} ({eng.last_kernel()[10:15]})"
    _native.set_knob("VICTOR_HIP_MAPPING", "point")
    line += f" | point {timed(batch):7.1f}"
    _native.set_knob("VICTOR_HIP_MAPPING", "cells")
    for parts in (1, 2, 4, 6, 8, 12, 16):
        _native.set_knob("VICTOR_HIP_CELLS_PARTS", parts)
        line += f" | cells/{parts} {timed(batch):7.1f}"
    _native.set_knob("VICTOR_HIP_CELLS_PARTS", None)
    with_nofuse = ""
    _native.set_knob("VICTOR_HIP_NO_FUSE", "1")
    with_nofuse = f" | cells auto, separate chi2 {timed(batch):7.1f}"
    _native.set_knob("VICTOR_HIP_NO_FUSE", None)
    line += f" | cells auto {timed(batch):7.1f}" + with_nofuse
    _native.set_knob("VICTOR_HIP_MAPPING", None)
    print(line, flush=True)
