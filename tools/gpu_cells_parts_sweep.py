"""Cells kernel: ranges per point (VICTOR_HIP_CELLS_PARTS) against batch size, resident buffers - the data behind the launch
planner's choice of `parts` (victor_hip.hip: plan_cells_parts).  Prints microseconds per batch, the default first.
Usage: gpu_cells_parts_sweep.py [config3|boss] [batches...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases
from victor_amd import _native

which = sys.argv[1] if len(sys.argv) > 1 else "config3"
batches = [int(b) for b in sys.argv[2:]] or [128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 12288, 16384]
opts, beta = (cases.boss_options("config"), True) if which == "boss" else (cases.synth_options(3), False)
fit = victor_amd.CCFFit(*opts)
eng = fit._get_engine()
o = eng.make_opts(fit.model, fit.fit_options)
cap = max(batches)
bufs = [eng.alloc(cap * 12), eng.alloc(cap), eng.alloc(cap), eng.alloc(cap * eng.n_data)]
rows = fit._fit_rows(cases.halton_params(cap, with_beta=beta), fit.model)
eng.upload(bufs[0], rows)


def timed(batch):
    reps = max(3, min(200, int(20000 / batch)))
    for _ in range(3):
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
    timed(64)
_native.set_knob("VICTOR_HIP_MAPPING", "cells")
parts = [None, "1", "2", "3", "4", "5", "6", "8", "12", "16"]
print(f"{which}: us per batch; columns = parts " + " ".join(str(p or "default") for p in parts), flush=True)
for batch in batches:
    line = f"batch {batch:6d}:"
    for p in parts:
        _native.set_knob("VICTOR_HIP_CELLS_PARTS", p)
        line += f" {timed(batch):9.1f}"
    _native.set_knob("VICTOR_HIP_CELLS_PARTS", None)
    print(line, flush=True)
