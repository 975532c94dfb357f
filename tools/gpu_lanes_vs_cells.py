"""Large batches with batch-constant tables (configs 2, 3): lanes kernel against cells kernel, same box, resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _devlib import use_dev_library
use_dev_library()          # the lanes kernel lives in the development build of the library only
import victor_amd
from tests import cases
from victor_amd import _native

for config in (3, 2):
    fit = victor_amd.CCFFit(*cases.synth_options(config))
    eng = fit._get_engine()
    o = eng.make_opts(fit.model, fit.fit_options)
    nmax = 262144
    rows = fit._fit_rows(cases.halton_params(nmax), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(nmax), eng.alloc(nmax), eng.alloc(nmax * eng.n_data)]
    eng.upload(bufs[0], rows)
    for batch in (8192, 16384, 32768, 65536, 131072, 262144):
        line = f"config {config} batch {batch:6d}:"
        for rnd in range(2):
            for mapping in ("lanes", "cells"):
                _native.set_knob("VICTOR_HIP_MAPPING", mapping)
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.3:
                    eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3]); eng.sync()
                reps = max(3, int(0.4 / (batch / 2.4e6)))
                t0 = time.perf_counter()
                for _ in range(reps):
                    eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
                eng.sync()
                dt = (time.perf_counter() - t0) / reps
                line += f"  {mapping} {batch/dt/1e6:6.3f} M/s"
        _native.set_knob("VICTOR_HIP_MAPPING", None)
        print(line, flush=True)
