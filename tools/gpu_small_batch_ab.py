"""Point-major vs cells kernel below the 1024-point switch (resident, config 3 and BOSS)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases
from victor_amd import _native
for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    eng = fit._get_engine()
    o = eng.make_opts(fit.model, fit.fit_options)
    bufs = [eng.alloc(1100 * 12), eng.alloc(1100), eng.alloc(1100), eng.alloc(1100 * eng.n_data)]
    for batch in (64, 128, 192, 256, 384, 512, 768, 1000):
        rows = fit._fit_rows(cases.halton_params(batch, with_beta=beta), fit.model)
        eng.upload(bufs[0], rows)
        line = f"{name} batch {batch:5d}:"
        for mapping in ("point", "cells"):
            _native.set_knob("VICTOR_HIP_MAPPING", mapping)
            for _ in range(20):
                eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(200):
                eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
            eng.sync()
            dt = (time.perf_counter() - t0) / 200
            line += f"  {mapping} {dt*1e3:7.3f} ms {batch/dt:9.0f} evals/s"
        _native.set_knob("VICTOR_HIP_MAPPING", None)
        print(line, flush=True)
