import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _devlib import use_dev_library
use_dev_library()          # the lanes kernel lives in the development build of the library only
import victor_amd
from tests import cases
from victor_amd import _native
for name, opts, beta in (("boss", cases.boss_options("config"), True), ("synth3", cases.synth_options(3), False), ("synth2", cases.synth_options(2), False)):
    fit = victor_amd.CCFFit(*opts)
    eng = fit._get_engine()
    o = eng.make_opts(fit.model, fit.fit_options)
    batch = 65536
    rows = fit._fit_rows(cases.halton_params(batch, with_beta=beta), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
    eng.upload(bufs[0], rows)
    for mapping in ("point", "cells", "lanes", "point", "cells", "lanes"):
        if mapping == "lanes" and beta: continue
        _native.set_knob("VICTOR_HIP_MAPPING", mapping)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3]); eng.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name} {mapping} ({eng.last_kernel()}): {dt*1e3:.2f} ms/batch {batch/dt:.0f} evals/s")
