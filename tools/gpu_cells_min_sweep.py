"""Crossover between the point-major and the cells kernel: microseconds per resident launch at 8 ... 96 points through each
(VICTOR_HIP_MAPPING=point | cells), config 3 and BOSS - the data behind `cells_min` in launch_theory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases
from victor_amd import _native

for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    eng = fit._get_engine()
    o = eng.make_opts(fit.model, fit.fit_options)
    rows = fit._fit_rows(cases.halton_params(128, with_beta=beta), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(128), eng.alloc(128), eng.alloc(128 * eng.n_data)]
    eng.upload(bufs[0], rows)
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        eng.eval_device_async(o, bufs[0], 8, bufs[1], bufs[2], bufs[3]); eng.sync()
    for n in (8, 12, 16, 20, 24, 28, 32, 40, 48, 64, 96):
        line = f"{name} {n:3d} points:"
        for mapping in ("point", "cells"):
            _native.set_knob("VICTOR_HIP_MAPPING", mapping)
            best = 1e9
            for _ in range(3):
                for _ in range(30):
                    eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3])
                eng.sync()
                t0 = time.perf_counter()
                for _ in range(300):
                    eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3])
                eng.sync()
                best = min(best, (time.perf_counter() - t0) / 300)
            line += f"  {mapping} {best * 1e6:7.2f} us ({eng.last_kernel()[10:15]})"
        _native.set_knob("VICTOR_HIP_MAPPING", None)
        print(line, flush=True)
