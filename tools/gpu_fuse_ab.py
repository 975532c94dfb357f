"""chi-square fused into the cells kernel against the two-launch path, same box, resident, batch 65536
(and 8192 / 1024 for reference); config 3 and BOSS.  ms per batch including the likelihood stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases
from victor_amd import _native

for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    eng = fit._get_engine()
    o = eng.make_opts(fit.model, fit.fit_options)
    nmax = 65536
    rows = fit._fit_rows(cases.halton_params(nmax, with_beta=beta), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(nmax), eng.alloc(nmax), eng.alloc(nmax * eng.n_data)]
    eng.upload(bufs[0], rows)
    _native.set_knob("VICTOR_HIP_MAPPING", "cells")
    for batch in ([int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1024, 8192, 65536)):
        line = f"{name} cells kernel, batch {batch:6d}:"
        for rnd in range(2):
            for tag, knob in (("two launches", {"VICTOR_HIP_NO_FUSE": "1"}), ("fused", {"VICTOR_HIP_FUSE_MAX": "10000000"})):
                for k, v in knob.items():
                    _native.set_knob(k, v)
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.3:
                    eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3]); eng.sync()
                reps = 6 if batch > 10000 else 40
                t0 = time.perf_counter()
                for _ in range(reps):
                    eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
                eng.sync()
                dt = (time.perf_counter() - t0) / reps
                line += f"  {tag} {dt*1e3:8.3f} ms"
                for k in knob:
                    _native.set_knob(k, None)
        print(line, flush=True)
    _native.set_knob("VICTOR_HIP_MAPPING", None)
