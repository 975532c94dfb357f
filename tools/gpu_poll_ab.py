"""Same-box A/B of the hand-off inside a split single-point launch: polling (default) against the completion counters
(VICTOR_HIP_NO_POLL=1).  Prints microseconds per CCFFit.log_likelihood call and per resident launch, three rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VICTOR_HIP_DEV"] = "1"
import numpy as np
import victor_amd
from tests import cases
from victor_amd import _native

for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    hp = cases.halton_params(8, with_beta=beta)
    p = cases.point(hp, 3)
    eng = fit._get_engine()
    model = fit._merged({})
    o = eng.make_opts(model, fit.fit_options)
    rows = fit._fit_rows(hp, model)
    bufs = [eng.alloc(rows.size), eng.alloc(8), eng.alloc(8), eng.alloc(8 * eng.n_data)]
    eng.upload(bufs[0], rows)
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        fit.log_likelihood(p)
    for rnd in range(3):
        for knob in (None, "1"):
            _native.set_knob("VICTOR_HIP_NO_POLL", knob)
            for _ in range(300):
                fit.log_likelihood(p)
            t0 = time.perf_counter()
            for _ in range(3000):
                fit.log_likelihood(p)
            dt = (time.perf_counter() - t0) / 3000
            res = {}
            for n in (1, 2, 4):
                for _ in range(50):
                    eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3])
                eng.sync()
                t0 = time.perf_counter()
                for _ in range(2000):
                    eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3])
                eng.sync()
                res[n] = (time.perf_counter() - t0) / 2000 * 1e6
            print(f"{name} round {rnd} {'completion counters' if knob else 'polling            '}: log_likelihood {dt * 1e6:.2f} us per call; "
                  + "  ".join(f"{n} resident point(s) {v:.2f} us" for n, v in res.items()) + f"   [{eng.last_kernel()}]", flush=True)
    _native.set_knob("VICTOR_HIP_NO_POLL", None)
