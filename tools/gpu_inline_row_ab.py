"""Same-box A/B of the single-point host call: parameter row in the kernel arguments (default) against the row read from the
pinned host buffer (VICTOR_HIP_NO_INLINE_ROW=1).  Prints microseconds per CCFFit.log_likelihood call, three rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases
from victor_amd import _native

for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    p = cases.point(cases.halton_params(8, with_beta=beta), 3)
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        fit.log_likelihood(p)
    for rnd in range(3):
        for knob in (None, "1"):
            _native.set_knob("VICTOR_HIP_NO_INLINE_ROW", knob)
            for _ in range(300):
                fit.log_likelihood(p)
            t0 = time.perf_counter()
            for _ in range(3000):
                fit.log_likelihood(p)
            dt = (time.perf_counter() - t0) / 3000
            print(f"{name} round {rnd} {'row from the pinned buffer' if knob else 'row in the kernel arguments'}: {dt * 1e6:.2f} us per call", flush=True)
    _native.set_knob("VICTOR_HIP_NO_INLINE_ROW", None)
