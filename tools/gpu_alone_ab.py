"""Same-box A/B of the single-point launch: the ALONE instantiation of the point-major kernel (256 registers, 16 rows of the
quadratic form in flight in the fused tail; default whenever every workgroup of a launch fits on the chip at once) against
the ordinary one (VICTOR_HIP_NO_ALONE=1).  Resident launches (microseconds per launch, back to back) and CCFFit.log_likelihood."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from tests import cases
from victor_amd import _native

for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    p = cases.point(cases.halton_params(8, with_beta=beta), 3)
    eng, _, _, _, o = fit._single_point_plan()
    rows = fit._fit_rows({k: np.array([v]) for k, v in p.items()}, fit.model)
    d = [eng.alloc(rows.size), eng.alloc(1), eng.alloc(1), eng.alloc(eng.n_data)]
    eng.upload(d[0], rows)
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        fit.log_likelihood(p)
    ref = None
    for rnd in range(3):
        for knob in (None, "1"):
            _native.set_knob("VICTOR_HIP_NO_ALONE", knob)
            for _ in range(300):
                fit.log_likelihood(p)
            t0 = time.perf_counter()
            for _ in range(3000):
                val = fit.log_likelihood(p)
            api = (time.perf_counter() - t0) / 3000
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(2000):
                eng.eval_device_async(o, d[0], 1, d[1], d[2], d[3])
            eng.sync()
            res = (time.perf_counter() - t0) / 2000
            ref = ref or val
            print(f"{name} round {rnd} {'ordinary instantiation' if knob else 'ALONE instantiation   '}: resident launch {res * 1e6:.2f} us, "
                  f"log_likelihood {api * 1e6:.2f} us, chi2 {val[1]!r} (rel. deviation from the first {abs(val[1] / ref[1] - 1):.1e})", flush=True)
    _native.set_knob("VICTOR_HIP_NO_ALONE", None)
