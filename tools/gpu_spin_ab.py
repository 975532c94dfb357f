"""In-place host-buffer calls: results polled for in pinned memory vs stream synchronisation (VICTOR_HIP_SPIN_MAX), wall us per
call through CCFFit.log_likelihood (one point) and CCFFit.log_likelihood_batch (8 / 64 / 256 rows)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from tests import cases
from victor_amd import _native

first = {}

for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    hp = cases.halton_params(256, with_beta=beta)
    rows = fit._fit_rows(hp, fit.model)
    p = cases.point(hp, 3)
    ref = fit.log_likelihood_batch(rows)
    for rnd in range(2):
        for spin in ("0", "64", "256"):
            _native.set_knob("VICTOR_HIP_SPIN_MAX", spin)
            out = []
            for _ in range(300):
                fit.log_likelihood(dict(p))
            t0 = time.perf_counter()
            for _ in range(3000):
                fit.log_likelihood(dict(p))
            out.append((time.perf_counter() - t0) / 3000 * 1e6)
            for n in (8, 64, 256):
                sub = np.ascontiguousarray(rows[:n])
                for _ in range(100):
                    got = fit.log_likelihood_batch(sub)
                t0 = time.perf_counter()
                for _ in range(1000):
                    got = fit.log_likelihood_batch(sub)
                out.append((time.perf_counter() - t0) / 1000 * 1e6)
                key = (name, n)
                if key in first:      # polled and synchronised calls run the same launch: identical bits
                    assert np.array_equal(got[0], first[key][0]) and np.array_equal(got[1], first[key][1])
                else:
                    first[key] = (np.array(got[0]), np.array(got[1]))
                    assert np.allclose(got[0], np.asarray(ref[0])[:n], rtol=1e-11) and np.allclose(got[1], np.asarray(ref[1])[:n], rtol=1e-11)
            print(f"{name} round {rnd} SPIN_MAX={spin:>3s}: 1 point {out[0]:6.1f} us   8 rows {out[1]:6.1f}   64 rows {out[2]:6.1f}   256 rows {out[3]:6.1f}", flush=True)
    _native.set_knob("VICTOR_HIP_SPIN_MAX", None)
