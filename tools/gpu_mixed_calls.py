"""Soak of the host-buffer entry point: 20 s (or argv[1] seconds; argv[2] = config3 for the fixed-covariance workload) of back-to-back calls of random size (1 ... 4096 rows: polled in-place calls,
synchronised in-place calls, the copy path) at random offsets of one parameter table, each compared with the same rows of
one large batch.  Prints the number of calls and of mismatches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from tests import cases

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
boss = not (len(sys.argv) > 2 and sys.argv[2] == "config3")
fit = victor_amd.CCFFit(*(cases.boss_options("config") if boss else cases.synth_options(3)))
rows = fit._fit_rows(cases.halton_params(5000, with_beta=boss), fit.model)
ref = fit.log_likelihood_batch(rows)
rng = np.random.default_rng(1)
t0 = time.perf_counter()
calls = bad = 0
last = t0
while time.perf_counter() - t0 < seconds:
    n = int(rng.choice([1, 1, 1, 2, 7, 33, 64, 200, 256, 257, 600, 4096]))
    o = int(rng.integers(0, 5000 - n))
    l, c = fit.log_likelihood_batch(rows[o:o + n])
    calls += 1
    if not (np.allclose(l, ref[0][o:o + n], rtol=1e-11) and np.allclose(c, ref[1][o:o + n], rtol=1e-11)):
        bad += 1
    if time.perf_counter() - last > 60:
        last = time.perf_counter()
        print(f"  ... {calls} calls, {bad} mismatches", flush=True)
print(f"mixed-size host calls ({'boss' if boss else 'config3'}): {calls} in {seconds:.0f} s ({calls / seconds:.0f} calls/s), mismatches: {bad}")
