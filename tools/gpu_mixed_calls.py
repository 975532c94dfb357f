"""Soak of the host-buffer entry point: 20 s of back-to-back calls of random size (1 ... 4096 rows: polled in-place calls,
synchronised in-place calls, the copy path) at random offsets of one parameter table, each compared with the same rows of
one large batch.  Prints the number of calls and of mismatches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from tests import cases

fit = victor_amd.CCFFit(*cases.boss_options("config"))
rows = fit._fit_rows(cases.halton_params(5000, with_beta=True), fit.model)
ref = fit.log_likelihood_batch(rows)
rng = np.random.default_rng(1)
t0 = time.perf_counter()
calls = bad = 0
while time.perf_counter() - t0 < 20:
    n = int(rng.choice([1, 1, 1, 2, 7, 33, 64, 200, 256, 257, 600, 4096]))
    o = int(rng.integers(0, 5000 - n))
    l, c = fit.log_likelihood_batch(rows[o:o + n])
    calls += 1
    if not (np.allclose(l, ref[0][o:o + n], rtol=1e-11) and np.allclose(c, ref[1][o:o + n], rtol=1e-11)):
        bad += 1
print(f"mixed-size host calls: {calls} in 20 s ({calls / 20:.0f} calls/s), mismatches: {bad}")
