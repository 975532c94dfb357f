"""Wall-clock of the host-buffer Python API (includes parameter-row building, H2D/D2H copies and synchronisation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from tests import cases

for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    hp = cases.halton_params(65536, with_beta=beta)
    rows = fit._fit_rows(hp, fit.model)
    for _ in range(5):
        fit.log_likelihood_batch(rows)
    t0 = time.perf_counter()
    for _ in range(5):
        fit.log_likelihood_batch(rows)
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: log_likelihood_batch(65536 rows, host buffers) {dt*1e3:.2f} ms -> {65536/dt:.0f} evals/s")
    t0 = time.perf_counter()
    for _ in range(5):
        fit.log_likelihood_batch(hp)
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: same from a dict of arrays {dt*1e3:.2f} ms -> {65536/dt:.0f} evals/s")
    p = cases.point(hp, 3)
    for _ in range(200):
        fit.log_likelihood(dict(p))
    t0 = time.perf_counter()
    for _ in range(2000):
        fit.log_likelihood(dict(p))
    dt = (time.perf_counter() - t0) / 2000
    print(f"{name}: log_likelihood(single point) {dt*1e6:.1f} us per call -> {1/dt:.0f} evals/s")
