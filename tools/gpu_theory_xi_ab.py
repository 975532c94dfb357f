"""CCFModel.theory_xi (vk_xi_smu_batch) on the cells kernel against the generic kernel it used until round 5: wall time per call of
theory_xi_batch (host buffers in, [n][100][n_s] out - the copies are the same on both sides) for 1, 64 and 1024 points, config 3
and BOSS, every RSD model; and the largest difference between the two, relative to max |xi|.
Usage: python tools/gpu_theory_xi_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import victor_amd
import workloads as cases
from victor_amd import _native


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


mu = np.linspace(0, 1, 100)
print(f"{'workload':10s} {'rsd':15s} {'n':>5s} {'cells':>12s} {'generic':>12s} {'ratio':>6s} {'max rel diff':>13s}")
for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    hp = cases.halton_params(1024, with_beta=beta)
    for rsd in ("streaming", "dispersion", "kaiser", "euclid_special"):
        for n in (1, 64, 1024):
            sub = {k: v[:n] for k, v in hp.items()}
            reps = 200 if n == 1 else (40 if n == 64 else 5)
            call = lambda: fit.theory_xi_batch(fit.s, mu, sub, rsd_model=rsd)      # noqa: E731
            t_c = timed(call, reps)
            xi_c = call()
            k_c = fit._get_engine().last_kernel()
            _native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
            try:
                t_g = timed(call, reps)
                xi_g = call()
                k_g = fit._get_engine().last_kernel()
            finally:
                _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
            assert k_c == "vk_theory_cells_kernel" and k_g == "vk_xi_smu_kernel", (k_c, k_g)
            unit = 1e6 if n == 1 else 1e3
            u = "us" if n == 1 else "ms"
            print(f"{name:10s} {rsd:15s} {n:5d} {t_c * unit:9.2f} {u} {t_g * unit:9.2f} {u} {t_g / t_c:6.2f} "
                  f"{np.max(np.abs(xi_c - xi_g)) / np.max(np.abs(xi_g)):13.2e}", flush=True)
