"""Walker ensembles on the BOSS cobaya configuration: the step loop inside the library (vk_walk_run) against the Python loop
(native=False), same box, interleaved.  us per step and evaluations/s for 8, 64, 512 walkers.
Usage: python tools/gpu_walker_native_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import victor_amd
import workloads as cases
from victor_amd.sampler import EnsembleMetropolis, parse_cobaya_params

info = cases.cobaya_info()
lk = info["likelihood"]["CCFLikelihood"]
os.chdir(ROOT)
fit = victor_amd.CCFFit(lk["model"], lk["data"])
specs, fixed = parse_cobaya_params(info["params"])
for walkers in (8, 64, 512):
    ens = {flag: EnsembleMetropolis(None, specs, walkers, seed=1, fixed=fixed, fit=fit, native=flag).initialise() for flag in (False, True)}
    best = {}
    for flag in (False, True):             # the same number of steps on both sides: the two ensembles stay the same chain
        for _ in range(40):
            ens[flag].run(25)
    for rnd in range(5):
        for flag in (False, True):
            e0 = ens[flag].n_evals
            t0 = time.perf_counter()
            ens[flag].run(640)
            dt = time.perf_counter() - t0
            rate = (ens[flag].n_evals - e0) / dt
            if flag not in best or dt < best[flag][0]:
                best[flag] = (dt, rate)
    same = np.array_equal(ens[False].x, ens[True].x) and ens[False].n_accept == ens[True].n_accept
    print(f"{walkers:4d} walkers: python loop {1e6 * best[False][0] / 640:7.2f} us/step {best[False][1] / 1e3:8.1f} k evals/s | "
          f"library loop {1e6 * best[True][0] / 640:7.2f} us/step {best[True][1] / 1e3:8.1f} k evals/s | {best[False][0] / best[True][0]:.2f} x | "
          f"same positions after {ens[True].n_steps} steps: {same}", flush=True)
