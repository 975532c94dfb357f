"""Walker ensembles on the BOSS cobaya configuration: the step loop inside the library (vk_walk_run), one step per launch and two
steps per launch (speculate), against the Python loop (native=False), same box, interleaved.  us per step and evaluations/s.
Usage: python tools/gpu_walker_native_ab.py [walkers ...]"""
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
for walkers in [int(a) for a in sys.argv[1:]] or (8, 16, 32, 64, 128, 512):
    kinds = {"python": dict(native=False), "library": dict(native=True, speculate=False), "two": dict(native=True, speculate=True)}
    ens = {k: EnsembleMetropolis(None, specs, walkers, seed=1, fixed=fixed, fit=fit, **kw).initialise() for k, kw in kinds.items()}
    best = {}
    for k in kinds:                        # past the runtime's one-off stall (~65 ms, some tens of ms after fresh allocations)
        t_end = time.perf_counter() + 0.5
        while time.perf_counter() < t_end:
            ens[k].run(64)
    for rnd in range(5):
        for k in kinds:
            e0 = ens[k].n_evals
            t0 = time.perf_counter()
            ens[k].run(640)
            dt = time.perf_counter() - t0
            rate = (ens[k].n_evals - e0) / dt
            if k not in best or dt < best[k][0]:
                best[k] = (dt, rate)
    # the same chain: fresh ensembles, the same seed, 2000 steps each
    fresh = {k: EnsembleMetropolis(None, specs, walkers, seed=7, fixed=fixed, fit=fit, **kw).initialise() for k, kw in kinds.items()}
    ends = {k: (fresh[k].run(2000)[0][-1], fresh[k].n_accept, fresh[k].n_evals) for k in kinds}
    same = all(np.array_equal(ends["python"][0], ends[k][0]) and ends["python"][1:] == ends[k][1:] for k in ("library", "two"))
    us = {k: 1e6 * best[k][0] / 640 for k in kinds}
    print(f"{walkers:4d} walkers: python loop {us['python']:7.2f} us/step {best['python'][1] / 1e3:8.1f} k evals/s | "
          f"library loop {us['library']:7.2f} us/step {best['library'][1] / 1e3:8.1f} k evals/s ({us['python'] / us['library']:.2f} x) | "
          f"two steps per launch {us['two']:7.2f} us/step {best['two'][1] / 1e3:8.1f} k evals/s ({us['python'] / us['two']:.2f} x) | "
          f"same positions and counters after 2000 steps of fresh ensembles: {same}", flush=True)
