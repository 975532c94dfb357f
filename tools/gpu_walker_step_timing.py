"""Where a lock-step walker step spends its time (BOSS cobaya configuration, 8 / 64 walkers): the sampler's host work, the
row building, the library call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from victor_amd.sampler import EnsembleMetropolis, parse_cobaya_params
from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
info = cases.cobaya_info()
lk = info["likelihood"]["CCFLikelihood"]
os.chdir(ROOT)
fit = victor_amd.CCFFit(lk["model"], lk["data"])
specs, fixed = parse_cobaya_params(info["params"])
for walkers in (8, 64):
    calls = []
    def ev(batch):
        t0 = time.perf_counter()
        out = fit.log_likelihood_batch(batch)[0]
        calls.append((time.perf_counter() - t0, len(out)))
        return out
    ens = EnsembleMetropolis(ev, specs, walkers, seed=1, fixed=fixed)
    ens.initialise(); ens.run(20); calls.clear()
    t0 = time.perf_counter(); ens.run(300); dt = time.perf_counter() - t0
    tc = np.array([c[0] for c in calls]); nn = np.array([c[1] for c in calls])
    print(f"{walkers} walkers: {dt/300*1e6:.1f} us per step; likelihood call {tc.mean()*1e6:.1f} us (median {np.median(tc)*1e6:.1f}, max {tc.max()*1e6:.0f}), "
          f"batch sizes {nn.min()}..{nn.max()}, calls slower than 200 us: {(tc > 200e-6).sum()}")
    # the same rows through the array API
    hp = cases.halton_params(walkers, with_beta=True)
    rows = fit._fit_rows(hp, fit.model)
    for _ in range(50): fit.log_likelihood_batch(rows)
    t0 = time.perf_counter()
    for _ in range(500): fit.log_likelihood_batch(rows)
    print(f"   array API, {walkers} rows: {(time.perf_counter()-t0)/500*1e6:.1f} us per call")
    b = {k: hp[k] for k in hp}
    t0 = time.perf_counter()
    for _ in range(500): fit._fit_rows(b, fit.model)
    print(f"   _fit_rows(dict of {walkers}): {(time.perf_counter()-t0)/500*1e6:.1f} us")
