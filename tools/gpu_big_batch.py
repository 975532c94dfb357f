"""One-million-point batches through the host-buffer API (index arithmetic beyond 2^31 integrand points per launch, scratch growth); the timings include engine creation."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import victor_amd
from tests import cases
fit = victor_amd.CCFFit(*cases.synth_options(3))
n = 1 << 20
hp = cases.halton_params(n)
t0 = time.perf_counter(); lnl, chi2 = fit.log_likelihood_batch(hp); dt = time.perf_counter() - t0
print("n", n, "time", round(dt, 3), "s ->", round(n / dt), "evals/s; finite", bool(np.all(np.isfinite(lnl))))
idx = np.r_[0:5, n - 5:n, 777777]
sub = fit.log_likelihood_batch({k: v[idx] for k, v in hp.items()})
print("max rel dev vs standalone:", np.max(np.abs(sub[1] / chi2[idx] - 1)))
boss = victor_amd.CCFFit(*cases.boss_options("config"))
hpb = cases.halton_params(n, with_beta=True)
t0 = time.perf_counter(); lnl, chi2 = boss.log_likelihood_batch(hpb); dt = time.perf_counter() - t0
print("boss n", n, "time", round(dt, 3), "s ->", round(n / dt), "evals/s; finite", bool(np.all(np.isfinite(lnl))))
sub = boss.log_likelihood_batch({k: v[idx] for k, v in hpb.items()})
print("max rel dev vs standalone:", np.max(np.abs(sub[1] / chi2[idx] - 1)))
