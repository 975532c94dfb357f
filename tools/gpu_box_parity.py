import sys, numpy as np
sys.path.insert(0, ".")
import victor_amd
from tests import cases
g, meta = cases.golden_outputs("box")
fit = victor_amd.CCFFit(*cases.boss_options("config"))
hp = cases.halton_params(meta["n"], with_beta=True)
print("GPU vs the reference itself, 48 Halton points of the cobaya prior box (tests/golden/ref_outputs_box.npz), BOSS configuration")
for rsd in ("streaming", "dispersion", "kaiser", "euclid_special"):
    lnl, chi = fit.log_likelihood_batch(hp, rsd_model=rsd)
    t = np.array([fit.theory_multipole_vector(fit.s, cases.point(hp, i), fit.poles_s, rsd_model=rsd) for i in range(meta["n"])])
    w = g[f"{rsd}_theory"]
    print(f"{rsd:15s} max rel dchi2 {np.max(np.abs(chi / g[rsd + '_chi2'] - 1)):.2e}   max |dlnL|/max(|lnL|,1) "
          f"{np.max(np.abs(lnl - g[rsd + '_lnl']) / np.maximum(np.abs(g[rsd + '_lnl']), 1)):.2e}   max |dxi_l| / max|xi_l| "
          f"{np.max(np.max(np.abs(t - w), axis=1) / np.max(np.abs(w), axis=1)):.2e}")
