#!/usr/bin/env python3
"""The likelihood calls of the reference's usage notebook (notebooks/victor_usage_demo.ipynb, cells around lines
480-512) run through this package: same configuration file layout, same calls, same printed numbers.

    python examples/usage_demo.py            # needs one MI355X; prints chi2 / lnL for the five model variants

The reference prints 65.01 / 284.76 (streaming), 65.03 / 284.76 (dispersion), 103.90 / 266.81 (kaiser),
64.39 / 285.06 (anisotropic real-space ccf; the current reference code with SciPy 1.15 gives 64.40 / 285.05, as does this
package) and 64.80 / 285.30 (beta interpolation of the likelihood).
"""

import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import yaml
    from victor import CCFFit

    os.chdir(ROOT)
    with open(os.path.join("config", "boss_config.yaml")) as fh:
        info = yaml.full_load(fh)
    fit = CCFFit(info["model"], info["data"])
    params = {"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0}
    variants = [("streaming model", {}),
                ("dispersion model", {"rsd_model": "dispersion"}),
                ("Kaiser model", {"rsd_model": "kaiser"}),
                ("streaming, anisotropic real-space ccf", {"assume_isotropic": False}),
                ("streaming, likelihood interpolated in beta", {"beta_interpolation": "likelihood"})]
    for label, kwargs in variants:
        lnl, chi2 = fit.log_likelihood(dict(params), **kwargs)
        print(f"{label:45s} chi2 = {chi2:7.2f}   lnL = {lnl:7.2f}")
    t0 = time.perf_counter()
    n = 2000
    for _ in range(n):
        fit.log_likelihood(dict(params))
    dt = (time.perf_counter() - t0) / n
    print(f"one log_likelihood call: {dt * 1e6:.0f} us (the reference needs about 75 ms)")
    multipoles = fit.theory_multipoles(fit.s, dict(params), poles=[0, 2])
    print("model monopole at the first / last s bin:", multipoles["0"][0], multipoles["0"][-1])


if __name__ == "__main__":
    main()
