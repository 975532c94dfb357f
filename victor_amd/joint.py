"""Joint fit of several data vectors that share one parameter vector (density-split quantiles).

The reference mentions density-split centres only in passing (``ccf_model.py:28-30``) and has no joint-fit
class; BASELINE config "density-split 5-quantile joint fit" is defined as Q independent ``CCFFit`` blocks with a
block-diagonal covariance, so chi-square and log-likelihood add.  Each block keeps its own context (tables in
HBM) on the same GPU; one batch of parameter rows is run through every block.
"""

import numpy as np


class JointFit:
    def __init__(self, fits):
        self.fits = list(fits)
        if not self.fits:
            raise ValueError("need at least one fit")

    def log_likelihood_batch(self, params, **kwargs):
        lnl = chi2 = None
        for fit in self.fits:
            a, b = fit.log_likelihood_batch(params, **kwargs)
            lnl = a if lnl is None else lnl + a
            chi2 = b if chi2 is None else chi2 + b
        bad = ~np.isfinite(lnl)
        lnl[bad], chi2[bad] = -np.inf, np.inf
        return lnl, chi2

    def log_likelihood(self, params, **kwargs):
        lnl, chi2 = self.log_likelihood_batch(params, **kwargs)
        return float(lnl[0]), float(chi2[0])

    @property
    def n_data(self):
        return sum(len(f.s) * len(f.poles_s) for f in self.fits)
