"""Derived bounds for comparing two CORRECT evaluations of the same likelihood that differ in the order of their arithmetic
(another kernel mapping, another work split, a sub-batch on its own) - not a parity tolerance: parity against the oracle and
the reference's goldens is asserted with the contract's own numbers in the parity tests.

Why not a bare ``1e-12``: two mappings produce theory vectors that differ by a few rounding errors of their projection sums,
``|dt_k| <= u tau_k`` with ``tau_k = sum_i |W_l[i]| max(1 + xi)`` the magnitude of what is summed for entry k, and the
chi-square amplifies that by its conditioning.  With r = t - d and P the precision matrix,

    |d chi2|  <=  2 u  sum_jk |P_jk| |r_j| tau_k   +   u_s  sum_jk |P_jk| |r_j| |r_k|        (first order)

where the second term is the re-association of the quadratic form itself.  Both sums are computed here from the fit's own
arrays, per point, so the bound follows the point: a parameter set whose chi-square is a small difference of large terms is
allowed the error it must have, a well-conditioned one is held to a few 1e-14.  ``ulps`` is the number of unit roundoffs
(2^-53) the entries of the theory vector may differ by, relative to tau: 64 for two mappings of the same arithmetic (sums of
5000 terms in another order: ~sqrt(5000) u typical; an integrand whose exponent moved by one rounding: y^2 u <= 18 u), 1024
between the fast kernels and the generic one (library sqrt / exp / division against the refined hardware forms, each within
2 ulp - vk_devmath.h - over a chain of ~10 operations).

Every comparison appends its worst margin (observed difference / bound) to ``gpurun_out/tolerance_margins.txt`` when that
directory exists, so that a bound sitting close to what is observed is visible after a GPU run.
"""

import os

import numpy as np

U = 2.0 ** -53
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(what, margin):
    d = os.path.join(_ROOT, "gpurun_out")
    if os.path.isdir(d):
        try:
            with open(os.path.join(d, "tolerance_margins.txt"), "a") as fh:
                fh.write(f"{margin:.3e}  {what}\n")
        except OSError:
            pass


def _tau(fit):
    """Magnitude of the sums behind each entry of the theory vector: sum_i |W_l[i]| * 2 (1 + xi^r stays below 2)."""
    from victor_amd import tables as T
    poles = np.atleast_1d(fit.poles_s)
    w = T.projection_weights(T.mu_nodes_for(poles), poles)               # (n_ell, n_mu)
    return np.repeat(2.0 * np.abs(w).sum(axis=1), len(fit.s))            # (N,)


def chi2_bound(fit, params, ulps=64, **kwargs):
    """Per-point bound on |chi2_a - chi2_b| for two evaluation orders (see the module docstring).  ``params`` as for
    ``log_likelihood_batch``; ``kwargs`` are the model options of the call being compared.  Needs the GPU (theory vectors)."""
    from victor_amd import _native as N
    model = fit._merged(kwargs)
    rows = fit._fit_rows(params, model)
    t = fit.theory_vector_batch(rows, **kwargs)
    n, nd = t.shape
    beta = rows[:, N.P_BETA]
    if fit.fixed_data:
        d = np.broadcast_to(fit.multipole_datavector(), (n, nd))
    else:
        d = np.array([fit.multipole_datavector(b) for b in beta])
    absr = np.abs(t - d)
    v = absr + 2.0 * _tau(fit)[None, :]
    if fit.fixed_covmat:
        amp = np.einsum("ij,jk,ik->i", absr, np.abs(fit.icov), v)
    else:
        absP = np.abs(fit.icov)
        amp = np.empty(n)
        for i in range(n):
            lo, w = fit._bracket(beta[i]) if np.isfinite(beta[i]) else (0, 0.0)
            P = absP[lo] if w == 0.0 else (1 - w) * absP[lo] + w * absP[-1]
            amp[i] = absr[i] @ P @ v[i]
    return ulps * U * amp


def generic_chi2_bound(chi2, n_data, ulps=64):
    """Bound without the fit's arrays: ``ulps * u * n_data * 16`` relative - a conditioning budget of 16 (the entries of a
    chi-square of n_data terms, theory sums a few times the size of the residuals)."""
    return ulps * U * n_data * 16.0 * np.abs(np.asarray(chi2, float))


def assert_same_chi2(got, want, bound=None, n_data=None, what="", ulps=64):
    """chi-squares of two evaluation orders agree to ``bound`` (per point, from :func:`chi2_bound`; sliced by the caller for a
    sub-batch) or, without it, to :func:`generic_chi2_bound`.  Rows that failed in both (+inf) count as equal.  Returns the
    bound it used."""
    got, want = np.atleast_1d(np.asarray(got, float)), np.atleast_1d(np.asarray(want, float))
    assert got.shape == want.shape, (what, got.shape, want.shape)
    both_inf = np.isinf(got) & np.isinf(want) & (got == want)
    with np.errstate(invalid="ignore"):
        diff = np.where(both_inf, 0.0, np.abs(got - want))
    if bound is None:
        assert n_data, "assert_same_chi2 needs a bound or n_data"
        bound = generic_chi2_bound(np.maximum(np.abs(want), np.abs(got)), n_data, ulps)
    bound = np.broadcast_to(np.asarray(bound, float), diff.shape)
    bound = np.where(np.isfinite(bound), bound, 0.0)
    safe = np.where(bound > 0, bound, 1.0)
    ratio = np.where(bound > 0, diff / safe, np.where(diff > 0, np.inf, 0.0))
    margin = float(np.max(ratio)) if diff.size else 0.0
    _record(f"chi2 {what}", margin)
    assert margin <= 1.0, (what, "worst margin %.3g at index %d" % (margin, int(np.argmax(ratio))), float(np.max(diff)))
    return bound


def assert_same_lnl(got, want, chi2_bounds, what="", offset_scale=1000.0):
    """log-likelihoods of two evaluation orders: every likelihood form is a function of chi2 with |d lnL / d chi2| <= 0.51
    (gaussian -1/2; hartlap / percival rescale by a factor below one; sellentin -n/2(n-1) / (1 + chi2/(n-1))), plus a
    beta-dependent log-determinant of magnitude <= ``offset_scale`` that carries a few roundings of its own.  ABSOLUTE bound:
    lnL passes through zero where the log-determinant term cancels the chi-square term."""
    got, want = np.atleast_1d(np.asarray(got, float)), np.atleast_1d(np.asarray(want, float))
    both_inf = np.isinf(got) & np.isinf(want) & (got == want)
    with np.errstate(invalid="ignore"):
        diff = np.where(both_inf, 0.0, np.abs(got - want))
    bound = 0.51 * np.atleast_1d(np.asarray(chi2_bounds, float)) + 16 * U * (np.abs(want) + offset_scale)
    bound = np.where(np.isfinite(bound), bound, 0.0)
    margin = float(np.max(diff / np.where(bound > 0, bound, 1.0))) if diff.size else 0.0
    _record(f"lnl  {what}", margin)
    assert np.all(diff <= bound), (what, "worst margin %.3g" % margin, float(diff.max()))


def assert_same_theory(got, want, what="", ulps=512, tau=2.0):
    """Theory vectors / multipoles of two evaluation orders: entries differ by at most ``ulps`` unit roundoffs of the summed
    magnitude tau * sum|W| (see the module docstring); W-sums of the shipped grids are below 4."""
    got, want = np.asarray(got, float), np.asarray(want, float)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bound = ulps * U * tau * 4.0 * max(1.0, float(np.max(np.abs(want))) if want.size else 1.0)
    diff = float(np.max(np.abs(got - want))) if got.size else 0.0
    _record(f"theory {what}", diff / bound)
    assert diff <= bound, (what, diff, bound)
