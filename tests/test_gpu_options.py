"""linear_bias matter model, empirical velocity correction, real-space ccf measured from data.

Golden vectors come from the reference itself (oracle/make_golden.py, 'opt_*' keys)."""

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu
RTOL = 1e-9

OPT = {
    "lb_stream": dict(matter_model="linear_bias"),
    "lb_kaiser": dict(matter_model="linear_bias", rsd_model="kaiser"),
    "lb_disp": dict(matter_model="linear_bias", rsd_model="dispersion"),
    "emp_stream": dict(empirical_corr=True),
    "emp_disp": dict(empirical_corr=True, rsd_model="dispersion"),
    "emp_kaiser": dict(empirical_corr=True, rsd_model="kaiser"),
}


def close(a, b):
    return np.max(np.abs(a - b)) <= RTOL * np.max(np.abs(b))


@pytest.fixture(scope="module")
def gold():
    return cases.golden_outputs()


def test_boss_beta_dependent_tables(gold):
    import victor_amd
    g, meta = gold
    fit = victor_amd.CCFFit(*cases.boss_options("config"))
    pts = [dict(q, bias=2.1, Av=0.7, M=1.05, Q=0.95) for q in meta["boss_points"][:3]]
    for tag, kw in OPT.items():
        t = np.array([fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, **kw) for q in pts])
        assert close(t, g[f"opt_boss_{tag}"]), tag
    from victor_amd import InputError
    with pytest.raises(InputError):       # not implemented: needs products of beta polynomials
        fit.theory_multipole_vector(fit.s, dict(pts[0]), fit.poles_s, matter_model="linear_bias", empirical_corr=True)


def test_synthetic_fixed_tables(gold):
    import victor_amd
    g, meta = gold
    fit = victor_amd.CCFFit(*cases.synth_options(3))
    pts = [dict(q, beta=0.4, bias=1.7, Av=-0.5, M=1.1, Q=0.9) for q in meta["synth_points"][:3]]
    opt = dict(OPT, lb_emp_stream=dict(matter_model="linear_bias", empirical_corr=True),
               lb_emp_disp=dict(matter_model="linear_bias", empirical_corr=True, rsd_model="dispersion"))
    for tag, kw in opt.items():
        t = np.array([fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, **kw) for q in pts])
        assert close(t, g[f"opt_synth_{tag}"]), tag
    # linear_bias on fixed tables runs through the fast kernel: batch API and single calls agree
    batch = {k: np.array([p[k] for p in pts]) for k in pts[0]}
    tb = fit.theory_vector_batch(batch, matter_model="linear_bias")
    assert close(tb, g["opt_synth_lb_stream"])


def test_realspace_ccf_from_data_with_md_covariance(gold):
    """CMASS_..._measured_model + variable isotropic MD covariance (15-node beta grid that differs from the
    data's 31-node grid) - the combination the reference's data README prescribes (README.txt:50-58)."""
    import victor_amd
    g, meta = gold
    m, d = cases.boss_options("config")
    m["input_model_data_file"] = "boss/measured_model.npy"
    m["realspace_ccf"]["from_data"] = True
    d["covariance_matrix"]["data_file"] = "boss/cov_md_iso.npy"
    fit = victor_amd.CCFFit(m, d)
    assert fit.beta_covmat.shape == (15,) and fit.beta_ccf.shape == (31,)
    pts = meta["boss_points"]
    batch_rows = np.concatenate([fit._fit_rows(dict(p), fit.model) for p in pts])
    lnl, chi2 = fit.log_likelihood_batch(batch_rows)
    th = fit.theory_vector_batch(batch_rows)
    assert close(th, g["opt_fromdata_theory"])
    assert np.max(np.abs(chi2 / g["opt_fromdata_chi2"] - 1)) < RTOL
    assert np.max(np.abs(lnl / g["opt_fromdata_lnl"] - 1)) < RTOL
    t = np.array([fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, assume_isotropic=False) for q in pts[:3]])
    assert close(t, g["opt_fromdata_aniso_theory"])
    # growth term beta*bias: fsigma8 is not needed at all
    t = np.array([fit.theory_multipole_vector(fit.s, {k: v for k, v in dict(q, bias=2.0).items() if k != "fsigma8"},
                                              fit.poles_s, matter_model="linear_bias", rsd_model="kaiser")
                  for q in pts[:3]])
    assert close(t, g["opt_fromdata_lb_kaiser"])
