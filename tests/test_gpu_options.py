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


@pytest.mark.parametrize("with_beta", [False, True])
def test_rmu_real_space_input_on_gpu(tmp_path, with_beta):
    """format: rmu real-space input (with simulation_number) through the kernels vs the oracle."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_amd
    import victor_oracle as vo
    from tests.test_host import _rmu_inputs
    model, _ = _rmu_inputs(tmp_path, with_beta)
    m = victor_amd.CCFModel(model)
    o = vo.OracleModel(model)
    s = np.linspace(4, 110, 25)
    for p in ({"fsigma8": 0.5, "beta": 0.33, "sigma_v": 350, "epsilon": 1.02},
              {"fsigma8": 0.4, "beta": 0.55, "sigma_v": 420, "aperp": 0.95, "apar": 1.04}):
        got = m.theory_multipoles(s, dict(p), poles=[0, 2, 4])
        want, _x = o.theory_multipoles(s, dict(p), poles=[0, 2, 4])
        for key in ("0", "2", "4"):
            assert np.max(np.abs(got[key] - want[key])) < RTOL * np.max(np.abs(want[key])), (with_beta, key)


@pytest.mark.parametrize("non_uniform_mu", [False, True])
def test_anisotropic_sigma_v_template_on_gpu(tmp_path, non_uniform_mu):
    """3-key sigma_v(r, mu) template (bicubic, box-clamped) through the generic kernels vs the oracle."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_amd
    import victor_oracle as vo
    from tests.test_host import _aniso_inputs
    model, data = _aniso_inputs(tmp_path, non_uniform_mu)
    fit = victor_amd.CCFFit(model, data)
    ora = vo.OracleFit(model, data)
    hp = cases.halton_params(40)
    for i, kw in ((3, {}), (11, {"rsd_model": "dispersion"}), (29, {"assume_isotropic": True})):
        p = cases.point(hp, i)
        got = fit.log_likelihood(dict(p), **kw)
        want = ora.log_likelihood(dict(p), **kw)
        assert abs(got[1] / want[1] - 1) < RTOL and abs(got[0] / want[0] - 1) < RTOL, (i, kw)
    assert fit._get_engine().last_kernel() == "vk_theory_kernel"          # not a fast-path configuration


def test_velocity_template_mean_model_on_gpu(tmp_path):
    """velocity_pdf.mean.model = 'template' (ccf_model.py:439-443, 483-490) for every RSD mapping vs the oracle."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_amd
    import victor_oracle as vo
    from tests.test_host import _aniso_inputs
    model, data = _aniso_inputs(tmp_path)
    model["velocity_pdf"]["mean"]["model"] = "template"
    model["velocity_pdf"]["dispersion"] = {"model": "template", "template_keys": ["rsv", "sigmav"]}
    fit = victor_amd.CCFFit(model, data)
    ora = vo.OracleFit(model, data)
    hp = cases.halton_params(40)
    for i, kw in ((5, {}), (6, {"rsd_model": "dispersion"}), (7, {"rsd_model": "kaiser"}),
                  (8, {"rsd_model": "euclid_special"}), (9, {"empirical_corr": True})):
        p = dict(cases.point(hp, i), M=1.05, Q=0.9, Av=0.4)
        t = fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s, **kw)
        to = ora.theory_multipole_vector(ora.s, dict(p), ora.poles_s, **kw)
        assert np.max(np.abs(t - to)) < RTOL * np.max(np.abs(to)), (i, kw)
    # a per-call switch back to the linear mean model uses the other table set
    t = fit.theory_multipole_vector(fit.s, cases.point(hp, 5), fit.poles_s, mean_model="linear")
    to = ora.theory_multipole_vector(ora.s, cases.point(hp, 5), ora.poles_s, mean_model="linear")
    assert np.max(np.abs(t - to)) < RTOL * np.max(np.abs(to))
