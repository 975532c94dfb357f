"""The CPU oracle against the committed golden vectors (which came from the reference itself).

Runs everywhere (no GPU, no /root/reference).  This is what pins the oracle on the GPU box: the same
``oracle/victor_oracle.py`` that the ``-m gpu`` tests and ``bench.py`` compare the HIP path with must
reproduce the reference's outputs stored in ``tests/golden/ref_outputs.npz`` and the numbers printed in
the reference's notebook.
"""

import os
import sys

import numpy as np
import pytest

from tests import cases

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import victor_oracle as vo  # noqa: E402

TOL = 1e-12


@pytest.fixture(scope="module")
def gold():
    return cases.golden_outputs()


@pytest.fixture(scope="module")
def boss():
    return vo.OracleFit(*cases.boss_options("config"))


def test_init_tables_match_reference(boss, gold):
    g, _ = gold
    assert boss.iaH == float(g["boss_iaH"]) == 0.011421076289465598      # SURVEY.md 8(c) pin
    assert np.array_equal(boss.sv_rmu, g["boss_sv_rmu"])
    r_ext = np.append([0.01], boss.r)
    assert np.max(np.abs(boss.delta(r_ext) - g["boss_delta_ext"])) < 1e-15
    assert np.max(np.abs(boss.integrated_delta(r_ext) / g["boss_int_delta_ext"] - 1)) < 1e-13
    assert np.max(np.abs(boss.icov[12] - g["boss_icov_12"])) <= 1e-12 * np.max(np.abs(g["boss_icov_12"]))


def test_notebook_known_answers(boss, gold):
    """victor_usage_demo.ipynb:491-499 prints five (chi2, lnL) pairs at two decimals.  They were produced with
    SciPy < 1.11, whose ``simps`` treats the 50 velocity nodes with even='avg' (ccf_model.py:690): under that rule
    ALL FIVE reproduce at printed precision; under the SciPy >= 1.11 rule (the default here) four do and the
    anisotropic pair moves to 64.40 / 285.05."""
    g, _ = gold
    ga, _ = cases.golden_outputs("avg")
    p = cases.NOTEBOOK_POINT
    for name, ((chi_nb, lnl_nb), kw) in cases.NOTEBOOK_PRINTED.items():
        lnl, chi = boss.log_likelihood(dict(p), simpson_even="avg", **kw)
        assert round(chi, 2) == chi_nb and round(lnl, 2) == lnl_nb, name
        assert abs(chi - ga[f"boss_nb_{name}"][0]) < TOL * chi, name
        assert abs(lnl - ga[f"boss_nb_{name}"][1]) < TOL * abs(lnl), name
        lnl, chi = boss.log_likelihood(dict(p), **kw)
        assert abs(chi - g[f"boss_nb_{name}"][0]) < TOL * chi, name
        assert abs(lnl - g[f"boss_nb_{name}"][1]) < TOL * abs(lnl), name
        if name != "anisotropic":
            assert round(chi, 2) == chi_nb and round(lnl, 2) == lnl_nb, name
        else:
            assert round(chi, 2) == 64.40 and round(lnl, 2) == 285.05


def test_legacy_simpson_goldens(boss):
    """The oracle under simpson_even='avg' against the reference run with the SciPy < 1.11 ``simps`` stand-in
    (tests/golden/ref_outputs_avg.npz), and the size of the old-vs-new difference."""
    ga, meta = cases.golden_outputs("avg")
    g, _ = cases.golden_outputs()
    for i, p in enumerate(meta["boss_points"]):
        t = boss.theory_multipole_vector(boss.s, dict(p), boss.poles_s, simpson_even="avg")
        lnl, chi = boss.log_likelihood(dict(p), simpson_even="avg")
        assert np.max(np.abs(t - ga["boss_config_theory"][i])) < TOL
        assert abs(chi / ga["boss_config_chi2"][i] - 1) < TOL and abs(lnl / ga["boss_config_lnl"][i] - 1) < TOL
    for kw, key in ((dict(rsd_model="dispersion"), "boss_dispersion_theory"), (dict(assume_isotropic=False), "boss_aniso_theory")):
        t = np.array([boss.theory_multipole_vector(boss.s, dict(q), boss.poles_s, simpson_even="avg", **kw)
                      for q in meta["boss_points"][:3]])
        assert np.max(np.abs(t - ga[key])) < TOL
    xi = boss.theory_xi(boss.s, np.linspace(0, 1, 100), dict(meta["boss_points"][0]), simpson_even="avg")
    assert np.max(np.abs(xi - ga["boss_config_xi_smu_p0"])) < TOL
    for config in (2, 3):
        model, data = cases.synth_options(config)
        model["numerics"] = {"simpson_even": "scipy<1.11"}          # the construction-time spelling of the option
        fit = vo.OracleFit(model, data)
        pts = list(meta["synth_points"])
        if config == 3:
            pts = [{"fsigma8": 0.47, "sigma_v": 380, "aperp": 1.02, "apar": 0.97}] + pts
        for i in (0, 3, 8):
            t = fit.theory_multipole_vector(fit.s, dict(pts[i]), fit.poles_s)
            lnl, chi = fit.log_likelihood(dict(pts[i]))
            assert np.max(np.abs(t - ga[f"synth{config}_theory"][i])) < TOL
            assert abs(chi / ga[f"synth{config}_chi2"][i] - 1) < TOL
    # measured old-vs-new deltas: first order, not 1e-9 (the integrand has a cusp at r_par ~ 0)
    d_aniso = abs(ga["boss_nb_anisotropic"][0] / g["boss_nb_anisotropic"][0] - 1)
    d_iso = abs(ga["boss_nb_streaming"][0] / g["boss_nb_streaming"][0] - 1)
    d_xi3 = np.max(np.abs(ga["synth3_theory"] - g["synth3_theory"][:9])) / np.max(np.abs(g["synth3_theory"]))
    assert 1e-4 < d_aniso < 4e-4 and 1e-6 < d_iso < 3e-6 and 1e-5 < d_xi3 < 1e-4


@pytest.mark.parametrize("variant", ["config", "cobaya"])
def test_boss_points(gold, variant):
    g, meta = gold
    fit = vo.OracleFit(*cases.boss_options(variant))
    for i, p in enumerate(meta["boss_points"]):
        t = fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s)
        lnl, chi = fit.log_likelihood(dict(p))
        assert np.max(np.abs(t - g[f"boss_{variant}_theory"][i])) < TOL
        assert abs(chi / g[f"boss_{variant}_chi2"][i] - 1) < TOL
        assert abs(lnl / g[f"boss_{variant}_lnl"][i] - 1) < TOL


def test_survey_pins(boss):
    """SURVEY.md section 8(c) 'quick sanity pins' measured on the reference during the survey."""
    lnl, chi = boss.log_likelihood({"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0})
    assert abs(chi - 65.011658718810) < 1e-9 and abs(lnl - 284.764445204409) < 1e-9
    lnl, chi = boss.log_likelihood({"fsigma8": 0.55, "beta": 0.43, "sigma_v": 320, "epsilon": 1.04})
    assert abs(chi - 95.007453968916) < 1e-9 and abs(lnl - 271.142815675195) < 1e-9
    lnl, chi = boss.log_likelihood({"fsigma8": 0.30, "beta": 0.251, "sigma_v": 450, "aperp": 0.97, "apar": 1.05})
    assert abs(chi - 236.441581663657) < 1e-9 and abs(lnl - 210.241139535949) < 1e-9


def test_other_branches_and_forms(boss, gold):
    g, meta = gold
    pts = meta["boss_points"][:3]
    for rsd in ("dispersion", "kaiser", "euclid_special"):
        t = np.array([boss.theory_multipole_vector(boss.s, dict(q), boss.poles_s, rsd_model=rsd) for q in pts])
        assert np.max(np.abs(t - g[f"boss_{rsd}_theory"])) < TOL
    t = np.array([boss.theory_multipole_vector(boss.s, dict(q, M=1.1, Q=0.9), boss.poles_s, rsd_model="kaiser",
                                               kaiser_approximation=True) for q in pts])
    assert np.max(np.abs(t - g["boss_kaiser_approx_theory"])) < TOL
    for form in ("gaussian", "hartlap", "percival", "sellentin"):
        lnl, chi = boss.log_likelihood(dict(pts[0]), likelihood={"form": form, "nmocks": 1000, "nparams": 4})
        assert abs(lnl - g[f"boss_form_{form}"][1]) < TOL * abs(lnl)


@pytest.mark.parametrize("config", [2, 3])
def test_synthetic_points(gold, config):
    g, meta = gold
    fit = vo.OracleFit(*cases.synth_options(config))
    pts = list(meta["synth_points"])
    if config == 3:
        pts = [{"fsigma8": 0.47, "sigma_v": 380, "aperp": 1.02, "apar": 0.97}] + pts
    for i in (0, 1, 7, len(pts) - 1):
        t = fit.theory_multipole_vector(fit.s, dict(pts[i]), fit.poles_s)
        lnl, chi = fit.log_likelihood(dict(pts[i]))
        assert np.max(np.abs(t - g[f"synth{config}_theory"][i])) < TOL
        assert abs(chi / g[f"synth{config}_chi2"][i] - 1) < TOL
        assert abs(lnl / g[f"synth{config}_lnl"][i] - 1) < TOL
    xi = fit.theory_xi(fit.s, np.linspace(0, 1, 100), dict(pts[0]))
    assert np.max(np.abs(xi - g[f"synth{config}_xi_smu_p0"])) < TOL


def test_survey_appendix_e_pin():
    """SURVEY.md App. E quotes chi2 = 194.23320017005028 for the synthetic metric-grid inputs.  Its data vector perturbs
    the fiducial theory by (1 + 0.01 sin i) with i the bin index WITHIN each multipole; the committed fixtures
    (oracle/make_golden.py) use the index along the stacked vector, which gives 192.5756 at the same point
    (ref_outputs.npz: synth3_chi2[0]).  Same tables, same covariance, same algorithm - only that recipe differs."""
    model, data = cases.synth_options(3)
    fit = vo.OracleFit(model, data)
    p = {"fsigma8": 0.47, "sigma_v": 380, "aperp": 1.02, "apar": 0.97}
    assert abs(fit.log_likelihood(dict(p))[1] - 192.5756) < 1e-4
    t_fid = fit.theory_multipole_vector(fit.s, {"fsigma8": 0.45, "sigma_v": 360, "aperp": 1.0, "apar": 1.0}, fit.poles_s)
    dvec = t_fid * (1 + 0.01 * np.sin(np.tile(np.arange(40), 3)))
    per_pole = vo.OracleFit(model, data, data_input={"s": fit.s, "monopole": dvec[:40], "quadrupole": dvec[40:80],
                                                     "hexadecapole": dvec[80:]})
    lnl, chi2 = per_pole.log_likelihood(dict(p))
    assert abs(chi2 - 194.23320017005028) < 1e-8 and abs(lnl + 97.11660008502514) < 1e-8


def test_covariance_bracket_quirk(boss):
    """ccf_fit.py:226 takes the LAST grid entry as the upper bracket."""
    g = boss.beta_covmat
    m = boss._interp_stack(boss.icov, 0.37)
    lo = np.where(g < 0.37)[0][-1]
    t = (0.37 - g[lo]) / (g[-1] - g[lo])
    assert lo == 12 and abs(t - 0.04537) < 1e-4
    assert np.array_equal(m, (1 - t) * boss.icov[lo] + t * boss.icov[-1])
    assert np.array_equal(boss._interp_stack(boss.icov, 0.1), boss.icov[0])
    assert np.array_equal(boss._interp_stack(boss.icov, 0.7), boss.icov[-1])
    assert np.array_equal(boss._interp_stack(boss.icov, g[5]), boss.icov[5])


def test_model_options_match_reference(boss, gold):
    """SURVEY 8(f3) options on the oracle: linear_bias, empirical_corr and both together, every RSD branch
    (reference outputs: 'opt_boss_*' / 'opt_synth_*' keys of ref_outputs.npz)."""
    g, meta = gold
    base = {"lb": dict(matter_model="linear_bias"), "emp": dict(empirical_corr=True),
            "lb_emp": dict(matter_model="linear_bias", empirical_corr=True)}
    rsd = {"stream": {}, "disp": dict(rsd_model="dispersion"), "kaiser": dict(rsd_model="kaiser")}
    pts = [dict(q, bias=2.1, Av=0.7, M=1.05, Q=0.95) for q in meta["boss_points"][:3]]
    synth = vo.OracleFit(*cases.synth_options(3))
    spts = [dict(q, beta=0.4, bias=1.7, Av=-0.5, M=1.1, Q=0.9) for q in meta["synth_points"][:3]]
    checked = 0
    for b, bkw in base.items():
        for r, rkw in rsd.items():
            kw = dict(bkw, **rkw)
            for fit, points, prefix in ((boss, pts, "opt_boss"), (synth, spts, "opt_synth")):
                key = f"{prefix}_{b}_{r}"
                if key not in g:
                    continue
                t = np.array([fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, **kw) for q in points])
                assert np.max(np.abs(t - g[key])) <= TOL * np.max(np.abs(g[key])), key
                checked += 1
    assert checked >= 17


@pytest.mark.parametrize("case", sorted(cases.SHIPPED_COMBINATIONS))
def test_shipped_file_combinations(case, tmp_path):
    """Every remaining combination of the shipped model / data / covariance files (data/BOSS_DR12_CMASS_data/README.txt):
    the beta-independent covariance with the beta-dependent data vector, the Patchy-mean data vector with its (1e-6 x)
    covariance - chi2 of order 1e7-1e9 -, the measured real-space ccf with the ANISOTROPIC (M+D) covariance, the Patchy mean
    with the isotropic (M+D) covariance.  Reference outputs: tests/golden/ref_outputs_more.npz."""
    g, meta = cases.golden_outputs("more")
    model, data, kw = cases.shipped_combination(case, tmp_path)
    fit = vo.OracleFit(model, data)
    pts = meta["boss_points"]
    th = np.array([fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, **kw) for q in pts])
    assert np.max(np.abs(th - g[f"{case}_theory"])) <= TOL * np.max(np.abs(g[f"{case}_theory"]))
    for form in ("sellentin", "gaussian"):
        ll = [fit.log_likelihood(dict(q), likelihood={"form": form, "nmocks": 1000, "nparams": 4}, **kw) for q in pts]
        lnl, chi = np.array([a for a, b in ll]), np.array([b for a, b in ll])
        assert np.max(np.abs(chi / g[f"{case}_{form}_chi2"] - 1)) < 1e-10, form      # 1e4-1e9: precision stack x 1e3-1e6
        assert np.max(np.abs(lnl / g[f"{case}_{form}_lnl"] - 1)) < 1e-10, form


def test_dispersion_model_where_it_is_ill_conditioned():
    """The rows of tests/golden/ref_outputs_disp.npz (reference outputs where five fixed-point iterations amplify rounding,
    ccf_model.py:658-671, with the reference's own spread under 1-ulp input moves): the oracle performs the reference's
    arithmetic operation for operation, so it follows the reference even there - including the row on which the reference
    returns NaN for every bin."""
    g, meta, options = cases.dispersion_fixture()
    for name, m in meta.items():
        fit = vo.OracleFit(*options[name])
        kw = dict(m["kwargs"])
        for i, row in enumerate(g[f"{name}_rows"]):
            p = dict(zip(m["keys"], (float(x) for x in row)))
            t = fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s, **kw)
            ref = g[f"{name}_theory"][i]
            if not np.all(np.isfinite(ref)):
                assert not np.all(np.isfinite(t)), (name, i)
                continue
            spread = float(g[f"{name}_spread_theory"][i])
            assert np.max(np.abs(t - ref)) <= max(1e-10, 20 * spread) * np.max(np.abs(ref)), (name, i)


def test_oracle_reproduces_the_reference_over_the_prior_box(boss):
    """tests/golden/ref_outputs_box.npz (oracle/make_golden.py --set box): the unmodified reference on 48 Halton points of the
    cobaya prior box (fsigma8, sigma_v, aperp, apar and beta all sampled), BOSS configuration, for the streaming, dispersion,
    kaiser and euclid_special models - theory vector, lnL and chi2.  The oracle must reproduce every one of them (a third of
    the points here, all of them on the GPU: test_gpu_parity.py)."""
    g, meta = cases.golden_outputs("box")
    hp = cases.halton_params(meta["n"], with_beta=True)
    for rsd in ("streaming", "dispersion", "kaiser", "euclid_special"):
        for i in range(0, meta["n"], 3):
            p = cases.point(hp, i)
            t = boss.theory_multipole_vector(boss.s, dict(p), boss.poles_s, rsd_model=rsd)
            lnl, chi = boss.log_likelihood(dict(p), rsd_model=rsd)
            want = g[f"{rsd}_theory"][i]
            assert np.max(np.abs(t - want)) <= TOL * np.max(np.abs(want)), (rsd, i)
            assert abs(chi - g[f"{rsd}_chi2"][i]) <= 1e-11 * chi, (rsd, i)
            assert abs(lnl - g[f"{rsd}_lnl"][i]) <= 1e-11 * max(abs(lnl), 1.0), (rsd, i)
