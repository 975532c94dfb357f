"""DEVELOPMENT CONTAINER ONLY: the oracle against the unmodified reference imported from /root/reference.

Skipped wherever the reference is absent (the GPU box).  The committed golden vectors carry the same
information there (tests/test_oracle_golden.py).
"""

import os
import sys

import numpy as np
import pytest

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.reference


@pytest.fixture(scope="module")
def ref():
    import ref_shim
    if not ref_shim.available():
        pytest.skip("reference not present")
    return ref_shim.load()


def test_reference_runs_its_own_config_and_files(ref):
    import ref_shim
    info = ref_shim.boss_config()
    fit = ref.CCFFit(info["model"], info["data"])
    lnl, chi2 = fit.log_likelihood({"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0})
    assert round(chi2, 2) == 65.01 and round(lnl, 2) == 284.76           # notebook line 491


def test_reference_reproduces_all_five_published_pairs_with_the_legacy_simps(ref):
    """notebooks/victor_usage_demo.ipynb:491-499: all five (chi2, lnL) pairs at printed precision when ``simps`` stands
    for SciPy < 1.11 (even='avg'); with the SciPy >= 1.11 rule the anisotropic pair is 64.40 / 285.05.  Unmodified
    reference, its own config and HDF5 files."""
    import ref_shim
    import victor_oracle as vo
    info = ref_shim.boss_config()
    fit = ref.CCFFit(info["model"], info["data"])
    ofit = vo.OracleFit(*cases.boss_options("config"))
    try:
        ref_shim.set_simpson_rule("avg")
        for name, ((chi_nb, lnl_nb), kw) in cases.NOTEBOOK_PRINTED.items():
            lnl, chi2 = fit.log_likelihood(dict(cases.NOTEBOOK_POINT), **kw)
            assert round(chi2, 2) == chi_nb and round(lnl, 2) == lnl_nb, name
            o_lnl, o_chi2 = ofit.log_likelihood(dict(cases.NOTEBOOK_POINT), simpson_even="avg", **kw)
            assert abs(o_chi2 - chi2) <= 1e-12 * chi2 and abs(o_lnl - lnl) <= 1e-12 * abs(lnl), name
        hp = cases.halton_params(64, with_beta=True)
        for i in (7, 41):
            p = cases.point(hp, i)
            for kw in ({}, {"assume_isotropic": False}, {"rsd_model": "dispersion"}):
                ta = fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s, **kw)
                tb = ofit.theory_multipole_vector(ofit.s, dict(p), ofit.poles_s, simpson_even="avg", **kw)
                assert np.max(np.abs(ta - tb)) < 1e-13
    finally:
        ref_shim.set_simpson_rule("simpson")
    lnl, chi2 = fit.log_likelihood(dict(cases.NOTEBOOK_POINT), assume_isotropic=False)
    assert round(chi2, 2) == 64.40 and round(lnl, 2) == 285.05


def test_oracle_equals_reference_on_fresh_points(ref):
    import victor_oracle as vo
    for opts in (cases.boss_options("config"), cases.boss_options("cobaya"), cases.synth_options(2),
                 cases.synth_options(3)):
        rfit = ref.CCFFit(cases.clone(opts[0]), cases.clone(opts[1]))
        ofit = vo.OracleFit(*opts)
        beta_dep = not ofit.fixed_data
        hp = cases.halton_params(64, with_beta=beta_dep)
        for i in (5, 33, 63):
            p = cases.point(hp, i)
            a = rfit.log_likelihood(dict(p))
            b = ofit.log_likelihood(dict(p))
            assert abs(a[0] - b[0]) <= 1e-12 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-12 * abs(a[1])
            ta = rfit.theory_multipole_vector(rfit.s, dict(p), rfit.poles_s)
            tb = ofit.theory_multipole_vector(ofit.s, dict(p), ofit.poles_s)
            assert np.max(np.abs(ta - tb)) < 1e-13


def test_cobaya_plugin_contract_matches_reference(ref):
    import ref_shim
    Plugin = ref_shim.load_cobaya_plugin()
    model, data = cases.boss_options("cobaya")
    lk = Plugin(model=cases.clone(model), data=cases.clone(data))
    state = {}
    lk.calculate(state, want_derived=True, fsigma8=0.47, beta=0.37, sigma_v=380, epsilon=1.0)
    assert set(state) == {"logp", "derived"} and set(state["derived"]) == {"chi2_ccf_correct"}
    assert lk.get_can_provide_params() == ["fsigma8"]
    g, _ = cases.golden_outputs()
    assert abs(state["derived"]["chi2_ccf_correct"] - g["boss_cobaya_chi2"][0]) < 1e-9


def test_plugin_defaults_list_their_keys_in_the_reference_order(ref):
    """cobaya lists the parameters a likelihood declares in the order of its defaults file, and a chain's columns follow: the
    plug-in's CCFLikelihood.yaml must load to the reference's dictionary WITH the reference's key order (top level and params)."""
    import yaml
    import ref_shim
    with open(os.path.join(ref_shim.REFERENCE_ROOT, "victor", "likelihoods", "CCFLikelihood.yaml")) as fh:
        theirs = yaml.full_load(fh)
    with open(os.path.join(ROOT, "victor", "likelihoods", "CCFLikelihood.yaml")) as fh:
        ours = yaml.full_load(fh)
    assert ours == theirs
    assert list(ours) == list(theirs) and list(ours["params"]) == list(theirs["params"])
    assert list(ours["params"])[:7] == ["fsigma8", "beta", "epsilon", "b", "alpha", "aperp", "apar"]


@pytest.mark.parametrize("with_beta", [False, True])
def test_oracle_rmu_format_equals_reference(ref, tmp_path, with_beta):
    import victor_oracle as vo
    from tests.test_host import _rmu_inputs
    model, _ = _rmu_inputs(tmp_path, with_beta)
    rm = ref.CCFModel(cases.clone(model))
    om = vo.OracleModel(model)
    for ell in ("0", "2", "4"):
        assert np.max(np.abs(rm.real_multipoles[ell] - om.real_multipoles[ell])) < 1e-15
    s = np.linspace(4, 110, 12)
    p = {"fsigma8": 0.5, "beta": 0.33, "sigma_v": 350, "epsilon": 1.02}
    a = rm.theory_multipole_vector(s, dict(p), [0, 2, 4])
    b = om.theory_multipole_vector(s, dict(p), [0, 2, 4])
    assert np.max(np.abs(a - b)) < 1e-13


def test_oracle_anisotropic_dispersion_and_velocity_template_equal_reference(ref, tmp_path):
    import victor_oracle as vo
    from tests.test_host import _aniso_inputs
    model, data = _aniso_inputs(tmp_path, non_uniform_mu=True)
    for mean in ("linear", "template"):
        mdl = cases.clone(model)
        mdl["velocity_pdf"]["mean"]["model"] = mean
        rm = ref.CCFFit(cases.clone(mdl), cases.clone(data))
        om = vo.OracleFit(mdl, data)
        assert np.max(np.abs(rm.sv_rmu - om.sv_rmu)) < 1e-15
        for kw in ({}, {"rsd_model": "dispersion"}, {"rsd_model": "kaiser"}):
            p = {"fsigma8": 0.52, "sigma_v": 330, "aperp": 1.04, "apar": 0.95, "M": 1.05, "Q": 0.9}
            a = rm.log_likelihood(dict(p), **kw)
            b = om.log_likelihood(dict(p), **kw)
            assert abs(a[0] - b[0]) <= 1e-11 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-11 * abs(a[1]), (mean, kw)


def _flatten(keep):
    out = []
    for item in keep:
        if isinstance(item, tuple):
            out.extend(_flatten(item))
        else:
            out.append(np.asarray(item))
    return out


@pytest.mark.parametrize("case", ["boss_config", "boss_cobaya", "synth3"])
def test_device_tables_compile_from_reference_objects(ref, case):
    """INTEGRATION.md route B: victor_amd.engine.build_tables fed with the REFERENCE's own CCFFit object gives the
    same device tables as with this package's CCFFit (no GPU needed to check that)."""
    import victor_amd
    from victor_amd.engine import build_tables
    opts = {"boss_config": cases.boss_options("config"), "boss_cobaya": cases.boss_options("cobaya"),
            "synth3": cases.synth_options(3)}[case]
    rfit = ref.CCFFit(cases.clone(opts[0]), cases.clone(opts[1]))
    mfit = victor_amd.CCFFit(*opts)
    for matter in ("template", "linear_bias"):
        ta, ka = build_tables(rfit, rfit, matter)
        tb, kb = build_tables(mfit, mfit, matter)
        for name in ("n_s", "n_mu", "n_x", "n_ell", "n_ell_r", "n_beta_r", "n_beta_d", "n_beta_c", "matter_model",
                     "vr_beta_dep", "sv_n_mu"):
            assert getattr(ta, name) == getattr(tb, name), name
        assert ta.iaH == tb.iaH and ta.template_sigma8 == tb.template_sigma8
        fa, fb = _flatten(ka), _flatten(kb)
        assert len(fa) == len(fb)
        for x, y in zip(fa, fb):
            assert x.shape == y.shape
            scale = max(1e-300, float(np.max(np.abs(y))))
            assert np.max(np.abs(x - y)) <= 1e-11 * scale


def test_package_reads_the_reference_yaml_and_hdf5_files_unchanged(ref):
    """Drop-in at the data-format boundary: this package's CCFFit built from the reference's own,
    unmodified YAML files and shipped HDF5 data (read by the bundled pure-Python reader, h5py is absent here) compiles
    the same device tables as from the committed .npy fixtures."""
    import yaml
    import ref_shim
    import victor_amd
    from victor_amd.engine import build_tables
    root = ref_shim.REFERENCE_ROOT
    with open(os.path.join(root, "config", "boss_config.yaml")) as fh:
        info = yaml.full_load(fh)
    info["model"]["dir"] = info["data"]["dir"] = root
    with open(os.path.join(root, "config", "boss_cobaya_config.yaml")) as fh:
        cob = yaml.full_load(fh)["likelihood"]["CCFLikelihood"]
    cob["model"]["dir"] = cob["data"]["dir"] = root
    for (model, data), variant in (((info["model"], info["data"]), "config"), ((cob["model"], cob["data"]), "cobaya")):
        assert model["input_model_data_file"].endswith(".hdf5")
        a = victor_amd.CCFFit(model, data)
        b = victor_amd.CCFFit(*cases.boss_options(variant))
        assert a.model == {**b.model} or all(a.model[k] == b.model[k] for k in b.model if k != "dir")
        assert a.fit_options["beta_interpolation"] == b.fit_options["beta_interpolation"]
        for key in ("form", "nmocks"):
            assert a.fit_options["likelihood"][key] == b.fit_options["likelihood"][key]
        ta, ka = build_tables(a, a)
        tb, kb = build_tables(b, b)
        fa, fb = _flatten(ka), _flatten(kb)
        assert len(fa) == len(fb)
        for x, y in zip(fa, fb):
            assert x.shape == y.shape and np.array_equal(x, y)


def test_dispersion_filter_options_against_the_reference(ref):
    """velocity_pdf.dispersion filter / filter_window / filter_order (ccf_model.py:278-283): the oracle's normalised
    sigma_v(r, mu) table and a likelihood through it equal the reference's for non-default options."""
    import victor_oracle as vo
    p = dict(cases.NOTEBOOK_POINT)
    for opt in ({"filter_window": 7, "filter_order": 3}, {"filter": False}):
        model, data = cases.boss_options("config")
        model["velocity_pdf"]["dispersion"].update(opt)
        rfit = ref.CCFFit(cases.clone(model), cases.clone(data))
        ofit = vo.OracleFit(model, data)
        assert np.max(np.abs(ofit.sv_rmu / rfit.sv_rmu - 1)) < 1e-14, opt
        a, b = rfit.log_likelihood(dict(p)), ofit.log_likelihood(dict(p))
        assert abs(a[0] - b[0]) <= 1e-12 * abs(a[0]) and abs(a[1] - b[1]) <= 1e-12 * abs(a[1]), opt
