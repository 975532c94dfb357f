"""linear_bias matter model, empirical velocity correction, real-space ccf measured from data.

Golden vectors come from the reference itself (oracle/make_golden.py, 'opt_*' keys)."""

import numpy as np
import pytest

from tests import cases
from tests.devlib import mapped
from tests.tolerances import assert_same_chi2, chi2_bound
from victor_amd import _native

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
    # linear_bias + empirical_corr on beta-dependent tables: V2 and Ge2 are degree-6 polynomials in beta (vr_emp)
    opt = dict(OPT, lb_emp_stream=dict(matter_model="linear_bias", empirical_corr=True),
               lb_emp_disp=dict(matter_model="linear_bias", empirical_corr=True, rsd_model="dispersion"),
               lb_emp_kaiser=dict(matter_model="linear_bias", empirical_corr=True, rsd_model="kaiser"))
    for tag, kw in opt.items():
        t = np.array([fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, **kw) for q in pts])
        assert close(t, g[f"opt_boss_{tag}"]), tag
    # beta outside the grid (PCHIP extrapolation with the end pieces) against the live oracle
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_oracle as vo
    ora = vo.OracleFit(*cases.boss_options("config"))
    for beta in (0.12, 0.70):
        q = dict(pts[1], beta=beta)
        kw = dict(matter_model="linear_bias", empirical_corr=True)
        want = ora.theory_multipole_vector(ora.s, dict(q), ora.poles_s, **kw)
        got = fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, **kw)
        assert close(got, want), beta


def test_beta_dependent_velocity_tables_run_on_the_fast_kernels(gold):
    """linear_bias on a reconstructed real-space ccf makes the velocity profile beta-dependent (ccf_model.py:358-370).  Its
    dispersion model (Da as beta polynomials, vk_tables.uni_dab) and its empirical_corr branch in both RSD models (V2, Ge1, Ge2
    of degree 6 in beta, ccf_model.py:451-459, vk_tables.uni_empb) are rebuilt per point in LDS like V1, so these combinations
    run on the cells / point-major kernels, not only on the generic one: reference goldens in every mapping, then a batch
    with a different Av per point through the three mappings."""
    import victor_amd
    g, meta = gold
    fit = victor_amd.CCFFit(*cases.boss_options("config"))
    pts = [dict(q, bias=2.1, Av=0.7, M=1.05, Q=0.95) for q in meta["boss_points"][:3]]
    combos = {"lb_disp": dict(matter_model="linear_bias", rsd_model="dispersion"),
              "lb_emp_stream": dict(matter_model="linear_bias", empirical_corr=True),
              "lb_emp_disp": dict(matter_model="linear_bias", empirical_corr=True, rsd_model="dispersion")}
    hp = dict(cases.halton_params(1000, with_beta=True), bias=np.linspace(1.6, 2.4, 1000), Av=np.linspace(-1.0, 1.0, 1000))
    for tag, kw in combos.items():
        model = fit._merged(kw)
        eng = fit._get_engine(fit._engine_key(model))
        batch = np.vstack([fit._fit_rows(dict(p), model) for p in pts] + [fit._fit_rows(hp, model)])
        res = {}
        for mapping in ("point", "cells", "generic"):
            with mapped(fit, mapping) as f:
                res[mapping] = f.theory_vector_batch(batch, **kw)
                assert eng.last_kernel().endswith({"point": "fast_kernel", "cells": "cells_kernel", "generic": "vk_theory_kernel"}[mapping]), (tag, mapping)
            assert close(res[mapping][:3], g[f"opt_boss_{tag}"]), (tag, mapping)
        fit.theory_vector_batch(batch, **kw)                               # the default choice at this size
        assert eng.last_kernel() == "vk_theory_cells_kernel", (tag, eng.last_kernel())
        for mapping in ("cells", "generic"):
            row_err = np.max(np.abs(res[mapping] - res["point"]), axis=1) / np.max(np.abs(res["point"]))
            assert np.quantile(row_err, 0.995) < 1e-10 and row_err.max() < 1e-6, (tag, mapping, row_err.max())
        # and the likelihood through the fused tail of the fast kernels against the generic path
        a = fit.log_likelihood_batch(batch[:200], **kw)
        _native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
        try:
            b = fit.log_likelihood_batch(batch[:200], **kw)
        finally:
            _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
        assert np.max(np.abs(a[1] / b[1] - 1)) < 1e-8, tag


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
    # the streaming branch runs on the fast kernels with a second interval look-up at the fiducial coordinates
    import os
    hp = cases.halton_params(1500, with_beta=True)
    rows = np.vstack([batch_rows, fit._fit_rows(hp, fit.model)])
    res = {}
    for aniso in (False, True):
        for mapping in ("point", "cells", "generic"):
            with mapped(fit, mapping) as f:
                res[mapping] = f.theory_vector_batch(rows, assume_isotropic=not aniso)
                assert f._get_engine().last_kernel().endswith(
                    {"point": "fast_kernel", "cells": "cells_kernel", "generic": "vk_theory_kernel"}[mapping])
            want = g["opt_fromdata_aniso_theory"] if aniso else g["opt_fromdata_theory"][:3]
            assert close(res[mapping][:3], want), (aniso, mapping)
        for mapping in ("cells", "generic"):
            assert np.max(np.abs(res[mapping] - res["point"])) < 1e-10 * np.max(np.abs(res["point"])), (aniso, mapping)


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
    """3-key sigma_v(r, mu) template (bicubic, box-clamped; uniform and non-uniform mu knots) vs the oracle: streaming model on
    the fast kernels, dispersion model on the cells kernel (patches in LDS, SVA instantiations)."""
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
    for i, kw, kernel in ((3, {}, "vk_theory_fast_kernel"), (11, {"rsd_model": "dispersion"}, "vk_theory_cells_kernel"),
                          (29, {"assume_isotropic": True}, "vk_theory_fast_kernel"), (30, {"rsd_model": "kaiser"}, "vk_theory_cells_kernel")):
        p = cases.point(hp, i)
        got = fit.log_likelihood(dict(p), **kw)
        want = ora.log_likelihood(dict(p), **kw)
        assert abs(got[1] / want[1] - 1) < RTOL and abs(got[0] / want[0] - 1) < RTOL, (i, kw)
        # the patches ride in LDS (uniform r grid, lattice form); kaiser does not read sigma_v at all
        assert fit._get_engine().last_kernel() == kernel, (kw, fit._get_engine().last_kernel())
    # a batch through every mapping that can take it: oracle on a few rows, mapping against mapping on all of them
    # (rows with |mu_r| beyond the template's box and r beyond its r range are clamped as FITPACK's bispeu does)
    hb = cases.halton_params(600)
    rows = fit._fit_rows(hb, fit.model)
    res = {}
    for mapping in ("point", "cells", "generic"):
        with mapped(fit, mapping) as f:
            res[mapping] = f.theory_vector_batch(rows)
            assert f._get_engine().last_kernel().endswith(
                {"point": "fast_kernel", "cells": "cells_kernel", "generic": "vk_theory_kernel"}[mapping]), mapping
    for i in (0, 17, 599):
        want = ora.theory_multipole_vector(ora.s, cases.point(hb, i), ora.poles_s)
        for mapping in res:
            assert close(res[mapping][i], want), (mapping, i)
    for mapping in ("cells", "generic"):
        assert np.max(np.abs(res[mapping] - res["point"])) < 1e-10 * np.max(np.abs(res["point"])), mapping
    fit.theory_vector_batch(rows)
    assert fit._get_engine().last_kernel() == "vk_theory_cells_kernel"        # the default at this size
    # the dispersion model on the same template: cells kernel at every batch size against the generic kernel and the oracle
    got = fit.log_likelihood_batch(rows, rsd_model="dispersion")
    assert fit._get_engine().last_kernel() == "vk_theory_cells_kernel"
    few = fit.log_likelihood_batch(rows[:3], rsd_model="dispersion")
    assert fit._get_engine().last_kernel() == "vk_theory_cells_kernel"
    _native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
    try:
        ref = fit.log_likelihood_batch(rows, rsd_model="dispersion")
        assert fit._get_engine().last_kernel() == "vk_theory_kernel"
    finally:
        _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
    bound = chi2_bound(fit, rows, ulps=4096, rsd_model="dispersion")       # another arithmetic, and an iteration that amplifies it
    assert_same_chi2(got[1], ref[1], bound, what="dispersion x sigma_v(r, mu): cells vs generic")
    assert_same_chi2(few[1], got[1][:3], bound[:3], what="dispersion x sigma_v(r, mu): 3 rows vs 600")
    for i in (0, 17, 599):
        want = ora.log_likelihood(cases.point(hb, i), rsd_model="dispersion")
        assert abs(got[1][i] / want[1] - 1) < RTOL, i
    with mapped(fit, "lanes") as f:                      # (development build) the lanes kernel cannot take it
        f.theory_vector_batch(rows[:64])
        assert f._get_engine().last_kernel() != "vk_theory_lanes_kernel"


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


def test_example_void_model_hdf5_non_uniform_grids():
    """The reference's example model file (HDF5, distances in units of the void radius, NON-uniform r grid): the
    generic kernel with knot search, for the parameter sets of notebooks/model_options_demo.ipynb."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_amd
    import victor_oracle as vo
    from victor_amd import h5lite
    model = {"dir": cases.GOLDEN, "input_model_data_file": "h5/void_model_example.h5", "rsd_model": "streaming",
             "z_eff": 0.50, "cosmology": {"Omega_m": 0.31},
             "realspace_ccf": {"reconstruction": False, "format": "multipoles", "ccf_keys": ["r", "monopole"]},
             "matter_ccf": {"model": "template", "integrated": False, "template_keys": ["rdelta", "delta"],
                            "template_sigma8": 0.628, "bias": 1.9},
             "velocity_pdf": {"mean": {"model": "linear"},
                              "dispersion": {"model": "template", "template_keys": ["rsv", "sigmav"]}}}
    m = victor_amd.CCFModel(model)
    o = vo.OracleModel(model, h5lite.read_all(os.path.join(cases.GOLDEN, "h5", "void_model_example.h5")))
    assert np.ptp(np.diff(m.r)) > 1e-3                      # the grid really is non-uniform
    s = np.linspace(0.01, 3, 100)                           # as in the notebook
    for p, kw in (({"fsigma8": 0.47, "sigma_v": 7, "epsilon": 1.0}, {}),
                  ({"fsigma8": 0.47, "sigma_v": 7, "epsilon": 1.0}, {"rsd_model": "dispersion"}),
                  ({"fsigma8": 0.47, "sigma_v": 7, "epsilon": 1.0, "M": 1.0, "Q": 1.0},
                   {"rsd_model": "kaiser", "kaiser_approximation": True, "kaiser_coord_shift": False}),
                  ({"beta": 0.4, "epsilon": 1.0, "fsigma8": 0.47}, {"rsd_model": "kaiser", "matter_model": "linear_bias"}),
                  ({"fsigma8": 0.47, "sigma_v": 7, "epsilon": 1.0, "Av": 1},
                   {"rsd_model": "dispersion", "empirical_corr": True})):
        got = m.theory_multipoles(s, dict(p), poles=[0, 2], **kw)
        want, _ = o.theory_multipoles(s, dict(p), poles=[0, 2], **kw)
        for key in ("0", "2"):
            assert np.max(np.abs(got[key] - want[key])) < RTOL * np.max(np.abs(want[key])), (kw, key)
    # non-uniform grids take the union-grid form of the fast kernels, dispersion + empirical_corr included
    assert m._get_engine().last_kernel() == "vk_theory_fast_kernel"


def test_special_amplitudes_in_every_fast_mapping(tmp_path, gold):
    """linear_bias on fixed tables and the velocity-template mean model change only the per-point amplitude, so they run
    on the fast kernels; all three mappings must agree with each other and with the reference / oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_amd
    import victor_oracle as vo
    from tests.test_host import _aniso_inputs
    g, meta = gold
    hp = cases.halton_params(8192 + 11)
    # linear_bias, fixed real-space input: golden rows first, then a large batch through each mapping
    fit = victor_amd.CCFFit(*cases.synth_options(3))
    pts = [dict(q, beta=0.4, bias=1.7) for q in meta["synth_points"][:3]]
    batch = dict({k: np.concatenate([[p[k] for p in pts], v]) for k, v in hp.items()}, beta=0.4, bias=1.7)
    res = {}
    for mapping in ("point", "cells", "lanes"):
        with mapped(fit, mapping) as f:
            th = f.theory_vector_batch(batch, matter_model="linear_bias")
        assert close(th[:3], g["opt_synth_lb_stream"]), mapping
        res[mapping] = th
    assert close(res["cells"], res["point"]) and close(res["lanes"], res["point"])
    # velocity template mean model
    model, data = _aniso_inputs(tmp_path)
    model["velocity_pdf"]["mean"]["model"] = "template"
    model["velocity_pdf"]["dispersion"] = {"model": "template", "template_keys": ["rsv", "sigmav"]}
    fit = victor_amd.CCFFit(model, data)
    ora = vo.OracleFit(model, data)
    res = {}
    for mapping in ("point", "cells", "lanes"):
        with mapped(fit, mapping) as f:
            res[mapping] = f.log_likelihood_batch(hp)
            assert f._get_engine("velocity_template").last_kernel().endswith(
                {"point": "fast_kernel", "cells": "cells_kernel", "lanes": "lanes_kernel"}[mapping])
    for i in (0, 4000, 8202):
        want = ora.log_likelihood(cases.point(hp, i))
        for mapping in res:
            assert abs(res[mapping][1][i] / want[1] - 1) < RTOL, (mapping, i)
    assert_same_chi2(res["lanes"][1], res["point"][1], chi2_bound(fit, hp), what="velocity template, lanes vs point")


def test_offset_commensurate_grids_in_every_fast_mapping(tmp_path):
    """r and sigma_v grids whose first knots are not multiples of the common spacing: the unified grid then starts
    below zero (off != 0) and the leading V interval [0.01, r_0] spans a fraction of a refined interval."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_amd
    import victor_oracle as vo
    src = np.load(os.path.join(cases.GOLDEN, "synth", "model.npy"), allow_pickle=True).item()
    r = 1.3 + 3.0 * np.arange(40)
    rsv = 2.8 + 4.5 * np.arange(26)
    tab = dict(src, r=r, rsv=rsv, sigmav=np.interp(rsv, src["rsv"], src["sigmav"]))
    for key in ("monopole", "quadrupole", "hexadecapole"):
        tab[key] = np.interp(r, src["r"], src[key])
    np.save(tmp_path / "model_offset.npy", tab, allow_pickle=True)
    model, data = cases.synth_options(3)
    model = dict(model, dir=str(tmp_path), input_model_data_file="model_offset.npy")
    fit = victor_amd.CCFFit(model, data)
    from victor_amd.engine import build_tables
    tabs, _keep = build_tables(fit, fit)
    assert tabs.uni_n > 0 and tabs.uni_u0 < 0 and abs(tabs.uni_u0 + 0.2) < 1e-12     # 1.3 - 1 * 1.5
    ora = vo.OracleFit(model, data)
    hp = cases.halton_params(4096 + 5)
    res = {}
    for mapping in ("point", "cells", "lanes"):
        with mapped(fit, mapping) as f:
            res[mapping] = f.log_likelihood_batch(hp)
            assert f._get_engine().last_kernel().endswith(
                {"point": "fast_kernel", "cells": "cells_kernel", "lanes": "lanes_kernel"}[mapping])
    for i in (0, 1, 2047, 4100):
        want = ora.log_likelihood(cases.point(hp, i))
        for mapping in res:
            assert abs(res[mapping][1][i] / want[1] - 1) < RTOL, (mapping, i)
    bound = chi2_bound(fit, hp)
    assert_same_chi2(res["lanes"][1], res["point"][1], bound, what="offset lattice, lanes vs point")
    assert_same_chi2(res["cells"][1], res["point"][1], bound, what="offset lattice, cells vs point")


def test_non_uniform_grids_in_every_fast_mapping(tmp_path):
    """Jittered r and sigma_v grids (neither uniform nor commensurate): the unified tables take their union-grid form
    (look-up table + one knot comparison) and every fast mapping must agree with the oracle and the generic kernel."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_amd
    import victor_oracle as vo
    from victor_amd.engine import build_tables
    src = np.load(os.path.join(cases.GOLDEN, "synth", "model.npy"), allow_pickle=True).item()
    rng = np.random.default_rng(11)
    r = src["r"] + rng.uniform(-0.9, 0.9, len(src["r"]))
    rsv = src["rsv"] + rng.uniform(-2.0, 2.0, len(src["rsv"]))
    tab = dict(src, r=r, rsv=rsv, sigmav=np.interp(rsv, src["rsv"], src["sigmav"]))
    for key in ("monopole", "quadrupole", "hexadecapole"):
        tab[key] = np.interp(r, src["r"], src[key])
    np.save(tmp_path / "model_jitter.npy", tab, allow_pickle=True)
    model, data = cases.synth_options(3)
    model = dict(model, dir=str(tmp_path), input_model_data_file="model_jitter.npy")
    fit = victor_amd.CCFFit(model, data)
    tabs, _keep = build_tables(fit, fit)
    assert tabs.uni_n == len(r) + len(rsv) and tabs.uni_lut_n >= 64
    # a C-ABI client that hands over a look-up table that does not match its knots (the kernels index LDS with it) is
    # turned away by vk_create with VK_E_ARG-style diagnostics instead of reading out of bounds
    import ctypes as C
    lib = _native.load()
    err = C.create_string_buffer(256)
    good_inv_g, good_n = tabs.uni_lut_inv_g, tabs.uni_lut_n
    lut = np.ctypeslib.as_array(tabs.uni_lut, shape=(good_n,))
    for what in ("range", "entry", "cells"):
        saved = lut.copy()
        if what == "range":
            tabs.uni_lut_inv_g = good_inv_g * 1.5             # the clamp range now maps beyond the table
        elif what == "entry":
            lut[good_n // 2] += 3                            # an entry that is not the interval of its cell's left edge
        else:
            tabs.uni_lut_n = good_n // 2                      # a table that stops half way
        ctx = lib.vk_create(C.byref(tabs), 0, err, len(err))
        assert not ctx and b"look-up" in err.value, (what, err.value)
        tabs.uni_lut_inv_g, tabs.uni_lut_n = good_inv_g, good_n
        lut[:] = saved
    ctx = lib.vk_create(C.byref(tabs), 0, err, len(err))
    assert ctx, err.value
    lib.vk_destroy(ctx)
    ora = vo.OracleFit(model, data)
    hp = cases.halton_params(4096 + 5)
    res = {}
    for mapping in ("point", "cells", "lanes", "generic"):
        with mapped(fit, mapping) as f:
            res[mapping] = f.log_likelihood_batch(hp)
            assert f._get_engine().last_kernel().endswith(
                {"point": "fast_kernel", "cells": "cells_kernel", "lanes": "lanes_kernel",
                 "generic": "vk_theory_kernel"}[mapping])
    for i in (0, 1, 2047, 4100):
        want = ora.log_likelihood(cases.point(hp, i))
        for mapping in res:
            assert abs(res[mapping][1][i] / want[1] - 1) < RTOL, (mapping, i)
    for mapping in ("cells", "lanes", "generic"):
        bound = chi2_bound(fit, hp, ulps=1024 if mapping == "generic" else 64)      # the generic kernel: another arithmetic
        assert_same_chi2(res[mapping][1], res["point"][1], bound, what=f"union grid, {mapping} vs point")


def test_fine_grids_need_more_than_64k_of_lds(tmp_path):
    """300 r bins and 150 sigma_v bins: the tables of every theory kernel exceed the 64 KiB default of dynamic LDS (the
    launcher opts in to the 160 KiB of gfx950); all mappings against the oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_amd
    import victor_oracle as vo
    from victor_amd.engine import build_tables
    src = np.load(os.path.join(cases.GOLDEN, "synth", "model.npy"), allow_pickle=True).item()
    r = 0.5 + 0.4 * np.arange(300)
    rsv = 1.0 + 0.8 * np.arange(150)
    tab = dict(src, r=r, rsv=rsv, sigmav=np.interp(rsv, src["rsv"], src["sigmav"]))
    for key in ("monopole", "quadrupole", "hexadecapole"):
        tab[key] = np.interp(r, src["r"], src[key])
    np.save(tmp_path / "model_fine.npy", tab, allow_pickle=True)
    model, data = cases.synth_options(3)
    model = dict(model, dir=str(tmp_path), input_model_data_file="model_fine.npy")
    fit = victor_amd.CCFFit(model, data)
    tabs, _keep = build_tables(fit, fit)
    assert tabs.uni_n == 450 and tabs.uni_lut_n > 0          # 450 records x 176 B = 79 KB
    ora = vo.OracleFit(model, data)
    hp = cases.halton_params(1024 + 3)
    res = {}
    for mapping in ("point", "cells", "lanes", "generic"):
        with mapped(fit, mapping) as f:
            res[mapping] = f.log_likelihood_batch(hp)
    for i in (0, 1026):
        want = ora.log_likelihood(cases.point(hp, i))
        for mapping in res:
            assert abs(res[mapping][1][i] / want[1] - 1) < RTOL, (mapping, i)
    for mapping in ("cells", "lanes", "generic"):
        bound = chi2_bound(fit, hp, ulps=1024 if mapping == "generic" else 64)      # the generic kernel: another arithmetic
        assert_same_chi2(res[mapping][1], res["point"][1], bound, what=f"union grid, {mapping} vs point")


def test_boss_linear_bias_runs_on_the_fast_kernels(gold):
    """linear_bias on the reconstructed BOSS tables: V1 = r Delta follows xi^r_0(beta), rebuilt per point in the fast
    kernels (vk_tables.uni_vb); golden rows from the reference, then a batch through every mapping."""
    import os
    import victor_amd
    g, meta = gold
    fit = victor_amd.CCFFit(*cases.boss_options("config"))
    pts = [dict(q, bias=2.1, Av=0.7, M=1.05, Q=0.95) for q in meta["boss_points"][:3]]
    hp = cases.halton_params(2048 + 7, with_beta=True)
    lb_model = fit._merged({"matter_model": "linear_bias"})
    batch = np.vstack([fit._fit_rows(dict(p), lb_model) for p in pts] + [fit._fit_rows(dict(hp, bias=2.1), lb_model)])
    res = {}
    for mapping in ("point", "cells", "generic"):
        with mapped(fit, mapping) as f:
            res[mapping] = f.theory_vector_batch(batch, matter_model="linear_bias")
            assert f._get_engine("linear_bias").last_kernel().endswith(
                {"point": "fast_kernel", "cells": "cells_kernel", "generic": "vk_theory_kernel"}[mapping])
        assert close(res[mapping][:3], g["opt_boss_lb_stream"]), mapping
    for mapping in ("cells", "generic"):
        assert np.max(np.abs(res[mapping] - res["point"])) < 1e-10 * np.max(np.abs(res["point"])), mapping


def test_empirical_corr_runs_on_the_fast_kernels(gold):
    """empirical_corr on fixed velocity tables: the kernels that own a point per workgroup rewrite V = V1 + Av V2 for it
    (vk_tables.uni_v2); golden rows from the reference first, then a batch with a different Av per point."""
    import os
    import victor_amd
    g, meta = gold
    for name, opts, beta, key in (("synth", cases.synth_options(3), False, "opt_synth_emp_stream"),
                                  ("boss", cases.boss_options("config"), True, "opt_boss_emp_stream")):
        fit = victor_amd.CCFFit(*opts)
        if name == "synth":
            pts = [dict(q, beta=0.4, bias=1.7, Av=-0.5, M=1.1, Q=0.9) for q in meta["synth_points"][:3]]
        else:
            pts = [dict(q, bias=2.1, Av=0.7, M=1.05, Q=0.95) for q in meta["boss_points"][:3]]
        model = fit._merged({"empirical_corr": True})
        hp = cases.halton_params(1500, with_beta=beta)
        hp = dict(hp, Av=np.linspace(-1.0, 1.0, 1500))
        batch = np.vstack([fit._fit_rows(dict(p), model) for p in pts] + [fit._fit_rows(hp, model)])
        res = {}
        for mapping in ("point", "cells", "generic"):
            with mapped(fit, mapping) as f:
                res[mapping] = f.theory_vector_batch(batch, empirical_corr=True)
                assert f._get_engine().last_kernel().endswith(
                    {"point": "fast_kernel", "cells": "cells_kernel", "generic": "vk_theory_kernel"}[mapping])
            assert close(res[mapping][:3], g[key]), (name, mapping)
        # Mapping against mapping: rounding-level agreement, except for the isolated rows where a velocity node puts the
        # fixed-point iteration of ccf_model.py:660-664 next to r = 0 (mu = 1): there the five iterations amplify a 1-ulp
        # difference in any input - in the reference as much as here (test_dispersion_model_where_it_is_ill_conditioned pins
        # that against the reference's own 1-ulp spread) - so the bound is on the bulk of the rows plus the contract's 1e-6
        # on the worst one of this batch.
        for mapping in ("cells", "generic"):
            row_err = np.max(np.abs(res[mapping] - res["point"]), axis=1) / np.max(np.abs(res["point"]))
            assert np.quantile(row_err, 0.995) < 1e-10 and row_err.max() < 1e-6, (name, mapping, row_err.max())


def test_dispersion_model_where_it_is_ill_conditioned():
    """Parity against the REFERENCE on the rows where the dispersion model is ill-conditioned (a velocity node that lets the
    fixed-point iteration of ccf_model.py:660-664 collapse towards r = 0 at mu = 1).  tests/golden/ref_outputs_disp.npz
    holds the reference's outputs on the worst rows of two 131072-point scans per configuration (tools/gpu_find_disp_rows.py)
    AND its own spread when one input moves by one ulp (oracle/make_golden.py --set disp).  Every GPU mapping must
      * meet the contract (1e-6 of max|xi_l|, 1e-6 relative on chi2) wherever the reference is itself stable at that level,
      * deviate by more than 1e-9 only where the reference does too, and by a comparable amount (the amplification is chaotic:
        a factor 50 on the recorded 1-ulp spread, measured worst case 6),
      * on the row where the reference returns NaN for every bin (an integrand point whose Jacobian cancels to exactly 0,
        smeared over all bins by its global bicubic fit - and finite again after a 1-ulp move), agree between its mappings or
        report the failure pair."""
    import victor_amd
    g, meta, options = cases.dispersion_fixture()
    worst = 0.0
    for name, m in meta.items():
        fit = victor_amd.CCFFit(*options[name])
        kw = dict(m["kwargs"])
        model = fit._merged(kw)
        pd = {k: g[f"{name}_rows"][:, j] for j, k in enumerate(m["keys"])}
        rows = fit._fit_rows(pd, model)
        ref_t, ref_c = g[f"{name}_theory"], g[f"{name}_chi2"]
        sp_t, sp_c = g[f"{name}_spread_theory"], g[f"{name}_spread_chi2"]
        res = {}
        for mapping in ("point", "cells", "generic"):
            with mapped(fit, mapping) as f:
                th = f.theory_vector_batch(rows, **kw)
                lnl, chi = f.log_likelihood_batch(rows, **kw)
            res[mapping] = th
            for i in range(len(rows)):
                if not np.all(np.isfinite(ref_t[i])):
                    assert np.isposinf(ref_c[i])
                    assert (np.isneginf(lnl[i]) and np.isposinf(chi[i])) or np.all(np.isfinite(th[i])), (name, mapping, i)
                    continue
                scale = np.max(np.abs(ref_t[i]))
                dev = np.max(np.abs(th[i] - ref_t[i])) / scale
                dchi = abs(chi[i] / ref_c[i] - 1)
                worst = max(worst, dev / max(sp_t[i], 1e-16))
                assert dev <= max(1e-9, 50 * sp_t[i]), (name, mapping, i, dev, sp_t[i])
                assert dchi <= max(1e-9, 50 * sp_c[i], 50 * sp_t[i]), (name, mapping, i, dchi, sp_c[i])
                if sp_t[i] <= 2e-8:                      # the reference is stable at the contract's level here
                    assert dev <= 1e-6 and dchi <= 1e-6, (name, mapping, i, dev, dchi)
        for i in range(len(rows)):
            if not np.all(np.isfinite(ref_t[i])) and all(np.all(np.isfinite(res[mp][i])) for mp in res):
                scale = np.max(np.abs(res["generic"][i]))
                assert max(np.max(np.abs(res[mp][i] - res["generic"][i])) for mp in ("point", "cells")) <= 1e-6 * scale, (name, i)


def test_dispersion_model_runs_on_the_fast_kernels(gold):
    """rsd_model='dispersion' on fixed velocity tables: the fixed-point coordinate, Jacobian and zero-mean pdf evaluated on
    the unified records (vk_tables.uni_da); reference goldens first, then a batch through every mapping."""
    import os
    import victor_amd
    g, meta = gold
    emp_pts = [dict(q, bias=2.1, Av=0.7, M=1.05, Q=0.95) for q in meta["boss_points"][:3]]
    for name, opts, beta, key, pts, kw in (
            ("synth", cases.synth_options(3), False, None, [], {}),
            ("boss", cases.boss_options("config"), True, "boss_dispersion_theory", meta["boss_points"][:3], {}),
            ("boss+empirical_corr", cases.boss_options("config"), True, "opt_boss_emp_disp", emp_pts,
             {"empirical_corr": True})):
        kw = dict(kw, rsd_model="dispersion")
        fit = victor_amd.CCFFit(*opts)
        model = fit._merged(kw)
        hp = cases.halton_params(1200, with_beta=beta)
        if "empirical_corr" in kw:
            hp = dict(hp, Av=np.linspace(-1.0, 1.0, 1200))
        npts = len(pts)
        parts = [fit._fit_rows(dict(p), model) for p in pts] + [fit._fit_rows(hp, model)]
        rows = np.vstack(parts)
        res = {}
        for mapping in ("point", "cells", "generic"):
            with mapped(fit, mapping) as f:
                res[mapping] = f.theory_vector_batch(rows, **kw)
                assert f._get_engine().last_kernel().endswith(
                    {"point": "fast_kernel", "cells": "cells_kernel", "generic": "vk_theory_kernel"}[mapping])
            if key:
                assert close(res[mapping][:npts], g[key]), (name, mapping)
        # Mapping against mapping: rounding-level agreement, except for the isolated rows where a velocity node puts the
        # fixed-point iteration of ccf_model.py:660-664 next to r = 0 (mu = 1): there the five iterations amplify a 1-ulp
        # difference in any input - in the reference as much as here (test_dispersion_model_where_it_is_ill_conditioned pins
        # that against the reference's own 1-ulp spread) - so the bound is on the bulk of the rows plus the contract's 1e-6
        # on the worst one of this batch.
        for mapping in ("cells", "generic"):
            row_err = np.max(np.abs(res[mapping] - res["point"]), axis=1) / np.max(np.abs(res["point"]))
            assert np.quantile(row_err, 0.995) < 1e-10 and row_err.max() < 1e-6, (name, mapping, row_err.max())


@pytest.mark.parametrize("case", sorted(cases.SHIPPED_COMBINATIONS))
def test_shipped_file_combinations_on_gpu(case, tmp_path):
    """The remaining combinations of the shipped model / data / covariance files against the reference's own outputs
    (tests/golden/ref_outputs_more.npz; data/BOSS_DR12_CMASS_data/README.txt): fixed covariance with a beta-dependent data
    vector (one precision matrix, no log-det term), the Patchy-mean data vector with its covariance (chi2 up to 1e9),
    the measured real-space ccf with the anisotropic (M+D) covariance and the anisotropic sum, the Patchy mean with the
    isotropic (M+D) covariance.  Single-point API, batch API and every chi-square kernel."""
    import victor_amd
    g, meta = cases.golden_outputs("more")
    model, data, kw = cases.shipped_combination(case, tmp_path)
    fit = victor_amd.CCFFit(model, data)
    pts = meta["boss_points"]
    rows = np.concatenate([fit._fit_rows(dict(p), fit._merged(kw)) for p in pts])
    assert close(fit.theory_vector_batch(rows, **kw), g[f"{case}_theory"])
    for form in ("sellentin", "gaussian"):
        like = {"form": form, "nmocks": 1000, "nparams": 4}
        want_l, want_c = g[f"{case}_{form}_lnl"], g[f"{case}_{form}_chi2"]
        lnl, chi = fit.log_likelihood_batch(rows, likelihood=like, **kw)                   # fused chi-square
        assert np.max(np.abs(chi / want_c - 1)) < RTOL and np.max(np.abs(lnl / want_l - 1)) < RTOL, form
        one = [fit.log_likelihood(dict(p), likelihood=like, **kw) for p in pts[:3]]        # the reference's call
        assert np.max(np.abs(np.array([a for a, b in one]) / want_l[:3] - 1)) < RTOL, form
        big = np.tile(rows, (700, 1))                                                      # 5600 rows: separate K2 launch
        for knob in ({}, {"VICTOR_HIP_LIKE_UNTILED": "1"}, {"VICTOR_HIP_NO_FUSE": "1"}):
            for k, v in knob.items():
                _native.set_knob(k, v)
            try:
                lb, cb = fit.log_likelihood_batch(big, likelihood=like, **kw)
            finally:
                for k in knob:
                    _native.set_knob(k, None)
            assert np.max(np.abs(cb.reshape(700, -1) / want_c - 1)) < RTOL, (form, knob)
            assert np.max(np.abs(lb.reshape(700, -1) / want_l - 1)) < RTOL, (form, knob)


def test_kaiser_and_euclid_special_run_on_the_cells_kernel(tmp_path):
    """The models without a velocity integral (ccf_model.py:692-784) take the cells kernel on the unified-grid records (one lane
    per (s, mu) cell, chi-square fused) for every grid form and option the tables allow; FORCE_GENERIC keeps the generic kernel
    (library sqrt / division, knot search) as the yardstick.  Every option combination against the live oracle on a few rows
    and against the generic kernel on the whole batch."""
    import os
    import sys
    import victor_amd
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_oracle as vo

    def run(fit, ora, hp, tag, probe, **kw):
        got = fit.log_likelihood_batch(hp, **kw)
        key = fit._engine_key(fit._merged(kw))
        assert fit._get_engine(key).last_kernel() == "vk_theory_cells_kernel" and fit._get_engine(key).last_fused(), (tag, kw)
        _native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
        try:
            ref = fit.log_likelihood_batch(hp, **kw)
            assert fit._get_engine(key).last_kernel() == "vk_theory_kernel"
        finally:
            _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
        assert_same_chi2(got[1], ref[1], chi2_bound(fit, hp, ulps=1024, **kw), what=f"kaiser cells vs generic: {tag} {kw}")
        for i in probe:
            want = ora.log_likelihood(cases.point(hp, i), **kw)
            assert abs(got[1][i] / want[1] - 1) < RTOL and abs(got[0][i] - want[0]) < RTOL * (abs(want[0]) + want[1]), (tag, kw, i)
        # one point per call and a handful of points (cell ranges, partial sums) agree with the rows of the batch
        one = fit.log_likelihood(cases.point(hp, probe[0]), **kw)
        assert abs(one[1] / got[1][probe[0]] - 1) < 1e-11, (tag, kw)
        few = fit.log_likelihood_batch({k: (v[:5] if np.ndim(v) else v) for k, v in hp.items()}, **kw)
        assert np.max(np.abs(few[1] / got[1][:5] - 1)) < 1e-11, (tag, kw)

    # BOSS: beta-dependent tables, isotropic xi^r, sellentin form, blended covariance
    opts = cases.boss_options("config")
    fit, ora = victor_amd.CCFFit(*opts), vo.OracleFit(*opts)
    hp = dict(cases.halton_params(700, with_beta=True), M=1.07, Q=0.93, bias=2.0, Av=0.6)
    for kw in ({"rsd_model": "kaiser"}, {"rsd_model": "euclid_special"}, {"rsd_model": "kaiser", "kaiser_coord_shift": False},
               {"rsd_model": "kaiser", "kaiser_approximation": True}, {"rsd_model": "kaiser", "niter": 2},
               {"rsd_model": "kaiser", "matter_model": "linear_bias"}, {"rsd_model": "euclid_special", "empirical_corr": True},
               {"rsd_model": "kaiser", "matter_model": "linear_bias", "empirical_corr": True}):
        run(fit, ora, hp, "boss", (0, 350, 699), **kw)
    # config 3: fixed tables, three real-space multipoles, l = 0, 2, 4 (lattice grid)
    opts = cases.synth_options(3)
    fit, ora = victor_amd.CCFFit(*opts), vo.OracleFit(*opts)
    hp = dict(cases.halton_params(9000), M=0.95, Q=1.1)
    for kw in ({"rsd_model": "kaiser"}, {"rsd_model": "euclid_special"}, {"rsd_model": "kaiser", "assume_isotropic": True}):
        run(fit, ora, hp, "config3", (0, 4500, 8999), **kw)
    # the union-grid form (r grid and sigma_v grid not commensurate)
    src = np.load(os.path.join(cases.GOLDEN, "synth", "model.npy"), allow_pickle=True).item()
    tab = dict(src)
    tab["rsv"] = np.asarray(src["rsv"]) * 1.037 + 0.21
    np.save(tmp_path / "model_union.npy", tab, allow_pickle=True)
    model, data = cases.synth_options(3)
    model = dict(model, dir=str(tmp_path), input_model_data_file="model_union.npy")
    fit, ora = victor_amd.CCFFit(model, data), vo.OracleFit(model, data)
    from victor_amd.engine import build_tables
    tabs, _keep = build_tables(fit, fit)
    assert tabs.uni_n > 0 and tabs.uni_lut_n > 0
    hp = dict(cases.halton_params(300), M=1.0, Q=1.0)
    run(fit, ora, hp, "union grid", (0, 299), rsd_model="kaiser")


def test_theory_xi_with_every_table_option_on_the_cells_kernel(tmp_path):
    """``CCFModel.theory_xi`` through the cells kernel's store-every-cell form for the table options of SURVEY 8(f3): a measured
    real-space ccf (``from_data``: the second look-up at the fiducial coordinates), ``linear_bias`` on reconstruction-beta tables
    (velocity tables rebuilt per point), ``empirical_corr``, the anisotropic sigma_v(r, mu) template (patches in LDS, streaming
    and dispersion) - every RSD model each, against the generic kernel on the whole array and the oracle on single cells."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_amd
    import victor_oracle as vo
    from tests.test_host import _aniso_inputs
    mu = np.linspace(0, 1, 100)
    m_fd, d_fd = cases.boss_options("config")
    m_fd["input_model_data_file"] = "boss/measured_model.npy"
    m_fd["realspace_ccf"]["from_data"] = True
    d_fd["covariance_matrix"]["data_file"] = "boss/cov_md_iso.npy"
    m_an, d_an = _aniso_inputs(tmp_path, False)
    setups = [("from_data", m_fd, d_fd, True, [{}, {"assume_isotropic": False}, {"matter_model": "linear_bias", "rsd_model": "kaiser"}]),
              ("boss", *cases.boss_options("config"), True, [{"matter_model": "linear_bias"}, {"empirical_corr": True},
                                                              {"matter_model": "linear_bias", "empirical_corr": True, "rsd_model": "dispersion"},
                                                              {"rsd_model": "euclid_special", "kaiser_coord_shift": False},
                                                              {"rsd_model": "kaiser", "kaiser_approximation": True}]),
              ("aniso_sigma_v", m_an, d_an, False, [{}, {"rsd_model": "dispersion"}, {"rsd_model": "kaiser"}])]
    for tag, model, data, beta, variants in setups:
        fit = victor_amd.CCFFit(model, data)
        ora = vo.OracleFit(model, data)
        hp = cases.halton_params(40, with_beta=beta)
        sub = dict({k: v[:6] for k, v in hp.items()}, bias=2.0, Av=0.6, M=1.04, Q=0.93)
        for kw in variants:
            xi = fit.theory_xi_batch(fit.s, mu, sub, **kw)
            assert fit._get_engine(fit._engine_key(fit._merged(kw))).last_kernel() == "vk_theory_cells_kernel", (tag, kw)
            _native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
            try:
                ref = fit.theory_xi_batch(fit.s, mu, sub, **kw)
            finally:
                _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
            tol = 1e-8 if kw.get("rsd_model") == "dispersion" else 1e-9
            assert xi.shape == ref.shape and np.max(np.abs(xi - ref)) < tol * np.max(np.abs(ref)), (tag, kw)
            for pt, i, j in ((0, 0, 0), (5, 99, len(fit.s) - 1), (2, 37, 4)):
                want = ora.theory_xi(np.array([fit.s[j]]), np.array([mu[i]]), cases.point(sub, pt), **kw)[0, 0]
                assert abs(xi[pt, i, j] - want) < RTOL * max(abs(want), 1e-2), (tag, kw, pt, i, j, xi[pt, i, j], want)
