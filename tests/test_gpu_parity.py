"""GPU parity: HIP path (through the C ABI) vs the reference's golden vectors and the CPU oracle.

Tolerances: the north star asks for multipoles and chi-square within 1e-6 relative.  The tests hold the
HIP path to RTOL = 1e-9 (multipoles are compared relative to max|xi_l| over the vector, because individual
bins cross zero).
"""

import os
import sys

import numpy as np
import pytest

from tests import cases
from tests.devlib import mapped
from tests.tolerances import assert_same_chi2, assert_same_lnl, assert_same_theory, chi2_bound
from victor_amd import _native

pytestmark = pytest.mark.gpu

RTOL = 1e-9
ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.fixture(scope="module")
def oracle():
    sys.path.insert(0, ORACLE_DIR)
    import victor_oracle
    return victor_oracle


@pytest.fixture(scope="module")
def gold():
    return cases.golden_outputs()


@pytest.fixture(scope="module")
def boss_fit():
    import victor_amd
    return {v: victor_amd.CCFFit(*cases.boss_options(v)) for v in ("config", "cobaya")}


@pytest.fixture(scope="module")
def synth_fit():
    import victor_amd
    return {c: victor_amd.CCFFit(*cases.synth_options(c)) for c in (2, 3)}


def vec_close(a, b, rtol=RTOL):
    scale = np.max(np.abs(b))
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) <= rtol * scale


def test_native_library_is_loaded():
    from victor_amd import _native
    lib = _native.load()
    assert lib.vk_device_count() >= 1
    with open("/proc/self/maps") as fh:
        assert "libvictor_hip.so" in fh.read()


@pytest.mark.parametrize("variant", ["config", "cobaya"])
def test_boss_golden_single_point_api(boss_fit, gold, variant):
    g, meta = gold
    fit = boss_fit[variant]
    for i, p in enumerate(meta["boss_points"]):
        lnl, chi2 = fit.log_likelihood(dict(p))
        t = fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s)
        assert vec_close(t, g[f"boss_{variant}_theory"][i]), (variant, i)
        assert abs(chi2 - g[f"boss_{variant}_chi2"][i]) <= RTOL * abs(g[f"boss_{variant}_chi2"][i]), (variant, i)
        assert abs(lnl - g[f"boss_{variant}_lnl"][i]) <= RTOL * abs(g[f"boss_{variant}_lnl"][i]), (variant, i)


def test_boss_known_answer_from_reference_notebook(boss_fit):
    # victor_usage_demo.ipynb:491 prints 65.01, 284.76 for this point
    lnl, chi2 = boss_fit["config"].log_likelihood({"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0})
    assert round(chi2, 2) == 65.01 and round(lnl, 2) == 284.76
    chi2_only, cov = boss_fit["config"].chi_squared({"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0})
    assert_same_chi2(chi2_only, chi2, n_data=60, what="chi_squared vs log_likelihood")
    assert cov.shape == (60, 60)


def test_all_five_published_pairs_under_the_legacy_simpson_rule(boss_fit, oracle):
    """notebooks/victor_usage_demo.ipynb:491-499 on the GPU.  The notebook was produced with SciPy < 1.11, whose ``simps``
    (ccf_model.py:690, default even='avg') differs from SciPy >= 1.11's on the 50 velocity nodes; with
    ``simpson_even='avg'`` all five printed (chi2, lnL) pairs reproduce at two decimals - given per call and at
    construction (model['numerics']) - and match the reference run with that ``simps`` (ref_outputs_avg.npz)."""
    import victor_amd
    ga, meta = cases.golden_outputs("avg")
    model, data = cases.boss_options("config")
    model["numerics"] = {"simpson_even": "avg"}
    fit_avg = victor_amd.CCFFit(model, data)
    assert fit_avg.model["simpson_even"] == "avg" and boss_fit["config"].model["simpson_even"] == "simpson"
    for name, ((chi_nb, lnl_nb), kw) in cases.NOTEBOOK_PRINTED.items():
        for fit, extra in ((fit_avg, {}), (boss_fit["config"], {"simpson_even": "scipy<1.11"})):
            lnl, chi2 = fit.log_likelihood(dict(cases.NOTEBOOK_POINT), **kw, **extra)
            assert round(chi2, 2) == chi_nb and round(lnl, 2) == lnl_nb, name
            assert abs(chi2 - ga[f"boss_nb_{name}"][0]) <= RTOL * chi2, name
            assert abs(lnl - ga[f"boss_nb_{name}"][1]) <= RTOL * abs(lnl), name
    assert fit_avg._get_engine().simpson_even == "avg"
    # the default rule is untouched by the per-call override (separate device contexts)
    lnl, chi2 = boss_fit["config"].log_likelihood(dict(cases.NOTEBOOK_POINT), assume_isotropic=False)
    assert round(chi2, 2) == 64.40 and round(lnl, 2) == 285.05
    # batch + goldens under 'avg': BOSS (cells / point-major kernels) ...
    rows = np.concatenate([fit_avg._fit_rows(dict(p), fit_avg.model) for p in meta["boss_points"]])
    lnl, chi2 = fit_avg.log_likelihood_batch(rows)
    assert np.max(np.abs(chi2 / ga["boss_config_chi2"] - 1)) < RTOL and np.max(np.abs(lnl / ga["boss_config_lnl"] - 1)) < RTOL
    th = fit_avg.theory_vector_batch(rows)
    assert vec_close(th, ga["boss_config_theory"])
    for kw, key in ((dict(rsd_model="dispersion"), "boss_dispersion_theory"), (dict(assume_isotropic=False), "boss_aniso_theory")):
        t = fit_avg.theory_vector_batch(rows[:3], **kw)
        assert vec_close(t, ga[key]), key
    xi = fit_avg.theory_xi(*np.meshgrid(fit_avg.s, np.linspace(0, 1, 100)), dict(meta["boss_points"][0]))
    assert np.max(np.abs(xi - ga["boss_config_xi_smu_p0"])) < RTOL * np.max(np.abs(ga["boss_config_xi_smu_p0"]))
    # ... and the synthetic configs 2 / 3, every fast mapping (the lanes kernel needs a deep launch: tile the points)
    for config in (2, 3):
        model, data = cases.synth_options(config)
        model["numerics"] = {"simpson_even": "avg"}
        fit = victor_amd.CCFFit(model, data)
        pts = list(meta["synth_points"])
        if config == 3:
            pts = [{"fsigma8": 0.47, "sigma_v": 380, "aperp": 1.02, "apar": 0.97}] + pts
        pts = pts[:9]
        rows = np.concatenate([fit._fit_rows(dict(p), fit.model) for p in pts])
        lnl, chi2 = fit.log_likelihood_batch(rows)
        assert np.max(np.abs(chi2 / ga[f"synth{config}_chi2"] - 1)) < RTOL
        assert np.max(np.abs(lnl / ga[f"synth{config}_lnl"] - 1)) < RTOL
        assert vec_close(fit.theory_vector_batch(rows), ga[f"synth{config}_theory"])
        big = np.tile(rows, (4096, 1))
        lnl_b, chi_b = fit.log_likelihood_batch(big)
        assert fit._get_engine().last_kernel() == "vk_theory_cells_kernel"        # the large-batch default since round 3
        assert np.max(np.abs(chi_b.reshape(4096, 9) / ga[f"synth{config}_chi2"] - 1)) < RTOL
        # live oracle on a fresh point under the same rule
        ofit = oracle.OracleFit(model, data)
        p = cases.point(cases.halton_params(40), 37)
        o_lnl, o_chi = ofit.log_likelihood(dict(p))
        g_lnl, g_chi = fit.log_likelihood(dict(p))
        assert abs(g_chi - o_chi) <= RTOL * o_chi and abs(g_lnl - o_lnl) <= RTOL * abs(o_lnl)


def test_boss_batch_equals_single(boss_fit, gold):
    g, meta = gold
    fit = boss_fit["config"]
    pts = meta["boss_points"]
    # rows in ABI column order, built by the class itself
    rows = np.concatenate([fit._fit_rows(dict(p), fit.model) for p in pts])
    lnl, chi2 = fit.log_likelihood_batch(rows)
    assert np.max(np.abs(chi2 / g["boss_config_chi2"] - 1)) < RTOL
    assert np.max(np.abs(lnl / g["boss_config_lnl"] - 1)) < RTOL


def test_boss_theory_xi_grid(boss_fit, gold):
    g, meta = gold
    fit = boss_fit["config"]
    mu = np.linspace(0, 1, 100)
    for tag, idx in (("p0", 0), ("p2", 2)):
        xi = fit.theory_xi(*np.meshgrid(fit.s, mu), dict(meta["boss_points"][idx]))
        assert fit._get_engine().last_kernel() == "vk_theory_cells_kernel"          # round 5: theory_xi on the fast kernels
        ref = g[f"boss_config_xi_smu_{tag}"]
        assert xi.shape == ref.shape == (100, 30)
        assert np.max(np.abs(xi - ref)) < RTOL * np.max(np.abs(ref))


def test_theory_xi_on_the_cells_kernel(synth_fit, boss_fit, oracle):
    """CCFModel.theory_xi (ccf_model.py:538-690) is served by the cells kernel's store-every-cell form: every RSD model, per-point
    (beta-dependent) and batch-constant tables, mu grids of any length and sign (no projection: the n_mu >= 64 of the projected
    launches does not apply), batches whose points are cut into ranges and batches of whole points - against the generic kernel
    on the whole array and against the oracle on single cells."""
    cases_ = [(boss_fit["config"], True, "boss"), (synth_fit[3], False, "config3")]
    grids = [(None, np.linspace(0, 1, 100)), (np.linspace(3.0, 110.0, 17), np.linspace(-1, 1, 41)),
             (np.array([5.0, 11.5, 40.0, 77.7]), np.array([0.0, 0.3, 1.0]))]
    for fit, beta, tag in cases_:
        ora = oracle.OracleFit(*(cases.boss_options("config") if beta else cases.synth_options(3)))
        hp = cases.halton_params(300, with_beta=beta)
        for rsd in ("streaming", "dispersion", "kaiser", "euclid_special"):
            kw = {"rsd_model": rsd}
            for s, mu in grids:
                s = fit.s if s is None else s
                for n in (1, 5, 300):
                    sub = {k: v[:n] for k, v in hp.items()}
                    xi = fit.theory_xi_batch(s, mu, sub, **kw)
                    assert fit._get_engine().last_kernel() == "vk_theory_cells_kernel", (tag, rsd, len(mu), n)
                    assert xi.shape == (n, len(mu), len(s)) and np.all(np.isfinite(xi))
                    _native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
                    try:
                        ref = fit.theory_xi_batch(s, mu, sub, **kw)
                        assert fit._get_engine().last_kernel() == "vk_xi_smu_kernel"
                    finally:
                        _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
                    tol = 1e-9 if rsd != "dispersion" else 1e-8           # (the dispersion iteration amplifies roundings near r = 0)
                    assert np.max(np.abs(xi - ref)) < tol * np.max(np.abs(ref)), (tag, rsd, len(mu), n)
                    if n == 5:
                        for pt, i, j in ((0, 0, 0), (3, len(mu) - 1, len(s) - 1), (4, len(mu) // 2, 1)):
                            want = ora.theory_xi(np.array([s[j]]), np.array([mu[i]]), cases.point(sub, pt), **kw)[0, 0]
                            assert abs(xi[pt, i, j] - want) < RTOL * max(abs(want), 1e-2), (tag, rsd, pt, i, j, xi[pt, i, j], want)
        # grids at the edges of what the kernel takes: one s bin, two mu nodes; a long mu grid; more cells than one range holds
        for s, mu, want in ((np.array([33.3]), np.array([0.1, 0.9]), "vk_theory_cells_kernel"),
                            (np.linspace(2.0, 118.0, 59), np.linspace(0, 1, 1000), "vk_theory_cells_kernel"),
                            (fit.s, np.linspace(0, 1, 1025), "vk_xi_smu_kernel")):       # n_mu > 1024: the generic kernel's
            sub = {k: v[:2] for k, v in hp.items()}
            xi = fit.theory_xi_batch(s, mu, sub)
            assert fit._get_engine().last_kernel() == want, (tag, len(s), len(mu))
            _native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
            try:
                ref = fit.theory_xi_batch(s, mu, sub)
            finally:
                _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
            assert xi.shape == ref.shape == (2, len(mu), len(s))
            assert np.max(np.abs(xi - ref)) < 1e-9 * np.max(np.abs(ref)), (tag, len(s), len(mu))
        # theory_xi projected on the host with the fit's own weights IS the theory vector (utils.multipoles_from_fn restated as
        # W_l[100], victor_amd/tables.py): the stored cells against the kernel's own projection, 256 points, every multipole
        from victor_amd import tables as T
        poles = np.atleast_1d(fit.poles_s)
        mu_p = T.mu_nodes_for(poles)
        W = T.projection_weights(mu_p, poles)                                   # (n_ell, n_mu)
        sub = {k: v[:256] for k, v in hp.items()}
        xi = fit.theory_xi_batch(fit.s, mu_p, sub)                              # (256, n_mu, n_s)
        th = fit.theory_vector_batch(sub).reshape(256, len(poles), len(fit.s))
        proj = np.einsum("li,nij->nlj", W, xi)
        scale = np.max(np.abs(th), axis=2, keepdims=True)
        assert np.max(np.abs(proj - th) / scale) < 1e-11, tag
        # a NaN parameter poisons every cell of its point and nothing else
        bad = {k: v[:3].copy() for k, v in hp.items()}
        bad["sigma_v"][1] = np.nan
        xi = fit.theory_xi_batch(fit.s, np.linspace(0, 1, 100), bad)
        assert np.all(np.isnan(xi[1])) and np.all(np.isfinite(xi[0])) and np.all(np.isfinite(xi[2]))


def test_boss_anisotropic_kwarg(boss_fit, gold):
    g, meta = gold
    fit = boss_fit["config"]
    for i, p in enumerate(meta["boss_points"][:3]):
        t = fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s, assume_isotropic=False)
        assert vec_close(t, g["boss_aniso_theory"][i])
    lnl, chi2 = fit.log_likelihood(dict(meta["boss_points"][0]), assume_isotropic=False)
    assert abs(chi2 - g["boss_nb_anisotropic"][0]) < RTOL * chi2
    assert abs(lnl - g["boss_nb_anisotropic"][1]) < RTOL * abs(lnl)


def test_boss_beta_likelihood_interpolation(boss_fit, gold):
    g, meta = gold
    lnl, chi2 = boss_fit["config"].log_likelihood(dict(meta["boss_points"][0]), beta_interpolation="likelihood")
    assert abs(chi2 - g["boss_nb_beta_likelihood"][0]) < RTOL * chi2
    assert abs(lnl - g["boss_nb_beta_likelihood"][1]) < RTOL * abs(lnl)


@pytest.mark.parametrize("form", ["gaussian", "hartlap", "percival", "sellentin"])
def test_boss_likelihood_forms(boss_fit, gold, form):
    g, meta = gold
    lnl, chi2 = boss_fit["config"].log_likelihood(dict(meta["boss_points"][0]),
                                                  likelihood={"form": form, "nmocks": 1000, "nparams": 4})
    assert abs(chi2 - g[f"boss_form_{form}"][0]) < RTOL * chi2
    assert abs(lnl - g[f"boss_form_{form}"][1]) < RTOL * abs(lnl)


@pytest.mark.parametrize("config", [2, 3])
def test_synthetic_golden(synth_fit, gold, config):
    g, meta = gold
    fit = synth_fit[config]
    pts = list(meta["synth_points"])
    if config == 3:
        pts = [{"fsigma8": 0.47, "sigma_v": 380, "aperp": 1.02, "apar": 0.97}] + pts
    batch = {k: np.array([p[k] for p in pts]) for k in pts[0]}
    lnl, chi2 = fit.log_likelihood_batch(batch)
    th = fit.theory_vector_batch(batch)
    ref_t = g[f"synth{config}_theory"]
    assert th.shape == ref_t.shape
    for i in range(len(pts)):
        assert vec_close(th[i], ref_t[i]), i
    assert np.max(np.abs(chi2 / g[f"synth{config}_chi2"] - 1)) < RTOL
    assert np.max(np.abs(lnl / g[f"synth{config}_lnl"] - 1)) < RTOL
    xi = fit.theory_xi(fit.s, np.linspace(0, 1, 100), dict(pts[0]))
    ref = g[f"synth{config}_xi_smu_p0"]
    assert np.max(np.abs(xi - ref)) < RTOL * np.max(np.abs(ref))


def test_oracle_parity_on_fresh_halton_points(synth_fit, boss_fit, oracle):
    """Seeded points that are NOT in the fixtures: HIP vs the oracle run live (sizes the oracle finishes fast)."""
    for config in (2, 3):
        fit = synth_fit[config]
        ora = oracle.OracleFit(*cases.synth_options(config))
        hp = cases.halton_params(300)
        sel = [40, 123, 299]
        batch = {k: v[sel] for k, v in hp.items()}
        lnl, chi2 = fit.log_likelihood_batch(batch)
        for j, i in enumerate(sel):
            lo, co = ora.log_likelihood(cases.point(hp, i))
            assert abs(chi2[j] / co - 1) < RTOL and abs(lnl[j] / lo - 1) < RTOL
    fit = boss_fit["config"]
    ora = oracle.OracleFit(*cases.boss_options("config"))
    hp = cases.halton_params(200, with_beta=True)
    sel = [7, 77, 177]
    batch = {k: v[sel] for k, v in hp.items()}
    lnl, chi2 = fit.log_likelihood_batch(batch)
    for j, i in enumerate(sel):
        lo, co = ora.log_likelihood(cases.point(hp, i))
        assert abs(chi2[j] / co - 1) < RTOL and abs(lnl[j] / lo - 1) < RTOL


def test_batch_split_modes_agree(synth_fit):
    """The kernel picks a different work split for small / medium / large batches; results must not depend on it."""
    fit = synth_fit[3]
    hp = cases.halton_params(1200)
    lnl_big, chi_big = fit.log_likelihood_batch(hp)                       # one workgroup per point
    bound = chi2_bound(fit, hp)
    for n in (1, 3, 30, 300):                                             # split s bins / split the (mu, v) plane
        sub = {k: v[:n] for k, v in hp.items()}
        lnl, chi = fit.log_likelihood_batch(sub)
        assert_same_chi2(chi, chi_big[:n], bound[:n], what=f"batch split {n}")
        assert_same_lnl(lnl, lnl_big[:n], bound[:n], what=f"batch split {n}")


def test_general_s_grid_and_poles(synth_fit, oracle):
    fit = synth_fit[3]
    ora = oracle.OracleModel(cases.synth_options(3)[0])
    s = np.array([5.0, 17.5, 33.0, 61.0, 90.0, 110.0])
    p = {"fsigma8": 0.5, "sigma_v": 300, "aperp": 1.03, "apar": 0.96}
    got = fit.theory_multipoles(s, dict(p), poles=[0, 2, 4])
    want, _ = ora.theory_multipoles(s, dict(p), poles=[0, 2, 4])
    for key in ("0", "2", "4"):
        assert np.max(np.abs(got[key] - want[key])) < RTOL * np.max(np.abs(want[key]))
    got = fit.theory_multipoles(s, dict(p), poles=[0])
    assert set(got) == {"0"}
    assert np.max(np.abs(got["0"] - want["0"])) < RTOL * np.max(np.abs(want["0"]))


def test_general_grid_batches_through_every_work_split(synth_fit, boss_fit):
    """``theory_multipoles_batch`` on a caller's s grid (13 bins, not the data's 40 / 30; l = 0, 2, 4 or odd l on the full mu range)
    for batch sizes that take the plane-split point-major kernel, the cell ranges and whole-point workgroups: every size must
    reproduce the rows of one large batch (the partial-sum slots and counters are sized by the context's own grid)."""
    s = np.linspace(4.0, 112.0, 13)
    for fit, beta in ((synth_fit[3], False), (boss_fit["config"], True)):
        hp = cases.halton_params(2100, with_beta=beta)
        for poles in ([0, 2, 4], [0, 1]):
            big = fit.theory_multipoles_batch(s, hp, poles)
            assert big.shape == (2100, len(poles), 13) and np.all(np.isfinite(big))
            for n in (1, 2, 7, 30, 200, 1500):
                sub = fit.theory_multipoles_batch(s, {k: v[:n] for k, v in hp.items()}, poles)
                assert_same_theory(sub, big[:n], what=f"general grid {beta} {poles} {n}")
    # a single multipole through the cells kernel (one sum per trip in its projection, not two or three)
    hp = cases.halton_params(300)
    mono = synth_fit[3].theory_multipoles_batch(s, hp, [0])
    full = synth_fit[3].theory_multipoles_batch(s, hp, [0, 2, 4])
    assert mono.shape == (300, 1, 13)
    assert_same_theory(mono[:, 0], full[:, 0], what="one multipole vs three")
    # and the reference-style scalar call
    one = synth_fit[3].theory_multipoles(s, cases.point(cases.halton_params(3), 2), poles=[0, 2, 4])
    assert set(one) == {"0", "2", "4"} and one["2"].shape == (13,)


def test_large_batch_properties(synth_fit):
    """Full-size batch (BASELINE config 3: 65536 points): size-independent checks."""
    fit = synth_fit[3]
    n = 65536
    hp = cases.halton_params(n)
    lnl, chi2 = fit.log_likelihood_batch(hp)
    assert lnl.shape == chi2.shape == (n,)
    assert np.all(np.isfinite(lnl)) and np.all(chi2 > 0)
    assert np.max(np.abs(lnl + 0.5 * chi2)) < 1e-9 * np.max(chi2)          # gaussian form, fixed covariance
    # any permutation of the batch gives the permuted result bit for bit
    perm = np.random.default_rng(0).permutation(n)
    lnl_p, chi_p = fit.log_likelihood_batch({k: v[perm] for k, v in hp.items()})
    assert np.array_equal(chi_p, chi2[perm])
    # a sub-batch run on its own (different work split) agrees
    idx = np.arange(0, n, 4099)
    lnl_s, chi_s = fit.log_likelihood_batch({k: v[idx] for k, v in hp.items()})
    sub = {k: v[idx] for k, v in hp.items()}
    assert_same_chi2(chi_s, chi2[idx], chi2_bound(fit, sub), what="sub-batch of the 65536")


def test_full_size_linearity_in_the_real_space_ccf(tmp_path):
    """xi^s + 1 = int (1 + xi^r) pdf dv is affine in the real-space multipoles: with the tables scaled by 0, 1 and 2 the
    theory vectors of the full 65536-point batch must satisfy T(2 xi) - 2 T(xi) + T(0) = 0 (size-independent check of
    the whole theory kernel; the velocity tables come from the matter template and do not change)."""
    import victor_amd
    src = np.load(os.path.join(cases.GOLDEN, "synth", "model.npy"), allow_pickle=True).item()
    n = 65536
    hp = cases.halton_params(n)
    th = {}
    for scale in (0.0, 1.0, 2.0):
        tab = dict(src)
        for key in ("monopole", "quadrupole", "hexadecapole"):
            tab[key] = scale * np.asarray(src[key])
        np.save(tmp_path / f"model_x{int(scale)}.npy", tab, allow_pickle=True)
        model, data = cases.synth_options(3)
        model = dict(model, dir=str(tmp_path), input_model_data_file=f"model_x{int(scale)}.npy")
        th[scale] = victor_amd.CCFFit(model, data).theory_vector_batch(hp)
        assert th[scale].shape == (n, 120)
    # three rounded theory vectors combined: four times the two-vector bound of tests/tolerances.py
    assert_same_theory(th[2.0] + th[0.0], 2.0 * th[1.0], what="linearity in xi^r at 65536 points", ulps=2048)
    assert np.max(np.abs(th[1.0] - th[0.0])) > 1e-3          # the tables do matter


def test_failure_guards(boss_fit):
    fit = boss_fit["config"]
    lnl, chi2 = fit.log_likelihood({"fsigma8": np.nan, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0})
    assert lnl == -np.inf and chi2 == np.inf                                # ccf_fit.py:477-481
    from victor_amd import InputError
    with pytest.raises(InputError):
        fit.log_likelihood({"fsigma8": 0.47, "beta": 0.37}, rsd_model="nonsense")
    with pytest.raises(KeyError):
        fit.log_likelihood({"beta": 0.37})
    with pytest.raises(InputError):
        fit.log_likelihood({"fsigma8": 0.47, "beta": 0.37}, likelihood={"form": "bogus"})


# --------------------------------------------------------------------------- other RSD mappings
@pytest.mark.parametrize("rsd", ["dispersion", "kaiser", "euclid_special"])
def test_other_rsd_models_golden(boss_fit, gold, rsd):
    g, meta = gold
    fit = boss_fit["config"]
    for i, p in enumerate(meta["boss_points"][:3]):
        t = fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s, rsd_model=rsd)
        assert vec_close(t, g[f"boss_{rsd}_theory"][i]), (rsd, i)


def test_kaiser_linearised_with_nuisance_parameters(boss_fit, gold):
    g, meta = gold
    fit = boss_fit["config"]
    for i, p in enumerate(meta["boss_points"][:3]):
        t = fit.theory_multipole_vector(fit.s, dict(p, M=1.1, Q=0.9), fit.poles_s, rsd_model="kaiser",
                                        kaiser_approximation=True)
        assert vec_close(t, g["boss_kaiser_approx_theory"][i]), i


def test_notebook_dispersion_and_kaiser_likelihoods(boss_fit, gold):
    g, meta = gold
    fit = boss_fit["config"]
    p = meta["boss_points"][0]
    for name, printed in (("dispersion", (65.03, 284.76)), ("kaiser", (103.90, 266.81))):
        lnl, chi2 = fit.log_likelihood(dict(p), rsd_model=name)
        assert abs(chi2 - g[f"boss_nb_{name}"][0]) < RTOL * chi2 and abs(lnl - g[f"boss_nb_{name}"][1]) < RTOL * abs(lnl)
        assert (round(chi2, 2), round(lnl, 2)) == printed            # victor_usage_demo.ipynb:493,495


def test_other_rsd_models_vs_oracle_on_synthetic(synth_fit, oracle):
    fit = synth_fit[3]
    ora = oracle.OracleFit(*cases.synth_options(3))
    hp = cases.halton_params(50)
    for rsd, kw in (("dispersion", {}), ("kaiser", {}), ("kaiser", {"kaiser_coord_shift": False}),
                    ("euclid_special", {}), ("dispersion", {"niter": 2})):
        p = dict(cases.point(hp, 17), M=0.95, Q=1.2)
        got = fit.log_likelihood(dict(p), rsd_model=rsd, **kw)
        want = ora.log_likelihood(dict(p), rsd_model=rsd, **kw)
        assert abs(got[1] / want[1] - 1) < RTOL and abs(got[0] / want[0] - 1) < RTOL, (rsd, kw)
        xi = fit.theory_xi(fit.s, np.linspace(0, 1, 100), dict(p), rsd_model=rsd, **kw)
        xo = ora.theory_xi(ora.s, np.linspace(0, 1, 100), dict(p), rsd_model=rsd, **kw)
        assert np.max(np.abs(xi - xo)) < RTOL * np.max(np.abs(xo)), (rsd, kw)


def test_generic_kernel_matches_fast_kernel(synth_fit, boss_fit):
    """VICTOR_HIP_FORCE_GENERIC routes streaming through the generic kernel (library sqrt/div/exp, knot search)."""
    hp = cases.halton_params(64)
    hb = cases.halton_params(64, with_beta=True)
    a3 = synth_fit[3].log_likelihood_batch(hp)
    ab = boss_fit["config"].log_likelihood_batch(hb)
    _native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
    try:
        b3 = synth_fit[3].log_likelihood_batch(hp)
        bb = boss_fit["config"].log_likelihood_batch(hb)
    finally:
        _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
    # another arithmetic (library sqrt / division / exp, knot search): 1024 roundings instead of 64 (tests/tolerances.py)
    assert_same_chi2(a3[1], b3[1], chi2_bound(synth_fit[3], hp, ulps=1024), what="generic vs fast, config 3")
    assert_same_chi2(ab[1], bb[1], chi2_bound(boss_fit["config"], hb, ulps=1024), what="generic vs fast, BOSS")
    assert not np.array_equal(a3[1], b3[1])          # different arithmetic, so not bit-identical: both paths really ran


def test_lanes_over_batch_mapping_matches_point_major(synth_fit, gold):
    """The two work mappings of the theory kernel (wave = point x s-bin vs wave = s-bin x 64 points) must agree."""
    g, meta = gold
    out = {}
    hp = cases.halton_params(8192 + 37)              # not a multiple of 64: exercises the ragged last chunk
    for config in (2, 3):
        fit = synth_fit[config]
        for mapping in ("point", "lanes", "cells"):
            with mapped(fit, mapping) as f:
                out[mapping] = f.log_likelihood_batch(hp)
                assert f._get_engine().last_kernel() == {"point": "vk_theory_fast_kernel", "lanes": "vk_theory_lanes_kernel",
                                                           "cells": "vk_theory_cells_kernel"}[mapping]
            # golden points through each mapping as well
            pts = list(meta["synth_points"])
            if config == 3:
                pts = [{"fsigma8": 0.47, "sigma_v": 380, "aperp": 1.02, "apar": 0.97}] + pts
            batch = {k: np.array([p[k] for p in pts]) for k in pts[0]}
            with mapped(fit, mapping) as f:
                lnl, chi2 = f.log_likelihood_batch(batch)
            assert np.max(np.abs(chi2 / g[f"synth{config}_chi2"] - 1)) < RTOL, (config, mapping)
        bound = chi2_bound(fit, hp)
        assert_same_chi2(out["lanes"][1], out["point"][1], bound, what=f"lanes vs point, config {config}")
        assert_same_chi2(out["cells"][1], out["point"][1], bound, what=f"cells vs point, config {config}")
        assert not np.array_equal(out["point"][1], out["lanes"][1])


# --------------------------------------------------------------------------- edge cases
def test_empty_and_ragged_batches(synth_fit):
    fit = synth_fit[3]
    hp = cases.halton_params(200)
    lnl, chi2 = fit.log_likelihood_batch({k: v[:0] for k, v in hp.items()})
    assert lnl.shape == (0,) and chi2.shape == (0,)
    assert fit.theory_vector_batch({k: v[:0] for k, v in hp.items()}).shape == (0, 120)
    full = fit.log_likelihood_batch(hp)
    bound = chi2_bound(fit, hp)
    for n in (1, 2, 63, 64, 65, 127, 129):
        for mapping in ("point", "lanes", "cells"):
            with mapped(fit, mapping) as f:
                lnl, chi2 = f.log_likelihood_batch({k: v[:n] for k, v in hp.items()})
            assert lnl.shape == (n,)
            assert_same_chi2(chi2, full[1][:n], bound[:n], what=f"ragged {n} {mapping}")


def test_bad_rows_do_not_contaminate_neighbours(synth_fit, boss_fit):
    """A NaN / inf parameter row yields (-inf, +inf) for that row only (ccf_fit.py:477-481), in both mappings."""
    fit = synth_fit[3]
    hp = cases.halton_params(9000)
    good = fit.log_likelihood_batch(hp)
    bound = chi2_bound(fit, hp)
    bad = {k: v.copy() for k, v in hp.items()}
    bad["fsigma8"][5] = np.nan
    bad["sigma_v"][77] = np.inf
    bad["aperp"][8999] = np.nan
    for mapping in ("point", "lanes", "cells"):
        with mapped(fit, mapping) as f:
            lnl, chi2 = f.log_likelihood_batch(bad)
        for i in (5, 77, 8999):
            assert lnl[i] == -np.inf and chi2[i] == np.inf, (mapping, i)
        keep = np.ones(9000, bool)
        keep[[5, 77, 8999]] = False
        assert_same_chi2(chi2[keep], good[1][keep], bound[keep], what=f"neighbours of bad rows, {mapping}")
    rows = boss_fit["config"]._fit_rows(cases.halton_params(50, with_beta=True), boss_fit["config"].model)
    rows[7, 5] = np.nan                                   # beta
    lnl, chi2 = boss_fit["config"].log_likelihood_batch(rows)
    assert lnl[7] == -np.inf and np.all(np.isfinite(np.delete(lnl, 7)))


def test_extreme_but_valid_parameters_vs_oracle(synth_fit, oracle):
    """Corners of parameter space that push the integrand into its clamps: tiny dispersion (pdf far in the tails,
    exp underflow), strong AP distortion (r beyond every table), very large growth rate."""
    fit = synth_fit[3]
    ora = oracle.OracleFit(*cases.synth_options(3))
    pts = [{"fsigma8": 1.5, "sigma_v": 40.0, "aperp": 0.8, "apar": 1.2},
           {"fsigma8": 0.05, "sigma_v": 900.0, "aperp": 1.35, "apar": 0.7},
           {"fsigma8": 3.0, "sigma_v": 100.0, "aperp": 1.0, "apar": 1.0},
           {"fsigma8": 0.0, "sigma_v": 250.0, "aperp": 0.6, "apar": 1.5}]
    for p in pts:
        got = fit.log_likelihood(dict(p))
        want = ora.log_likelihood(dict(p))
        assert np.isfinite(want[1])
        assert abs(got[1] / want[1] - 1) < RTOL and abs(got[0] / want[0] - 1) < RTOL, p
        t = fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s)
        to = ora.theory_multipole_vector(ora.s, dict(p), ora.poles_s)
        assert vec_close(t, to), p


def test_cells_mapping_on_beta_dependent_tables(boss_fit, gold):
    """BOSS (per-point xi^r tables): the cells kernel against the point-major kernel and the reference goldens."""
    g, meta = gold
    for variant in ("config", "cobaya"):
        fit = boss_fit[variant]
        rows = np.concatenate([fit._fit_rows(dict(p), fit.model) for p in meta["boss_points"]])
        res = {}
        for mapping in ("point", "cells"):
            with mapped(fit, mapping) as f:
                res[mapping] = f.log_likelihood_batch(rows)
                th = f.theory_vector_batch(rows, assume_isotropic=False)
            assert np.max(np.abs(res[mapping][1] / g[f"boss_{variant}_chi2"] - 1)) < RTOL, (variant, mapping)
            if variant == "config":
                assert vec_close(th[:3], g["boss_aniso_theory"]), mapping
        assert_same_chi2(res["cells"][1], res["point"][1], chi2_bound(fit, rows), what=f"cells vs point, BOSS {variant}")
    hb = cases.halton_params(3000, with_beta=True)
    fit = boss_fit["config"]
    a = fit.log_likelihood_batch(hb)                      # default choice at this size: cells
    assert fit._get_engine().last_kernel() == "vk_theory_cells_kernel"
    _native.set_knob("VICTOR_HIP_MAPPING", "point")
    try:
        b = fit.log_likelihood_batch(hb)
    finally:
        _native.set_knob("VICTOR_HIP_MAPPING", None)
    bound = chi2_bound(fit, hb)
    assert_same_chi2(a[1], b[1], bound, what="cells vs point, BOSS 3000")
    assert_same_lnl(a[0], b[0], bound, what="cells vs point, BOSS 3000")


def test_odd_multipoles_use_the_full_mu_range(synth_fit, oracle):
    """poles containing an odd l switch the mu grid to [-1, 1] and halve the prefactor (ccf_model.py:816-819)."""
    fit = synth_fit[3]
    ora = oracle.OracleModel(cases.synth_options(3)[0])
    s = np.array([6.0, 20.0, 41.0, 77.0])
    p = {"fsigma8": 0.6, "sigma_v": 300, "aperp": 1.04, "apar": 0.95}
    got = fit.theory_multipoles(s, dict(p), poles=[0, 1, 2])
    want, _ = ora.theory_multipoles(s, dict(p), poles=[0, 1, 2])
    assert set(got) == {"0", "1", "2"}
    scale = max(np.max(np.abs(want[k])) for k in want)
    for key in want:
        assert np.max(np.abs(got[key] - want[key])) < RTOL * scale, key
    assert np.max(np.abs(got["1"])) < 1e-6 * scale          # xi(s, mu) is even in mu, so the dipole vanishes
    # theory_xi on an unsorted grid with duplicates sorts and de-duplicates 2-D inputs (np.unique, ccf_model.py:577)
    S, M = np.meshgrid([20.0, 6.0, 20.0], [0.9, 0.1, 0.5])
    xi = fit.theory_xi(S, M, dict(p))
    xo = ora.theory_xi(np.array([6.0, 20.0]), np.array([0.1, 0.5, 0.9]), dict(p))
    assert xi.shape == (3, 2) and np.max(np.abs(xi - xo)) < RTOL * np.max(np.abs(xo))


class knobs:
    """Set VICTOR_HIP_* knobs for the duration of a ``with`` block."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            _native.set_knob("VICTOR_HIP_" + k, v)

    def __exit__(self, *exc):
        for k in self.kw:
            _native.set_knob("VICTOR_HIP_" + k, None)
        return False


def test_tiled_likelihood_kernel_matches_per_point_kernel(synth_fit):
    """Fixed covariance: the 8-points-per-wave chi-square kernel vs the one-point-per-wave kernel vs one workgroup per
    point (all three as separate launches: NO_FUSE)."""
    for config in (2, 3):
        fit = synth_fit[config]
        hp = cases.halton_params(1003)                       # not a multiple of the tile
        with knobs(NO_FUSE="1", LIKE_WIDE="0"):
            a = fit.log_likelihood_batch(hp)
            with knobs(LIKE_UNTILED="1"):
                b = fit.log_likelihood_batch(hp)
        with knobs(NO_FUSE="1", LIKE_WIDE="1"):
            c = fit.log_likelihood_batch(hp)
        bound = chi2_bound(fit, hp)
        for other in (b, c):
            assert_same_chi2(other[1], a[1], bound, what=f"K2 variants, config {config}")
            assert_same_lnl(other[0], a[0], bound, what=f"K2 variants, config {config}")
        for form in ("sellentin", "hartlap", "percival"):
            kw = {"likelihood": {"form": form, "nmocks": 800, "nparams": 4}}
            sub = {k: v[:200] for k, v in hp.items()}
            with knobs(NO_FUSE="1", LIKE_WIDE="0"):
                a = fit.log_likelihood_batch(sub, **kw)
                with knobs(LIKE_UNTILED="1"):
                    b = fit.log_likelihood_batch(sub, **kw)
            c = fit.log_likelihood_batch(sub, **kw)                     # default: fused into the theory kernel
            assert_same_lnl(b[0], a[0], bound[:200], what=f"K2 variants {form}")
            assert_same_lnl(c[0], a[0], bound[:200], what=f"K2 fused {form}")


@pytest.mark.parametrize("which", ["boss", "config3", "config2"])
def test_fused_and_split_launches_agree_with_separate_kernels(boss_fit, synth_fit, which):
    """Small and medium batches take the chi-square inside the theory kernel (the workgroup that completes a point's
    theory vector - found through a per-point completion counter when several workgroups share the point - computes it),
    and a handful of points is split over more workgroups (s bins x parts of the (mu, v) plane).  Every split and both
    kernels must agree with the plain two-launch path, launch after launch (the counters reset themselves)."""
    fit = boss_fit["config"] if which == "boss" else synth_fit[int(which[-1])]
    hp = cases.halton_params(700, with_beta=(which == "boss"))
    rows = fit._fit_rows(hp, fit.model)
    eng = fit._get_engine()
    with knobs(NO_FUSE="1", NO_GRAPH="1"):
        ref_l, ref_c = fit.log_likelihood_batch(rows)
        ref_t = fit.theory_vector_batch(rows)
    bound = chi2_bound(fit, rows)

    def check(n, tag, want_kernel=None):
        for rep in range(2):                                               # second launch: counters must be back at zero
            lnl, chi = fit.log_likelihood_batch(rows[:n])
            assert_same_chi2(chi, ref_c[:n], bound[:n], what=f"{which} {tag} n={n} rep={rep}")
            assert_same_lnl(lnl, ref_l[:n], bound[:n], what=f"{which} {tag} n={n} rep={rep}")
        if want_kernel:
            assert eng.last_kernel() == want_kernel, (tag, eng.last_kernel())
        lnl, chi, th = eng.eval_batch(eng.make_opts(fit.model, fit.fit_options), rows[:n], want_theory=True)
        assert_same_theory(th, ref_t[:n], what=f"{which} {tag} n={n}")             # a call that asks for the theory vector gets it
        assert_same_chi2(chi, ref_c[:n], bound[:n], what=f"{which} {tag} n={n} with theory")

    with knobs(NO_GRAPH="1"):
        for n in (1, 2, 3, 17):                                            # default choices: a handful of points ...
            check(n, "default", "vk_theory_fast_kernel")
        for n in (24, 64, 200, 700):                                       # ... and cell ranges from two dozen on
            check(n, "default", "vk_theory_cells_kernel")
        for split in ("1,4,1", "1,4,2", "1,4,4", "1,4,8", "1,2,3", "1,1,2", "4,1,1", "4,1,2", "40,1,1", "40,1,4"):
            with knobs(SPLIT=split, MAPPING="point"):
                for n in (1, 5, 64):
                    check(n, "split " + split, "vk_theory_fast_kernel")
        for parts in ("1", "2", "3", "4", "7", "11", "16"):
            with knobs(CELLS_PARTS=parts, MAPPING="cells"):
                for n in (1, 6, 130, 700):
                    check(n, "cells parts " + parts, "vk_theory_cells_kernel")
        with knobs(POINT_CAP="1"):                                          # 256 workgroups loop over many items each
            check(700, "capped cells")
            with knobs(MAPPING="point"):
                check(300, "capped point-major")
    # the hipGraph replay of host-buffer batches contains the fused launch
    for rep in range(4):
        lnl, chi = fit.log_likelihood_batch(rows[:9])
        assert_same_chi2(chi, ref_c[:9], bound[:9], what=f"{which} graph replay")


def test_determinant_sign_semantics_of_the_blended_covariance(tmp_path, oracle):
    """ccf_fit.py:447-450 rejects a point when np.linalg.slogdet of the blended covariance has sign != +1.  With a last
    slice that has TWO negative eigenvalues the blend passes through zero, one negative factor (sign -1: rejected) and two
    negative factors (sign +1: accepted with log|det|); the device decides by the parity of the negative factors
    1 - t + t lambda_i and must agree with the reference's rule (the oracle calls slogdet) in each regime, in the fused tail
    and in the separate chi-square kernels."""
    import victor_amd
    model, data = cases.boss_options("config")
    cov = np.load(os.path.join(cases.GOLDEN, "boss", "cov.npy"), allow_pickle=True).item()
    stack = np.array(cov["covmat"], dtype=float)
    w, v = np.linalg.eigh(stack[-1])
    # eigenvalues w5 -> -w5 / 4 and w40 -> -4 w40: the factors of the two directions turn negative at t ~ 0.8 and t ~ 0.2
    flipped = stack[-1] - 1.25 * w[5] * np.outer(v[:, 5], v[:, 5]) - 5.0 * w[40] * np.outer(v[:, 40], v[:, 40])
    stack[-1] = 0.5 * (flipped + flipped.T)
    assert np.linalg.slogdet(stack[-1])[0] == 1 and np.sum(np.linalg.eigvalsh(stack[-1]) < 0) == 2
    path = str(tmp_path / "cov_two_negative.npy")
    np.save(path, {"beta": cov["beta"], "covmat": stack}, allow_pickle=True)
    data = cases.clone(data)
    data["covariance_matrix"]["data_file"] = path
    fit = victor_amd.CCFFit(model, data)
    ofit = oracle.OracleFit(model, data)
    g = np.asarray(cov["beta"], dtype=float)
    # only the last interval of the grid lets the weight t of the last slice run from 0 to 1 (the lower bracket moves with beta)
    betas = np.concatenate([g[-2] + (g[-1] - g[-2]) * np.linspace(0.02, 0.98, 49), [g[0] - 0.01, g[-1] + 0.01, g[7], 0.5 * (g[3] + g[4])]])
    hp = cases.halton_params(len(betas))
    rows = fit._fit_rows(dict(hp, beta=betas), fit.model)
    want = [ofit.log_likelihood(dict(cases.point(hp, i), beta=float(b))) for i, b in enumerate(betas)]
    want_l = np.array([x[0] for x in want])
    rejected = np.isneginf(want_l)
    assert 10 <= rejected.sum() <= len(betas) - 15       # the sweep crosses all three regimes
    assert not rejected[0] and not rejected[48] and rejected[24]
    for kn in ({}, {"NO_FUSE": "1"}, {"NO_FUSE": "1", "LIKE_WIDE": "0"}, {"MAPPING": "cells"}):
        with knobs(NO_GRAPH="1", **kn):
            lnl, chi2 = fit.log_likelihood_batch(rows)
        assert np.array_equal(np.isneginf(lnl), rejected), kn
        assert np.all(np.isposinf(chi2[rejected]))
        ok = ~rejected
        assert np.max(np.abs(lnl[ok] - want_l[ok]) / (np.abs(want_l[ok]) + 300.0)) < 1e-9, kn


def test_fused_path_keeps_the_failure_guards(boss_fit):
    """NaN parameters and the singular-covariance guard return (-inf, inf) from the fused tail as from the K2 kernels."""
    fit = boss_fit["config"]
    rows = fit._fit_rows(cases.halton_params(6, with_beta=True), fit.model)
    rows[2, _native.P_SIGMAV] = np.nan
    rows[4, _native.P_FSIGMA8] = np.inf
    for kn in ({}, {"NO_FUSE": "1"}, {"MAPPING": "cells"}, {"SPLIT": "1,4,4", "MAPPING": "point"}):
        with knobs(NO_GRAPH="1", **kn):
            lnl, chi = fit.log_likelihood_batch(rows)
        assert np.all(np.isneginf(lnl[[2, 4]])) and np.all(np.isposinf(chi[[2, 4]])), kn
        assert np.all(np.isfinite(lnl[[0, 1, 3, 5]])) and np.all(chi[[0, 1, 3, 5]] > 0), kn


def test_point_major_fused_launch_beyond_the_counter_array(boss_fit):
    """A BOSS batch larger than the completion-counter array (16384 points) through the point-major kernel, whose chi-square
    is fused from 8192 points on when the covariance depends on beta: one workgroup owns a whole point there and must not
    touch counters[point] (round-2 advisor finding: an out-of-bounds atomic for point >= 16384).  Same results as the
    two-launch path, twice in a row (nothing may be left behind in foreign memory), and as the cells kernel."""
    fit = boss_fit["config"]
    n = 20000
    rows = fit._fit_rows(cases.halton_params(n, with_beta=True), fit.model)
    eng = fit._get_engine()
    with knobs(NO_FUSE="1"):
        ref_l, ref_c = fit.log_likelihood_batch(rows)
    bound = chi2_bound(fit, rows)
    with knobs(MAPPING="point"):
        for rep in range(2):
            lnl, chi = fit.log_likelihood_batch(rows)
            assert eng.last_kernel() == "vk_theory_fast_kernel" and eng.last_fused()
            assert_same_chi2(chi, ref_c, bound, what=f"point-major fused 20000, rep {rep}")
            assert_same_lnl(lnl, ref_l, bound, what=f"point-major fused 20000, rep {rep}")        # absolute: lnL passes through zero
    lnl, chi = fit.log_likelihood_batch(rows[:700])            # counters still zero: a split launch right after
    assert_same_chi2(chi, ref_c[:700], bound[:700], what="split launch after the 20000")


def test_nan_beta_reports_a_failed_row_in_every_chi_square_kernel(synth_fit):
    """beta entering only through the covariance bracket (fixed data vector, fixed real-space ccf; reachable through the C ABI,
    not through the reference's option files): a NaN beta must give (-inf, inf) from the fused tail, the workgroup-per-point
    kernel and the wave-per-point kernel alike (round-2 advisor finding: the count-based bracket took it for "below the grid";
    the reference raises IndexError for it, ccf_fit.py:213-228)."""
    import victor_amd
    model, data = cases.synth_options(3)
    fit = victor_amd.CCFFit(model, data)
    base = np.array(fit.covmat, dtype=float)
    fit.fixed_covmat = False                                   # plain attributes, read by engine.build_tables
    fit.beta_covmat = np.array([0.3, 0.4, 0.5])
    fit.covmat = np.stack([base, 1.1 * base, 1.2 * base])
    fit.icov = np.linalg.inv(fit.covmat)
    rows = fit._fit_rows(cases.halton_params(6), fit.model)
    rows[:, _native.P_BETA] = [0.35, np.nan, 0.45, 0.3, np.nan, 0.6]
    for kn in ({}, {"NO_FUSE": "1"}, {"NO_FUSE": "1", "LIKE_WIDE": "0"}, {"MAPPING": "cells"}, {"MAPPING": "point"}):
        with knobs(NO_GRAPH="1", **kn):
            lnl, chi = fit.log_likelihood_batch(rows)
        assert np.all(np.isneginf(lnl[[1, 4]])) and np.all(np.isposinf(chi[[1, 4]])), kn
        assert np.all(np.isfinite(lnl[[0, 2, 3, 5]])), kn
    ref = synth_fit[3].log_likelihood_batch(rows)              # fixed covariance = slice 0: the row ON the first grid value
    assert_same_chi2(chi[3], ref[1][3], n_data=120, what="row on the first covariance slice")


@pytest.mark.gpu
def test_quadratic_form_for_other_sizes_of_the_data_vector(tmp_path, oracle):
    """The fused tail and the wide K2 take chi2 on the precision matrix folded onto its upper triangle and stored by circular
    diagonals, the residual twice over in LDS (vk_kernel_like.h): N = 120 and 60 are the shipped sizes; here N = 135 (odd: a
    zero row and column make it even; more than 64 entry pairs per row: two chunks of lanes), N = 150 (even, two chunks) and
    N = 21 (odd, fewer rows than waves x rows in flight: the zeros behind the doubled residual are read) and N = 270 (more
    entries than threads: the data vector no longer sits in registers, three chunks of lanes, several batches of rows per
    wave) against the oracle, through the single-point call, a fused small batch and a large batch (tiled K2)."""
    import victor_amd
    rng = np.random.default_rng(7)
    for n_s, poles in ((45, 3), (50, 3), (7, 3), (90, 3)):
        N = n_s * poles
        s = np.linspace(3.0, 115.0, n_s)
        data = {"s": s}
        for l, name in zip(range(poles), ("monopole", "quadrupole", "hexadecapole")):
            data[name] = 0.05 * rng.standard_normal(n_s) / (1 + l)
        A = rng.standard_normal((N, N)) * 0.02
        cov = (np.diag(rng.uniform(0.5, 2.0, N)) + A @ A.T) * 1e-4
        cov[0, 1] += 3e-7                                   # a slightly unsymmetric precision matrix, as np.linalg.inv returns them
        np.save(tmp_path / f"data_{N}.npy", data, allow_pickle=True)
        np.save(tmp_path / f"cov_{N}.npy", {"covmat": cov}, allow_pickle=True)
        model, dopt = cases.synth_options(3)
        dopt = cases.clone(dopt)
        dopt["dir"] = str(tmp_path)
        dopt["redshift_space_ccf"]["data_file"] = f"data_{N}.npy"
        dopt["covariance_matrix"]["data_file"] = f"cov_{N}.npy"
        fit = victor_amd.CCFFit(model, dopt)
        ofit = oracle.OracleFit(model, dopt)
        hp = cases.halton_params(3000)
        want = np.array([ofit.log_likelihood(cases.point(hp, i)) for i in range(6)])
        for i in range(6):                                  # one point per call: point-major kernel, fused tail
            lnl, chi2 = fit.log_likelihood(cases.point(hp, i))
            assert abs(chi2 / want[i, 1] - 1) < RTOL and abs(lnl / want[i, 0] - 1) < RTOL, (N, i)
        lnl40, chi40 = fit.log_likelihood_batch({k: v[:40] for k, v in hp.items()})      # cells kernel, fused tail
        assert np.max(np.abs(chi40[:6] / want[:, 1] - 1)) < RTOL, N
        _native.set_knob("VICTOR_HIP_NO_FUSE", "1")                                      # the same 40 points through the wide K2
        try:
            _, chi_w = fit.log_likelihood_batch({k: v[:40] for k, v in hp.items()})
        finally:
            _native.set_knob("VICTOR_HIP_NO_FUSE", None)
        bound = chi2_bound(fit, {k: v[:40] for k, v in hp.items()})
        assert_same_chi2(chi_w, chi40, bound, what=f"wide K2 vs fused tail, N={N}")
        _, chi_big = fit.log_likelihood_batch(hp)                                         # 3000 points: K2 on the full matrix
        assert_same_chi2(chi_big[:40], chi40, bound, what=f"tiled K2 vs fused tail, N={N}")


@pytest.mark.gpu
def test_single_point_shortcut_does_not_change_the_result(synth_fit, boss_fit):
    """One point per call takes a shortcut of its own: the parameter row travels in the kernel arguments instead of the pinned
    host buffer.  It may not change a bit of the result (same kernel, same row); and the batch path (cells kernel from 20
    points on, smaller batches through the point-major kernel) agrees with the single-point calls to rounding."""
    for fit, beta in ((synth_fit[3], False), (boss_fit["config"], True)):
        hp = cases.halton_params(12, with_beta=beta)
        for i in range(12):
            p = cases.point(hp, i)
            want = fit.log_likelihood(p)
            assert fit._get_engine().last_kernel() == "vk_theory_fast_kernel"
            _native.set_knob("VICTOR_HIP_NO_INLINE_ROW", "1")
            try:
                got = fit.log_likelihood(p)
            finally:
                _native.set_knob("VICTOR_HIP_NO_INLINE_ROW", None)
            assert got == want, (i, got, want)
        lnl, chi2 = fit.log_likelihood_batch(hp)
        one = np.array([fit.log_likelihood(cases.point(hp, i)) for i in range(12)])
        bound = chi2_bound(fit, hp)
        assert_same_chi2(chi2, one[:, 1], bound, what="batch of 12 vs single-point calls")
        assert_same_lnl(lnl, one[:, 0], bound, what="batch of 12 vs single-point calls")


@pytest.mark.gpu
def test_polling_handoff_gives_the_bits_of_the_counter_handoff(synth_fit, boss_fit):
    """Launches of a few points whose planes are split over workgroups hand their partial sums over by polling (vk_common.h:
    kPollEmpty) instead of the completion counters: the same sums added in the same order, so not a bit may change - single
    points, the small batches that still split, call after call on one context (the polling area must be left empty every
    time), and interleaved with launches that use the counters."""
    for fit, beta in ((synth_fit[3], False), (boss_fit["config"], True)):
        hp = cases.halton_params(24, with_beta=beta)
        pts = [cases.point(hp, i) for i in range(24)]

        def run():
            out = [fit.log_likelihood(p) for p in pts]                      # one point per call, 24 times over
            for n in (2, 3, 5, 8):                                          # small batches (split planes, several points)
                sub = {k: v[:n] for k, v in hp.items()}
                out.append(tuple(np.concatenate(fit.log_likelihood_batch(sub)).tolist()))
                out.append(fit.log_likelihood(pts[n]))                      # ... and a single point right behind each
            fit.log_likelihood_batch({k: v[:16] for k, v in hp.items()})     # 16 points: counters (too many workgroups to poll)
            out.append(fit.log_likelihood(pts[7]))
            return out

        polled = run()
        eng = fit._get_engine()
        assert eng.last_polled()                                            # the single point behind the 16-point batch
        fit.log_likelihood_batch({k: v[:3] for k, v in hp.items()})
        assert eng.last_polled()                                            # three points still split their planes: the context's reservation grows to cover them
        fit.log_likelihood_batch({k: v[:16] for k, v in hp.items()})
        assert not eng.last_polled()
        _native.set_knob("VICTOR_HIP_NO_POLL", "1")
        try:
            counted = run()
            assert not eng.last_polled()
        finally:
            _native.set_knob("VICTOR_HIP_NO_POLL", None)
        assert polled == counted
        assert all(np.all(np.isfinite(np.asarray(v))) for v in polled)


@pytest.mark.gpu
def test_reference_outputs_over_the_prior_box(boss_fit):
    """The reference itself on 48 Halton points of the cobaya prior box (tests/golden/ref_outputs_box.npz, five parameters
    sampled, BOSS configuration), four RSD models: theory vectors, lnL and chi2 through the single-point call and the batch
    call.  rtol 1e-9 (the contract is 1e-6)."""
    g, meta = cases.golden_outputs("box")
    fit = boss_fit["config"]
    hp = cases.halton_params(meta["n"], with_beta=True)
    for rsd in ("streaming", "dispersion", "kaiser", "euclid_special"):
        lnl_b, chi_b = fit.log_likelihood_batch(hp, rsd_model=rsd)
        assert np.max(np.abs(chi_b / g[f"{rsd}_chi2"] - 1)) <= RTOL, (rsd, float(np.max(np.abs(chi_b / g[f"{rsd}_chi2"] - 1))))
        assert np.max(np.abs(lnl_b - g[f"{rsd}_lnl"]) / np.maximum(np.abs(g[f"{rsd}_lnl"]), 1.0)) <= RTOL, rsd
        for i in range(meta["n"]):
            p = cases.point(hp, i)
            t = fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s, rsd_model=rsd)
            assert vec_close(t, g[f"{rsd}_theory"][i]), (rsd, i)
            if i % 4 == 0:
                lnl, chi2 = fit.log_likelihood(dict(p), rsd_model=rsd)
                assert abs(chi2 - g[f"{rsd}_chi2"][i]) <= RTOL * g[f"{rsd}_chi2"][i], (rsd, i)
                assert abs(lnl - g[f"{rsd}_lnl"][i]) <= RTOL * max(abs(g[f"{rsd}_lnl"][i]), 1.0), (rsd, i)
