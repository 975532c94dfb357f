"""Host-side logic that needs no GPU: C-ABI surface, table compilation, option handling, readers."""

import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


@pytest.fixture(scope="module")
def lib():
    from victor_amd.build import build_native
    from victor_amd import _native
    build_native()
    return _native.load()


def have_gpu(lib):
    return lib.vk_device_count() > 0


# --------------------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol(lib):
    from victor_amd import _native
    header = open(os.path.join(ROOT, "include", "victor_hip.h")).read()
    declared = set(re.findall(r"\b(vk_[a-z0-9_]+)\s*\(", header))
    declared -= {"vk_ctx"}
    assert declared, "no prototypes found in the header"
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    from victor_amd import _native as N2
    assert lib.vk_abi_version() == N2.VK_ABI_VERSION == int(re.search(r"#define VK_ABI_VERSION (\d+)", header).group(1))


def test_struct_layouts_match_header(lib):
    """ctypes mirrors must have the C sizes (compile a tiny C program against the header)."""
    from victor_amd import _native
    src = ('#include <stdio.h>\n#include "victor_hip.h"\nint main(){printf("%zu %zu %zu %d\\n", sizeof(vk_pp), '
           'sizeof(vk_tables), sizeof(vk_eval_opts), VK_NPAR);return 0;}\n')
    exe = os.path.join("/tmp", f"vk_sizes_{os.getpid()}")
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    out = subprocess.check_output([exe]).decode().split()
    os.remove(exe)
    assert int(out[0]) == ctypes.sizeof(_native.vk_pp)
    assert int(out[1]) == ctypes.sizeof(_native.vk_tables)
    assert int(out[2]) == ctypes.sizeof(_native.vk_eval_opts)
    assert int(out[3]) == _native.VK_NPAR


def test_no_cpu_fallback_without_gpu(lib):
    if have_gpu(lib):
        pytest.skip("a GPU is visible here")
    import victor_amd
    from victor_amd._native import NativeError
    fit = victor_amd.CCFFit(*cases.synth_options(2))
    with pytest.raises(NativeError):
        fit.log_likelihood({"fsigma8": 0.45, "sigma_v": 360})
    with pytest.raises(NativeError):
        fit.theory_multipoles(fit.s, {"fsigma8": 0.45})


def test_vk_create_rejects_bad_tables(lib):
    from victor_amd import _native as N
    err = ctypes.create_string_buffer(256)
    t = N.vk_tables()
    assert not lib.vk_create(ctypes.byref(t), 0, err, len(err))
    assert b"grid sizes" in err.value or b"bad" in err.value
    assert not lib.vk_create(None, 0, err, len(err))


def test_product_never_imports_oracle():
    """The shipped package must not reference anything under oracle/."""
    for base, _, files in os.walk(os.path.join(ROOT, "victor_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(base, f)).read()
                assert "victor_oracle" not in text and "ref_shim" not in text, f
    for f in ("victor/__init__.py", "victor/likelihoods/CCFLikelihood.py"):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()
    # the bench: the oracle is the checker of its `cpu_baseline` leg and nothing else - imported inside the worker that runs in
    # child processes (bench_legs.cpu_worker), never by the measured path
    import ast
    for f in ("bench.py", "bench_pmc.py"):          # (bench.py NAMES the oracle in the text of the `cpu_baseline.sample` field)
        assert not re.search(r"^\s*(import|from)\s+victor_oracle", open(os.path.join(ROOT, f)).read(), flags=re.M), f
    tree = ast.parse(open(os.path.join(ROOT, "bench_legs.py")).read())
    users = [n.name for n in tree.body if isinstance(n, ast.FunctionDef) and "victor_oracle" in ast.unparse(n)]
    assert users == ["cpu_worker"], users
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.unparse(n) for n in tree.body)


# --------------------------------------------------------------------------- host set-up parity
@pytest.fixture(scope="module")
def boss_fit():
    import victor_amd
    return victor_amd.CCFFit(*cases.boss_options("config"))


def test_boss_init_matches_reference(boss_fit):
    g, _ = cases.golden_outputs()
    f = boss_fit
    assert f.iaH == float(g["boss_iaH"])
    assert np.max(np.abs(f.sv_rmu - g["boss_sv_rmu"])) < 1e-15
    r_ext = np.append([0.01], f.r)
    assert np.max(np.abs(f.delta(r_ext) - g["boss_delta_ext"])) < 1e-14
    assert np.max(np.abs(f.integrated_delta(r_ext) / g["boss_int_delta_ext"] - 1)) < 1e-12
    assert np.max(np.abs(f.icov[12] - g["boss_icov_12"])) <= 1e-12 * np.max(np.abs(g["boss_icov_12"]))
    assert list(f.poles_s) == [0, 2] and list(f.poles_r) == [0, 2]
    assert f.s.shape == (30,) and f.covmat.shape == (31, 60, 60) and f.icov.shape == (31, 60, 60)
    assert f.fit_options["likelihood"]["form"] == "sellentin"
    assert f.model["velocity_independent_of_AP"] is False and f.model["assume_isotropic"] is True


def test_host_accessors_match_oracle(boss_fit):
    import victor_oracle as vo
    ora = vo.OracleFit(*cases.boss_options("config"))
    for beta in (0.37, 0.1, 0.7, float(ora.beta_covmat[7]), 0.6486):
        assert np.max(np.abs(boss_fit.get_interpolated_covariance(beta) - ora._interp_stack(ora.covmat, beta))) == 0
        assert np.max(np.abs(boss_fit.get_interpolated_precision(beta) - ora._interp_stack(ora.icov, beta))) == 0
        assert np.max(np.abs(boss_fit.multipole_datavector(beta) - ora.data_vector(beta))) < 1e-15
        assert np.max(np.abs(boss_fit.get_interpolated_real_multipoles(beta) - ora.real_multipoles_at(beta))) < 1e-15
    assert boss_fit.diagonal_errors(0.4).shape == (2, 30)
    c = boss_fit.correlation_matrix(0.4)
    assert np.allclose(np.diag(c), 1.0)


def test_notebook_accessors_match_reference(boss_fit):
    """delta_profiles / velocity_terms as called in the reference's notebooks (host-side accessors)."""
    g, _ = cases.golden_outputs()
    r_plot = np.linspace(0.01, 120, 100)
    got = np.array(boss_fit.delta_profiles(r_plot, {"beta": 0.37}, matter_model="linear_bias"))
    assert np.max(np.abs(got - g["acc_delta_lb"])) < 1e-14
    got = np.array(boss_fit.delta_profiles(r_plot, {"beta": 0.37}))
    assert np.max(np.abs(got - g["acc_delta_tmpl"])) < 1e-14
    p = {"fsigma8": 0.47, "epsilon": 1.0, "beta": 0.37}
    assert np.max(np.abs(np.array(boss_fit.velocity_terms(boss_fit.r, p)) / g["acc_vterms"] - 1)) < 1e-12
    got = np.array(boss_fit.velocity_terms(boss_fit.r, dict(p, Av=1), empirical_corr=True))
    assert np.max(np.abs(got / g["acc_vterms_emp"] - 1)) < 1e-12


def test_compiled_tables(boss_fit):
    from victor_amd.engine import build_tables
    t, keep = build_tables(boss_fit, boss_fit)
    assert (t.n_s, t.n_mu, t.n_x, t.n_ell, t.n_ell_r) == (30, 100, 50, 2, 2)
    assert t.n_beta_r == t.n_beta_d == t.n_beta_c == 31
    assert t.xi.n_int == 29 and t.xi.lead == 0 and t.xi.inv_h == pytest.approx(0.25)
    assert t.vr.n_int == 30 and t.vr.lead == 1 and t.vr.inv_h == pytest.approx(0.25)
    assert t.sv.n_int == 24 and t.sv.inv_h == pytest.approx(1 / 6)
    # the generalised-eigenvalue identity used for log det of the blended covariance
    n = 60
    logdet = np.ctypeslib.as_array(t.logdet, shape=(31,))
    eig = np.ctypeslib.as_array(t.eig, shape=(31, n))
    for k, tt in ((12, 0.04537), (3, 0.5), (29, 0.9)):
        blend = (1 - tt) * boss_fit.covmat[k] + tt * boss_fit.covmat[-1]
        want = np.linalg.slogdet(blend)[1]
        got = logdet[k] + np.sum(np.log(1 - tt + tt * eig[k]))
        assert abs(got - want) < 1e-9
    ft, _ = build_tables(__import__("victor_amd").CCFFit(*cases.synth_options(3)), None)
    assert ft.n_beta_r == 0 and ft.n_ell_r == 3 and not ft.data and ft.vr_beta_dep == 0 and ft.matter_model == 0
    lb, _ = build_tables(boss_fit, None, "linear_bias")
    assert lb.vr_beta_dep == 1 and lb.matter_model == 1 and lb.vr.n_int == 30


def test_param_rows(boss_fit):
    from victor_amd import _native as N
    rows = boss_fit._fit_rows({"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.04, "alpha": 1.02},
                              boss_fit.model)
    apar = 1.02 * 1.04 ** (-2 / 3)
    assert rows.shape == (1, N.VK_NPAR)
    assert rows[0, N.P_APAR] == apar and rows[0, N.P_APERP] == 1.04 * apar and rows[0, N.P_EPSILON] == 1.04
    assert rows[0, N.P_BETA] == 0.37 and rows[0, N.P_ASTAR] == 1 and rows[0, N.P_M] == 1 and rows[0, N.P_Q] == 1
    rows = boss_fit._fit_rows({"fsigma8": [0.4, 0.5], "beta": 0.3, "aperp": [0.9, 1.1], "apar": 1.05}, boss_fit.model)
    assert rows.shape == (2, N.VK_NPAR) and np.all(rows[:, N.P_SIGMAV] == 380)
    assert np.allclose(rows[:, N.P_EPSILON], np.array([0.9, 1.1]) / 1.05)
    with pytest.raises(KeyError):
        boss_fit._fit_rows({"beta": 0.3}, boss_fit.model)
    from victor_amd import InputError
    with pytest.raises(InputError):
        boss_fit._fit_rows({"fsigma8": 0.4}, boss_fit.model)        # beta needed for beta-dependent data
    # a dict of arrays and the same points passed one by one give identical rows (both branches of _param_rows)
    hp = cases.halton_params(7, with_beta=True)
    hp["epsilon"] = hp.pop("aperp") / hp.pop("apar")
    hp["alpha"] = 1.01                                              # scalar next to arrays: broadcast
    batch = boss_fit._fit_rows(hp, boss_fit.model)
    single = np.concatenate([boss_fit._fit_rows(cases.point(hp, i), boss_fit.model) for i in range(7)])
    assert batch.shape == single.shape == (7, N.VK_NPAR)
    assert np.max(np.abs(batch - single)) < 4e-16 * np.max(np.abs(single[:, [N.P_APERP, N.P_APAR]]))   # array pow vs scalar pow: last ulp at most
    assert np.allclose(batch, single, rtol=4e-16, atol=0)
    cols = [N.P_FSIGMA8, N.P_SIGMAV, N.P_EPSILON, N.P_BETA, N.P_ASTAR, N.P_M, N.P_Q, N.P_BIAS, N.P_AV]
    assert np.array_equal(batch[:, cols], single[:, cols])
    assert boss_fit._fit_rows({"fsigma8": np.zeros(0), "beta": np.zeros(0)}, boss_fit.model).shape == (0, N.VK_NPAR)
    with pytest.raises(InputError):
        boss_fit._fit_rows({"fsigma8": np.zeros((2, 2)), "beta": 0.3}, boss_fit.model)
    with pytest.raises(InputError):
        boss_fit._fit_rows({"fsigma8": np.zeros(2), "beta": np.zeros(3)}, boss_fit.model)


def test_scalar_row_fast_path(boss_fit):
    """The single-point fast path of ``log_likelihood`` builds its row with ``_scalar_row``: same 12 numbers as the general
    ``_param_rows`` (bit for bit: Python float arithmetic in both), same KeyError / default behaviour as the reference
    (ccf_model.py:583-613, 638)."""
    from victor_amd import _native as N
    for p in ({"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.04, "alpha": 1.02},
              {"fsigma8": 0.3, "beta": 0.251, "aperp": 0.97, "apar": 1.05, "astar": 0.99, "M": 1.1, "Q": 0.9, "bias": 2.2, "Av": -0.3},
              {"fsigma8": np.float64(0.5), "beta": np.float32(0.25)}):
        row = boss_fit._scalar_row(p, True, True)
        assert len(row) == N.VK_NPAR and all(type(x) is float for x in row)
        assert np.array_equal(np.array(row), boss_fit._param_rows(p, True, True)[0])
    assert boss_fit._scalar_row({"fsigma8": 0.4}, False, True)[N.P_BETA] == 0.40            # dummy beta (ccf_model.py:583-587)
    assert boss_fit._scalar_row({"beta": 0.3}, True, False)[N.P_FSIGMA8] == 0.0
    with pytest.raises(KeyError):
        boss_fit._scalar_row({"beta": 0.3}, True, True)
    with pytest.raises(KeyError):
        boss_fit._scalar_row({"fsigma8": 0.4}, True, True)


def test_simpson_even_option(boss_fit):
    """The even-N Simpson convention of the velocity integral (ccf_model.py:690; SciPy < 1.11 vs >= 1.11) is an
    explicit option: ``model['numerics']['simpson_even']`` at construction, ``simpson_even=`` per call."""
    import victor_amd
    from victor_amd import InputError
    from victor_amd import tables as T
    from victor_amd.engine import build_tables
    assert boss_fit.model["simpson_even"] == "simpson"
    scale = (12.0 / 49) / np.sqrt(2 * np.pi)
    for rule, canon in (("avg", "avg"), ("scipy<1.11", "avg"), ("SciPy >= 1.11", "simpson"), (None, "simpson")):
        t, keep = build_tables(boss_fit, boss_fit, simpson_even=rule)
        w = np.ctypeslib.as_array(t.w_x, shape=(50,))
        assert np.allclose(w, T.simpson_weights(50, canon) * scale, rtol=1e-15, atol=0)
    model, data = cases.boss_options("config")
    model["numerics"] = {"simpson_even": "scipy<1.11"}
    assert victor_amd.CCFModel(model).model["simpson_even"] == "avg"
    model["numerics"] = {"simpson_even": "middle"}
    with pytest.raises(InputError):
        victor_amd.CCFModel(model)
    with pytest.raises(InputError):
        build_tables(boss_fit, boss_fit, simpson_even="middle")


def test_input_errors(tmp_path):
    import victor_amd
    from victor_amd import InputError
    model, data = cases.boss_options("config")
    bad = cases.clone(model)
    bad["input_model_data_file"] = "nope.npy"
    with pytest.raises(InputError):
        victor_amd.CCFModel(bad)
    bad = cases.clone(model)
    bad["realspace_ccf"]["ccf_keys"] = ["r"]
    with pytest.raises(InputError):
        victor_amd.CCFModel(bad)
    bad = cases.clone(model)
    bad["realspace_ccf"]["beta_key"] = None
    with pytest.raises(InputError):
        victor_amd.CCFModel(bad)
    bad = cases.clone(model)
    bad["matter_ccf"]["template_sigma8"] = None
    with pytest.raises(InputError):
        victor_amd.CCFModel(bad)
    bad = cases.clone(model)
    bad["velocity_pdf"]["dispersion"] = {"model": "constant"}      # crashes in the reference too
    with pytest.raises(InputError):
        victor_amd.CCFModel(bad)
    bad = cases.clone(model)
    bad["velocity_pdf"]["dispersion"]["model"] = "weird"
    with pytest.raises(InputError):
        victor_amd.CCFModel(bad)
    badd = cases.clone(data)
    badd["covariance_matrix"]["cov_key"] = "missing"
    with pytest.raises(InputError):
        victor_amd.CCFFit(model, badd)
    badd = cases.clone(data)
    badd["covariance_matrix"]["fixed_beta"] = True                 # stack given where a single matrix is expected
    with pytest.raises(InputError):
        victor_amd.CCFFit(model, badd)
    badd = cases.clone(data)
    badd["redshift_space_ccf"]["data_file"] = "nope.npy"
    with pytest.raises(InputError):
        victor_amd.CCFFit(model, badd)
    m = victor_amd.CCFModel(model)
    with pytest.raises(InputError):
        m._check_supported(dict(m.model, rsd_model="bogus"))
    with pytest.raises(InputError):
        m._check_supported(dict(m.model, matter_model="excursion_set"))
    m._check_supported(dict(m.model, matter_model="linear_bias", empirical_corr=True))   # beta-dependent input
    m._check_supported(dict(m.model, matter_model="linear_bias"))


# --------------------------------------------------------------------------- readers
def test_h5lite_reads_reference_style_file():
    from victor_amd import h5lite
    d = h5lite.read_all(os.path.join(cases.GOLDEN, "h5", "void_model_example.h5"))
    assert sorted(d) == ["delta", "monopole", "r", "rdelta", "rsv", "sigmav"]
    assert all(v.shape == (25,) and v.dtype == np.float64 for v in d.values())
    assert np.all(np.diff(d["r"]) > 0) and np.all(np.diff(d["rsv"]) > 0)
    assert d["monopole"].min() < -0.5 and abs(d["monopole"][-1]) < 0.1     # a void profile
    with pytest.raises(h5lite.H5LiteError):
        h5lite._Reader(b"not an hdf5 file" * 100)


def test_model_from_hdf5_like_example_config():
    """config/example_model_input.yaml of the reference, pointed at the HDF5 fixture."""
    import victor_amd
    model = {"dir": cases.GOLDEN, "input_model_data_file": "h5/void_model_example.h5", "rsd_model": "streaming",
             "z_eff": 0.50, "cosmology": {"Omega_m": 0.31},
             "realspace_ccf": {"reconstruction": False, "format": "multipoles", "ccf_keys": ["r", "monopole"]},
             "matter_ccf": {"model": "template", "integrated": False, "template_keys": ["rdelta", "delta"],
                            "template_sigma8": 0.628, "bias": 1.9},
             "velocity_pdf": {"mean": {"model": "linear"},
                              "dispersion": {"model": "template", "template_keys": ["rsv", "sigmav"]}}}
    m = victor_amd.CCFModel(model)
    import victor_oracle as vo
    from victor_amd import h5lite
    o = vo.OracleModel(model, h5lite.read_all(os.path.join(cases.GOLDEN, "h5", "void_model_example.h5")))
    assert m.iaH == o.iaH and np.max(np.abs(m.sv_rmu - o.sv_rmu)) < 1e-15
    r_ext = np.append([0.01], m.r)
    assert np.max(np.abs(m.integrated_delta(r_ext) - o.integrated_delta(r_ext))) < 1e-12


@pytest.mark.reference
def test_h5lite_against_all_reference_files():
    ref = "/root/reference/data"
    if not os.path.isdir(ref) or not os.path.isfile("/opt/conda/bin/h5dump"):
        pytest.skip("reference data or h5dump not present")
    import glob
    import tempfile
    from victor_amd import h5lite
    files = glob.glob(os.path.join(ref, "*", "*.hdf5"))
    assert len(files) >= 12
    for fn in files:
        for key, arr in h5lite.read_all(fn).items():
            with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
                subprocess.check_call(["/opt/conda/bin/h5dump", "-d", "/" + key, "-b", "LE", "-o", tmp.name, fn],
                                      stdout=subprocess.DEVNULL)
                assert np.array_equal(np.fromfile(tmp.name, "<f8"), arr.ravel()), (fn, key)


# --------------------------------------------------------------------------- drop-in alias + cobaya plug-in
def test_victor_alias_and_plugin_host_side():
    code = (
        "import sys, os\n"
        f"sys.path.insert(0, {ROOT!r}); os.chdir({ROOT!r})\n"
        "import victor, victor_amd\n"
        "assert victor.CCFFit is victor_amd.CCFFit and victor.CCFModel is victor_amd.CCFModel\n"
        "sys.path.insert(0, './victor/likelihoods/')            # cobaya's python_path mechanism\n"
        "from CCFLikelihood import CCFLikelihood\n"
        "from tests import cases\n"
        "m, d = cases.boss_options('cobaya')\n"
        "lk = CCFLikelihood(dict(model=m, data=d))\n"
        "assert lk.get_can_provide_params() == ['fsigma8'] and lk.ccf.s.shape == (30,)\n"
        "import yaml\n"
        "y = yaml.full_load(open('victor/likelihoods/CCFLikelihood.yaml'))\n"
        "assert y['config_file'] == 'config/boss_config.yaml' and 'chi2_ccf_correct' in y['params']\n"
        "print('ok')\n")
    out = subprocess.check_output([sys.executable, "-c", code]).decode()
    assert out.strip().endswith("ok")


# --------------------------------------------------------------------------- input variants
def _rmu_inputs(tmp_path, with_beta):
    """A tabulated xi(r, mu) real-space input plus several 'simulations' stacked on a leading axis."""
    r = 2.0 + 4.0 * np.arange(30)
    mu = np.linspace(0, 1, 21)
    base = cases.golden_outputs  # noqa: F841  (keeps flake quiet about unused import paths)
    prof = -0.8 * np.exp(-(r / 30.0) ** 2)[:, None] * (1 + 0.3 * (1.5 * mu[None, :] ** 2 - 0.5))
    sims = np.stack([prof * (1 + 0.05 * k) for k in range(3)])                  # (n_sim, n_r, n_mu)
    src = np.load(os.path.join(cases.GOLDEN, "boss", "model.npy"), allow_pickle=True).item()
    d = {"r": r, "mu": mu, "rdelta": src["rdelta"], "delta": src["delta"], "rsv": src["rsv"], "sigmav": src["sigmav"]}
    if with_beta:
        beta = np.linspace(0.2, 0.6, 5)
        d["beta"] = beta
        d["xi_rmu"] = np.stack([sims * (1 + b) for b in beta], axis=1)          # (n_sim, n_beta, n_r, n_mu)
    else:
        d["xi_rmu"] = sims
    fn = tmp_path / ("rmu_beta.npy" if with_beta else "rmu.npy")
    np.save(fn, d, allow_pickle=True)
    model = cases.boss_options("config")[0]
    model["dir"] = str(tmp_path)
    model["input_model_data_file"] = fn.name
    model["realspace_ccf"] = {"reconstruction": with_beta, "beta_key": "beta", "format": "rmu",
                              "ccf_keys": ["r", "mu", "xi_rmu"], "simulation_number": 1, "assume_isotropic": False}
    return model, d


@pytest.mark.parametrize("with_beta", [False, True])
def test_rmu_format_and_simulation_number_match_oracle(tmp_path, with_beta):
    import victor_amd
    import victor_oracle as vo
    model, d = _rmu_inputs(tmp_path, with_beta)
    m = victor_amd.CCFModel(model)
    o = vo.OracleModel(model)
    assert list(m.poles_r) == [0, 2, 4]
    for ell in ("0", "2", "4"):
        assert m.real_multipoles[ell].shape == o.real_multipoles[ell].shape
        assert np.max(np.abs(m.real_multipoles[ell] - o.real_multipoles[ell])) < 1e-14
    # simulation 1 was selected, not 0
    sim0 = dict(model, realspace_ccf=dict(model["realspace_ccf"], simulation_number=0))
    assert not np.allclose(victor_amd.CCFModel(sim0).real_multipoles["0"], m.real_multipoles["0"])
    from victor_amd import InputError
    bad = dict(model, realspace_ccf=dict(model["realspace_ccf"], simulation_number="one"))
    with pytest.raises(InputError):
        victor_amd.CCFModel(bad)


def _aniso_inputs(tmp_path, non_uniform_mu=False):
    """Synthetic model file with a 3-key sigma_v(r, mu) template and a mean-velocity template."""
    src = np.load(os.path.join(cases.GOLDEN, "synth", "model.npy"), allow_pickle=True).item()
    d = dict(src)
    mu = np.linspace(0, 1, 9) if not non_uniform_mu else np.array([0, 0.1, 0.25, 0.45, 0.6, 0.8, 0.93, 1.0])
    rsv = d["rsv"]
    d["musv"] = mu
    d["sigmav2d"] = d["sigmav"][:, None] * (1 + 0.25 * mu[None, :] ** 2 - 0.1 * np.exp(-rsv[:, None] / 40) * mu[None, :])
    rv = np.linspace(0.5, 125, 60)
    d["rv"] = rv
    d["vtemplate"] = -45.0 * (rv / 30) * np.exp(-(rv / 45) ** 2)
    fn = tmp_path / ("aniso_nu.npy" if non_uniform_mu else "aniso.npy")
    np.save(fn, d, allow_pickle=True)
    model, data = cases.synth_options(3)
    model["dir"] = str(tmp_path)
    model["input_model_data_file"] = fn.name
    model["velocity_pdf"]["dispersion"] = {"model": "template", "template_keys": ["rsv", "musv", "sigmav2d"]}
    model["velocity_pdf"]["mean"] = {"model": "linear", "template_fsigma8": 0.45, "z_sim": 0.52,
                                     "template_hubble_ratio": 1.03, "template_keys": ["rv", "vtemplate"]}
    return model, data


def test_anisotropic_dispersion_and_velocity_template_setup(tmp_path):
    import victor_amd
    import victor_oracle as vo
    from victor_amd import tables as T
    import scipy.interpolate as si
    model, data = _aniso_inputs(tmp_path)
    m = victor_amd.CCFModel(model)
    o = vo.OracleModel(model)
    assert m.sv_rmu.shape == o.sv_rmu.shape == (9, 25) and np.max(np.abs(m.sv_rmu - o.sv_rmu)) < 1e-14
    patches = T.bicubic_patches(m.r_for_sv, m.mu_for_sv, m.sv_rmu.T)
    ref = si.RectBivariateSpline(m.r_for_sv, m.mu_for_sv, m.sv_rmu.T)
    rng = np.random.default_rng(2)
    u = rng.uniform(m.r_for_sv[0], m.r_for_sv[-1], 300)
    v = rng.uniform(0, 1, 300)
    i = np.clip(np.searchsorted(m.r_for_sv, u, side="right") - 1, 0, len(m.r_for_sv) - 2)
    j = np.clip(np.searchsorted(m.mu_for_sv, v, side="right") - 1, 0, len(m.mu_for_sv) - 2)
    du, dv = u - m.r_for_sv[i], v - m.mu_for_sv[j]
    val = sum(patches[i, j, p, q] * du ** p * dv ** q for p in range(4) for q in range(4))
    assert np.max(np.abs(val - ref.ev(u, v))) < 1e-12
    # mean model 'template' needs the template at construction time
    mt = cases.clone(model)
    mt["velocity_pdf"]["mean"]["model"] = "template"
    mm = victor_amd.CCFModel(mt)
    assert mm.has_velocity_template and mm.model["mean_model"] == "template"
    from victor_amd import InputError
    with pytest.raises(InputError):
        m._check_supported(dict(m.model, mean_model="template"))       # no template was loaded for m
    p = {"fsigma8": 0.5, "epsilon": 1.03}
    om = vo.OracleModel(mt)
    a = np.array(mm.velocity_terms(mm.r, p))
    b = np.array(om.velocity_terms(om.r, p, om.model))
    assert np.max(np.abs(a / b - 1)) < 1e-12


def test_dispersion_filter_options(tmp_path):
    """velocity_pdf.dispersion: filter / filter_window / filter_order (ccf_model.py:278-283; the Savitzky-Golay pass along r
    before the template is normalised): defaults (3, 1), a wider window with a higher order, and no filter at all give
    three different tables, each equal to the oracle's (the oracle is checked against the reference on the same
    options in tests/test_oracle_vs_reference.py)."""
    import victor_amd
    import victor_oracle as vo
    seen = []
    for opt in ({}, {"filter_window": 7, "filter_order": 3}, {"filter": False}):
        for aniso in (True, False):
            if aniso:
                model, _ = _aniso_inputs(tmp_path)
            else:
                model, _ = cases.boss_options("config")
            model = cases.clone(model)
            model["velocity_pdf"]["dispersion"].update(opt)
            m, o = victor_amd.CCFModel(model), vo.OracleModel(model)
            assert np.max(np.abs(m.sv_rmu / o.sv_rmu - 1)) < 1e-13, (opt, aniso)
            seen.append(m.sv_rmu)
    assert not np.allclose(seen[0], seen[2], rtol=1e-6) and not np.allclose(seen[0], seen[4], rtol=1e-6)


def _gfx950_code_objects(tmp_dir):
    """Extract the gfx950 code objects from libvictor_hip.so - one clang offload bundle per translation unit in .hip_fatbin
    (victor_amd/build.py: UNITS - victor_hip.hip and one unit per family of theory-kernel instantiations) - into files."""
    import re
    import struct
    import subprocess
    from victor_amd.build import OUT
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.isfile(readelf):
        pytest.skip("llvm-readelf not found")
    sec = subprocess.run([readelf, "-S", OUT], capture_output=True, text=True).stdout
    m = re.search(r"\.hip_fatbin\s+PROGBITS\s+[0-9a-f]+\s+([0-9a-f]+)\s+([0-9a-f]+)", sec)
    assert m, "no .hip_fatbin section"
    with open(OUT, "rb") as fh:
        fh.seek(int(m.group(1), 16))
        blob = fh.read(int(m.group(2), 16))
    magic, paths, start = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    assert blob[:24] == magic
    while True:
        at = blob.find(magic, start)
        if at < 0:
            break
        n, pos = struct.unpack_from("<Q", blob, at + 24)[0], at + 32
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, pos)
            ident = blob[pos + 24:pos + 24 + idlen].decode()
            pos += 24 + idlen
            if "gfx950" in ident:
                path = os.path.join(str(tmp_dir), f"victor_gfx950_{len(paths)}.co")
                with open(path, "wb") as fh:
                    fh.write(blob[at + off:at + off + size])
                paths.append(path)
        start = at + len(magic)
    assert paths, "no gfx950 code object in the library"
    return paths


def _disassemble(paths):
    """{kernel name: [instructions]} over the code objects."""
    import re
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.isfile(objdump):
        pytest.skip("llvm-objdump not found")
    funcs, cur = {}, None
    for co in paths:
        asm = subprocess.run([objdump, "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
        for ln in asm.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
            if m:
                cur = funcs.setdefault(m.group(1), [])
            elif cur is not None and ln.strip():
                cur.append(ln.split("//")[0].strip())
    return funcs


def test_no_kernel_of_the_library_spills(lib, tmp_path):
    """The gfx950 code objects inside libvictor_hip.so (one per translation unit), read with llvm-readelf: every kernel's
    private segment is 0 bytes - no register spills, no scratch traffic (8 bytes per thread in the BOSS cells kernel once were
    134 MB of HBM writes per 65536-point launch; the iterative-ilp scheduler of vk_cells_streaming.hip cost 24-32 bytes until the
    cells kernel lost its grid-stride loop) - and the large-batch theory kernels keep the registers of five workgroups per CU."""
    import re
    import subprocess
    from victor_amd.build import UNITS
    paths = _gfx950_code_objects(tmp_path)
    assert len(paths) == len(UNITS) == 7                    # one code object per translation unit with device code
    notes = "".join(subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
                    for co in paths)
    kernels = re.findall(r"\.name:\s+(\S+)[\s\S]*?\.private_segment_fixed_size:\s+(\d+)[\s\S]*?\.vgpr_count:\s+(\d+)", notes)
    assert len(kernels) > 150
    spilling = [k for k, priv, _ in kernels if int(priv) > 0]
    assert not spilling, spilling
    aniso = [k for k, _, _ in kernels if re.search(r"vk_theory_cells_kernelILi[123]ELi[123]ELi[01]ELi0ELi0EEEv", k)]
    assert len(aniso) == 18                                 # the instantiations of vk_cells_streaming.hip are in the library
    funcs = _disassemble(paths)
    assert not any(ins.startswith("scratch_") for k in aniso for ins in funcs[k])
    for k, _, vgpr in kernels:
        if "vk_theory_lanes_kernel" in k or ("vk_theory_cells_kernel" in k and k.endswith("Li0ELi0EEEvNS_10TheoryArgsE")):   # MODE = streaming, SVA = 0
            assert int(vgpr) <= 96, (k, vgpr)          # 512 / 5 workgroups of four waves, in granules of 8
    # no static LDS in the fast theory kernels: their dynamic LDS then starts at address 0, which vkm::exp_gauss relies on
    # when it reads its table (at the start of dynamic LDS) with the byte offset as the address
    static = re.findall(r"\.group_segment_fixed_size:\s+(\d+)[\s\S]*?\.name:\s+(\S+)", notes)
    fast = [(k, int(sz)) for sz, k in static if re.search(r"vk_theory_(lanes|cells|fast)_kernel|vk_image_kernel", k)]
    assert len(fast) > 100 and all(sz == 0 for _, sz in fast), [x for x in fast if x[1]][:5]


def test_cross_workgroup_handoff_is_ordered_in_the_code_object(lib, tmp_path):
    """Partial sums and theory vectors handed from one workgroup to another inside a launch (vk_common.h: point_completed)
    are write-through (sc1) stores published by a device-scope counter increment.  On gfx950 nothing but an explicit
    `s_waitcnt vmcnt(0)` makes a wave wait for its stores before the barrier that precedes the increment (the back-off barrier
    carries no vmcnt wait, a workgroup-scope release fence emits lgkmcnt(0) only), so the shipped ISA is checked: in every
    kernel that contains the counter's returning `global_atomic_add`, walking back from the atomic there is an s_barrier, and
    walking back from that barrier an `s_waitcnt vmcnt(0)` comes before any global store; the finishing workgroup's reads
    of the handed-over data are sc1 loads."""
    import re
    funcs = _disassemble(_gfx950_code_objects(tmp_path))
    checked = 0
    for name, body in funcs.items():
        atomics = [i for i, ins in enumerate(body) if ins.startswith("global_atomic_add")]
        if not atomics:
            continue
        assert "vk_theory_cells_kernel" in name or "vk_theory_fast_kernel" in name, name    # only the fused / split theory kernels
        for i in atomics:
            assert " sc0" in body[i], (name, body[i])            # returning form: the finishing workgroup is told by the value
            j = i
            while j >= 0 and not body[j].startswith("s_barrier"):
                j -= 1
            assert j >= 0, (name, "no s_barrier in front of the counter increment")
            k, drained = j - 1, False
            while k >= 0:
                ins = body[k]
                if ins.startswith("s_waitcnt") and "vmcnt(0)" in ins:
                    drained = True
                    break
                assert not re.match(r"(global|buffer|flat|scratch)_store", ins), (name, "store between the wait and the barrier", ins)
                k -= 1
            assert drained, (name, "no s_waitcnt vmcnt(0) in front of the barrier")
            checked += 1
        stores = [ins for ins in body if ins.startswith("global_store") and " sc1" in ins]
        loads = [ins for ins in body if ins.startswith("global_load") and " sc1" in ins]
        assert stores and loads, (name, len(stores), len(loads))
    assert checked >= 100, checked


def test_background_cosmology_and_multipole_helpers(boss_fit):
    """victor.BackgroundCosmology (E(z) of the path, cosmology.py:27-45) and utils.fn_from_multipoles (utils.py:60-94)."""
    import victor
    g, _ = cases.golden_outputs()
    c = victor.BackgroundCosmology({"Omega_m": 0.31})
    assert (1 + 0.57) / (100 * c.Ez(0.57)) == float(g["boss_iaH"]) == boss_fit.iaH
    assert c.H(0.0) == 67.5 and abs(c.Om(0.0) - 0.31) < 1e-15 and c.OmegaL == 1 - 0.31
    ck = victor.BackgroundCosmology({"Omega_m": 0.3, "Omega_K": 0.05, "H0": 70.0})
    assert abs(ck.Ez(1.0) - np.sqrt(0.3 * 8 + 0.05 * 4 + 0.65)) < 1e-15 and abs(ck.H(0) - 70.0) < 1e-12
    r = np.linspace(1, 10, 10)
    f = victor.utils.fn_from_multipoles(r, [0, 2], np.vstack([r, np.ones_like(r)]))
    assert abs(f(5.0, 1.0)[0] - 6.0) < 1e-12 and f(r, np.linspace(-1, 1, 7)).shape == (7, 10)
    with pytest.raises(ValueError):
        victor.utils.fn_from_multipoles(r, [0, 2], np.ones((3, 10)))


def test_table_dump_covers_every_field_and_c_client_compiles(boss_fit, tmp_path):
    """victor_amd.engine.dump_tables writes one record per scalar / array of vk_tables (names = C field paths), and the
    plain-C client that reads it compiles against the header."""
    import struct
    from victor_amd.engine import build_tables, dump_tables, table_array_lengths
    t, keep = build_tables(boss_fit, boss_fit)
    path = tmp_path / "tables.bin"
    dump_tables(t, str(path))
    raw = path.read_bytes()
    assert raw[:8] == b"VKTB1\0\0\0"
    pos, names, sizes = 8, [], {}
    while True:
        name = raw[pos:pos + 24].rstrip(b"\0").decode()
        kind, count = struct.unpack_from("<iq", raw, pos + 24)
        pos += 36
        if name == "END":
            break
        nbytes = 8 if kind < 2 else ((count * (8 if kind == 2 else 2) + 7) // 8) * 8
        names.append(name)
        sizes[name] = count
        pos += nbytes
    assert pos == len(raw)
    src = open(os.path.join(ROOT, "examples", "c_abi_client.c")).read()
    declared = re.findall(r'\{"([a-z0-9_.A-Z]+)", [0-3], &t\.', src)
    assert sorted(declared) == sorted(names), set(declared) ^ set(names)
    lengths = table_array_lengths(t)
    assert sizes["prec"] == 31 * 60 * 60 and sizes["uni_xi"] == lengths["uni_xi"] > 0 and sizes["vr_emp"] == 0
    exe = str(tmp_path / "c_abi_client")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_client.c"), "-ldl", "-o", exe], check=True)
    done = subprocess.run([exe], capture_output=True, text=True)
    assert done.returncode == 2 and "usage" in done.stderr


_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_torch_in_the_product_or_the_bench():
    """The north star's "no PyTorch": nothing in the package, the bench or the examples imports torch - the multi-process
    rendezvous is victor_amd/rendezvous.py (standard library), the multi-GPU data path libvictor_hip.so + RCCL."""
    import glob
    import re
    files = glob.glob(os.path.join(_ROOT, "victor_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(_ROOT, "victor", "**", "*.py"), recursive=True) + \
        glob.glob(os.path.join(_ROOT, "examples", "*.py")) + [os.path.join(_ROOT, "bench.py"), os.path.join(_ROOT, "bench_legs.py"), os.path.join(_ROOT, "bench_pmc.py"), os.path.join(_ROOT, "__graft_entry__.py")]
    for f in files:
        assert not re.search(r"^\s*(import torch|from torch)", open(f).read(), flags=re.M), f


def test_plugin_stand_in_has_cobayas_shape():
    """Without cobaya the plug-in's base class is a stand-in with cobaya's constructor signature and defaults mechanism
    (class defaults from CCFLikelihood.yaml, overridden by the run's info); initialize=False builds nothing, so this runs
    without a GPU.  The GPU side is tests/test_gpu_workloads.py::test_cobaya_shaped_construction_and_full_parameter_set."""
    import inspect
    import yaml
    sys.path.insert(0, os.path.join(_ROOT, "victor", "likelihoods"))
    from CCFLikelihood import CCFLikelihood
    sig = list(inspect.signature(CCFLikelihood.__init__).parameters)
    assert sig[:7] == ["self", "info", "name", "timing", "packages_path", "initialize", "standalone"]
    lk = CCFLikelihood({"model": {"x": 1}, "config_file": "elsewhere.yaml"}, "like", None, None, False, False)
    assert lk.get_name() == "like" and lk.model == {"x": 1} and lk.data is None and lk.config_file == "elsewhere.yaml"
    with open(os.path.join(_ROOT, "victor", "likelihoods", "CCFLikelihood.yaml")) as fh:
        defaults = yaml.full_load(fh)
    assert lk.params == defaults["params"] and lk.output_params == ["chi2_ccf_correct"]
    assert {"fsigma8", "beta", "epsilon", "aperp", "apar", "alpha"} <= set(lk.input_params)
    assert lk.get_can_provide_params() == ["fsigma8"]


def test_quadratic_form_on_circular_diagonals_is_the_quadratic_form():
    """The layout the fused tail reads (vk_kernel_like.h, built in vk_create): the precision matrix folded onto its upper
    triangle, T_ii = P_ii, T_ij = P_ij + P_ji, stored by circular diagonals D_k[i] = T[i][(i + k) mod M] for k = 0 .. M/2 (the
    last one for i < M/2 only), M = N rounded up to even.  Restated here in NumPy: every unordered pair appears exactly once
    and sum_i r_i sum_k D_k[i] r_((i + k) mod M) is r.P.r - for even and odd N and an unsymmetric P."""
    rng = np.random.default_rng(11)
    for N in (60, 21, 135):
        P = rng.standard_normal((N, N))
        r = rng.standard_normal(N)
        M = (N + 1) & ~1
        Pm = np.zeros((M, M))
        Pm[:N, :N] = P
        rm = np.zeros(M)
        rm[:N] = r
        seen = np.zeros((M, M), dtype=int)
        chi = 0.0
        for k in range(M // 2 + 1):
            n_i = M // 2 if k == M // 2 else M
            for i in range(n_i):
                j = (i + k) % M
                d = Pm[i, i] if i == j else Pm[i, j] + Pm[j, i]
                seen[min(i, j), max(i, j)] += 1
                chi += d * rm[i] * rm[j]
        assert np.array_equal(seen, np.triu(np.ones((M, M), dtype=int)))
        assert abs(chi - r @ P @ r) <= 1e-12 * np.sum(np.abs(np.outer(r, r) * P))


def test_rccl_hooks_need_the_development_switch(lib, tmp_path, monkeypatch):
    """The two RCCL test hooks are development switches like every other VICTOR_HIP_* knob: an inherited
    VICTOR_HIP_RCCL_SHARED_DEVICE_OK=1 without VICTOR_HIP_DEV=1 leaves ``one_device_per_rank`` strict, and an inherited
    VICTOR_HIP_RCCL_LIB does not swap the collective library (``vk_comm_info`` still names the ROCm install's RCCL)."""
    from victor_amd.sharding import one_device_per_rank

    class Group:
        def allgather_bytes(self, mine, tag):
            return [mine, mine]                     # two ranks on the same GPU

    class Two:
        world, rank, group = 2, 0, Group()

    class Eng:
        def bus_id(self):
            return "0000:05:00.0"

    monkeypatch.delenv("VICTOR_HIP_DEV", raising=False)
    monkeypatch.setenv("VICTOR_HIP_RCCL_SHARED_DEVICE_OK", "1")
    assert one_device_per_rank(Two(), Eng()) is False
    monkeypatch.setenv("VICTOR_HIP_DEV", "1")
    assert one_device_per_rank(Two(), Eng()) is True

    # the library side, in child processes (the choice of library is made once per process)
    from tests.test_gpu_rccl_double import build_double
    double = build_double(tmp_path)
    code = ("import json, sys; sys.path.insert(0, %r); from victor_amd import _native; print(json.dumps(_native.comm_info()))" % _ROOT)
    env = {k: v for k, v in os.environ.items() if not k.startswith("VICTOR_HIP_")}
    env["VICTOR_HIP_RCCL_LIB"] = double
    import json
    plain = json.loads(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    assert plain["rccl_loaded"] and "rccl_double" not in plain["rccl"] and "rccl_double" not in plain["rccl_opened_as"]
    assert os.path.basename(plain["rccl"]).startswith("librccl.so") and plain["rccl_next_to_hip_runtime"] is True
    env["VICTOR_HIP_DEV"] = "1"
    dev = json.loads(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    assert dev["rccl"].endswith("librccl_double.so")


def test_polling_handoff_launch_rule(lib):
    """The rule that keeps waiting workgroups from ever filling an XCD (include/victor_hip.h: vk_poll_rule / vk_poll_grant; DESIGN.md
    section 5), as the pure host functions the launcher itself calls: a launch polls only with >= 2 workgroups per CU (64 slots per
    XCD), resident at once, at most 8 points, all of them covered by its context's reservation; a process's contexts reserve at
    most 32 waiters between them, all or nothing per request; one full-budget process per device (plus single-point contexts
    of others) stays below the 64 slots."""
    import ctypes as C
    per_process, slots = C.c_int32(), C.c_int32()
    full = lib.vk_poll_budget(C.byref(per_process), C.byref(slots))
    assert (per_process.value, slots.value, full) == (32, 64, 1)
    assert per_process.value * full + 31 < slots.value              # the owner's budget + 31 single waiters: below an XCD's slots
    rule = lib.vk_poll_rule
    assert rule(1, 2, 80, 3, 256, 1) == 1                            # one point per call: 40 s bins x 2 parts
    assert rule(8, 2, 480, 2, 256, 8) == 1                           # eight BOSS requests of the mailbox server: 8 x 30 x 2
    assert rule(8, 2, 480, 2, 256, 7) == 0                           # ... not covered by the reservation
    assert rule(9, 2, 540, 3, 256, 9) == 0                           # more than eight waiters per launch: never
    assert rule(1, 1, 40, 3, 256, 8) == 0                            # nothing to hand over
    assert rule(8, 2, 640, 2, 256, 8) == 0                           # 640 workgroups do not fit on 256 CUs x 2 at once
    assert rule(1, 2, 80, 1, 256, 8) == 0                            # tables so large that one workgroup fills a CU: 32 slots per XCD
    assert rule(1, 2, 80, 3, 64, 8) == 0                             # a device whose XCDs offer fewer than 64 slots
    assert rule(0, 2, 0, 3, 256, 8) == 0
    def grant(process, ctx, want, others=0):
        return lib.vk_poll_grant(others, process, ctx, want)
    assert grant(0, 0, 1) == 1 and grant(0, 0, 8) == 8 and grant(0, 1, 8) == 7
    assert grant(0, 8, 4) == 0                                       # already covered
    assert grant(24, 0, 8) == 8 and grant(25, 0, 8) == 0             # all or nothing at the budget's edge
    assert grant(31, 0, 1) == 1 and grant(32, 0, 1) == 0
    assert grant(0, 0, 100) == 8                                     # a context never reserves more than a launch may hold
    # four contexts of an owner process reserving eight each exhaust the budget; a fifth gets nothing and falls back to counters
    total, ctxs = 0, []
    for _ in range(5):
        g = grant(total, 0, 8)
        total += g
        ctxs.append(g)
    assert ctxs == [8, 8, 8, 8, 0] and total == 32
    # across processes (the ledger in /dev/shm): whatever the others hold counts against the device's 63
    assert grant(0, 0, 8, others=32) == 8 and grant(24, 0, 8, others=32) == 0      # an owner (32) + this process: 31 left
    assert grant(30, 0, 1, others=32) == 1 and grant(31, 0, 1, others=32) == 0
    assert grant(0, 0, 1, others=62) == 1 and grant(0, 0, 1, others=63) == 0       # the 64th waiter on the device: never
    assert grant(0, 0, 8, others=60) == 0 and grant(0, 0, 3, others=60) == 3


def test_algorithmic_flop_counts_of_the_bench():
    """The flop counts behind every roofline fraction of the bench line (bench.py: flops_per_eval, model_flops), pinned to the
    figures DESIGN.md section 5 derives from the reference's expressions with SURVEY 8(d)'s conventions."""
    sys.path.insert(0, _ROOT)
    import bench
    assert bench.flops_per_eval(40, 100, 50, 3, True) == 15_424_000            # the metric grid: 200000 x 77 + 2 x 3 x 4000
    assert bench.flops_per_eval(30, 100, 50, 2, False) == 7_062_000            # BOSS: 150000 x 47 + 2 x 2 x 3000
    assert bench.model_flops("streaming", 30, 100, 50, 2) == 7_062_000
    assert bench.model_flops("dispersion", 30, 100, 50, 2) == 150_000 * 156 + 3000 * 20 + 12_000 == 23_472_000
    assert bench.model_flops("dispersion", 30, 100, 50, 2, niter=3) == 150_000 * 120 + 3000 * 20 + 12_000
    assert bench.model_flops("kaiser", 30, 100, 50, 2) == 3000 * 157 + 12_000 == 483_000
    assert bench.model_flops("kaiser", 30, 100, 50, 2, linearised=True) == 3000 * 154 + 12_000
    assert bench.model_flops("kaiser", 30, 100, 50, 2, coord_shift=False) == 3000 * 49 + 12_000
    assert bench.model_flops("euclid_special", 30, 100, 50, 2) == 3000 * 154 + 12_000
    assert bench.model_flops("kaiser", 30, 100, 50, 2, n_data=60) == 483_000 + 2 * 3600 + 180      # + the chi-square of a fused launch
    assert bench.model_flops("dispersion", 40, 100, 50, 3, aniso=True) == 200_000 * 186 + 4000 * 20 + 24_000
    assert bench.at_sustained_clock(0.57, 2.0) == pytest.approx(0.684) and bench.at_sustained_clock(0.57, None) is None


def test_no_gpu_test_asserts_on_elapsed_time_rates_or_ratios():
    """The `-m gpu` suite's verdict is about correctness only: a wall-clock threshold on a shared box is the one kind of assertion
    that can turn the suite red for no defect (and `pytest -x` would then hide every test behind it).  Timings, rates and cost
    ratios are measured by bench.py and tools/, recorded under gpurun_out/ and profiles/, quoted in DESIGN.md - never asserted.
    Every `assert` of tests/test_gpu_*.py is read from the syntax tree; none may mention a timing quantity (the length of a list
    of timings aside)."""
    import ast
    import glob
    import re
    timing = re.compile(r"(_us\b|_us_|\bus_per|_ms\b|_ms_|\bms_per|per_s\b|_per_s_|_ratio\b|\bratio\b|elapsed|perf_counter|monotonic|"
                        r"time\.time|\bseconds\b|_seconds\b|\brate\b|_rate\b|speed|faster|slower|\bdt\b)")
    files = sorted(glob.glob(os.path.join(_ROOT, "tests", "test_gpu_*.py")))
    assert len(files) >= 7
    found = []
    for f in files:
        src = open(f).read()
        for node in ast.walk(ast.parse(src)):
            if isinstance(node, ast.Assert):
                text = ast.get_source_segment(src, node.test) or ""
                text = re.sub(r"len\([^()]*\)", "len()", text)        # how MANY timings a record holds (one per rank) is not a timing
                if timing.search(text):
                    found.append(f"{os.path.basename(f)}:{node.lineno}: {text[:120]}")
    assert not found, "\n".join(found)


def test_tools_are_current():
    """tools/ after the round-6 audit: every script byte-compiles (shell scripts pass `bash -n`), sets only VICTOR_HIP_* names
    that something reads - the library's development knobs (load_knobs), its two RCCL hooks, or the package's own switches -,
    has its line in tools/README.md, and no development knob of the library is left without a user (a tool or a test)."""
    import glob
    import re
    import subprocess
    tools = os.path.join(_ROOT, "tools")
    csrc = os.path.join(_ROOT, "victor_amd", "csrc")
    knobs = set(re.findall(r'getenv\("(VICTOR_HIP_[A-Z_]+)"\)', open(os.path.join(csrc, "victor_hip.hip")).read())) - {"VICTOR_HIP_DEV"}
    assert len(knobs) >= 10 and "VICTOR_HIP_MAPPING" in knobs
    hooks = set(re.findall(r'getenv\("(VICTOR_HIP_[A-Z_]+)"\)', open(os.path.join(csrc, "vk_rccl.cpp")).read()))
    package = set()
    for f in glob.glob(os.path.join(_ROOT, "victor_amd", "*.py")):
        package |= set(re.findall(r'"(VICTOR_HIP_[A-Z_]+)"', open(f).read()))
    known = knobs | hooks | package | {"VICTOR_HIP_DEV"}
    readme = open(os.path.join(tools, "README.md")).read()
    files = sorted(f for f in os.listdir(tools) if f.endswith((".py", ".sh", ".hip")) and not f.startswith("_run"))
    assert 40 <= len(files) <= 60
    used = set()
    for f in files:
        path = os.path.join(tools, f)
        src = open(path).read()
        if f.endswith(".py"):
            compile(src, path, "exec")
        elif f.endswith(".sh"):
            assert subprocess.run(["bash", "-n", path], capture_output=True).returncode == 0, f
        names = set(re.findall(r"VICTOR_HIP_[A-Z_]+[A-Z]", src))
        assert names <= known, (f, names - known)
        used |= names
        assert f"`{f}" in readme, f"{f} has no line in tools/README.md"
    for listed in re.findall(r"^\| `([\w.]+\.(?:py|sh|hip))", readme, flags=re.M):
        assert listed in files, f"tools/README.md lists {listed}, which is not there"
    for name in set(re.findall(r"VICTOR_HIP_[A-Z_]+[A-Z]", "".join(open(f).read() for f in glob.glob(os.path.join(_ROOT, "tests", "test_gpu_*.py"))))):
        assert name in known, f"a GPU test names {name}, which nothing reads"
    # the tests set knobs through helpers that add the prefix (tests/test_gpu_parity.py: knobs(NO_FUSE="1"), {"LIKE_WIDE": "0"})
    tests_src = "".join(open(f).read() for f in glob.glob(os.path.join(_ROOT, "tests", "*.py")))
    def in_tests(knob):
        short = knob[len("VICTOR_HIP_"):]
        return knob in tests_src or re.search(r"\b%s=|\"%s\"\s*:" % (short, short), tests_src) is not None
    idle = {k for k in knobs if k not in used and not in_tests(k)}
    assert not idle, f"development knobs nobody sets: {idle}"
    # ... and every knob a test sets that way is one the library reads
    for short in set(re.findall(r"knobs\(([^)]*)\)", tests_src)) | set():
        for name in re.findall(r"\b([A-Z][A-Z_]+)=", short):
            assert "VICTOR_HIP_" + name in knobs, f"a test sets VICTOR_HIP_{name}, which the library does not read"
    for line in tests_src.splitlines():
        if re.search(r"for kn in \(", line):                         # ... or as dictionaries handed to knobs(**kn)
            for name in re.findall(r"\"([A-Z][A-Z_]+)\":", line):
                assert "VICTOR_HIP_" + name in knobs, f"a test sets VICTOR_HIP_{name}, which the library does not read"


def test_one_routine_forms_the_alcock_paczynski_factors(lib):
    """apar = alpha * eps^(-2/3), aperp = eps * apar (reference: ccf_model.py:589-592, on Python floats): the library's
    vk_epsilon_to_ap gives the bits of that Python expression, and CCFModel._param_rows on ARRAYS goes through it - so a batch
    row, a single-point row and a row of the walkers' native step loop are the same row."""
    import victor_amd
    from victor_amd import _native as N2
    rng = np.random.default_rng(7)
    eps = np.concatenate([rng.uniform(0.5, 1.5, 4000), [1.0, 0.8, 1.2, 1.04, 1e-3, 37.0]])
    for alpha in (1.0, 0.97, 1):
        aperp, apar = N2.epsilon_to_ap(eps, alpha)
        want_apar = np.array([alpha * float(e) ** (-2 / 3) for e in eps])
        assert np.array_equal(apar, want_apar) and np.array_equal(aperp, eps * want_apar)
    model, _ = cases.boss_options("config")
    m = victor_amd.CCFModel(model)
    batch = {"fsigma8": np.full(eps.size, 0.47), "beta": np.full(eps.size, 0.37), "epsilon": eps, "alpha": 0.97}
    rows = m._param_rows(batch, need_beta=True)
    for i in (0, 17, eps.size - 1):
        one = m._param_rows({"fsigma8": 0.47, "beta": 0.37, "epsilon": float(eps[i]), "alpha": 0.97}, need_beta=True)
        assert np.array_equal(rows[i], one[0])


def test_every_kernel_is_generated_in_exactly_one_translation_unit(lib, tmp_path):
    """vk_instances.h says which theory-kernel instantiation lives in which unit; victor_hip.hip (the launch side) declares all
    of them `extern template`.  An instantiation the launch code reaches that is in no list would silently be generated in
    victor_hip.hip again (and the half-minute build would grow back towards its minute and a half): in the shipped library
    every kernel name occurs in exactly one code object, the theory kernels in none that also holds the chi-square kernels'
    unit, and the build's unit lists cover every source file of csrc/."""
    import glob
    import re
    import subprocess
    from victor_amd.build import CSRC, HOST_UNITS, UNITS
    assert sorted(u for u, _, _ in UNITS) == sorted(os.path.basename(f) for f in glob.glob(os.path.join(CSRC, "*.hip")))
    assert sorted(HOST_UNITS) == sorted(os.path.basename(f) for f in glob.glob(os.path.join(CSRC, "*.cpp")))
    where = {}
    for i, co in enumerate(_gfx950_code_objects(tmp_path)):
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        for name in set(re.findall(r"\.name:\s+(\S+)", notes)):
            where.setdefault(name, []).append(i)
    assert len(where) > 200
    twice = {k: v for k, v in where.items() if len(v) != 1}
    assert not twice, twice
    home = {i for k, v in where.items() if "vk_like_tiled_kernel" in k for i in v}          # victor_hip.hip's code object
    assert len(home) == 1
    strays = [k for k, v in where.items() if v[0] in home and re.search(r"vk_theory_(cells|fast)_kernel|vk_theory_kernel|vk_xi_smu_kernel", k)]
    assert not strays, strays
    # the lists themselves: 18 + 45 + 45 cells, 45 + 36 point-major, 36 + 12 generic instantiations
    counts = {fam: sum(1 for k in where if fam in k) for fam in ("vk_theory_cells_kernel", "vk_theory_fast_kernel", "vk_theory_kernelI", "vk_xi_smu_kernel")}
    assert counts == {"vk_theory_cells_kernel": 108, "vk_theory_fast_kernel": 81, "vk_theory_kernelI": 36, "vk_xi_smu_kernel": 12}, counts
