"""DEV-ONLY: produce ``tests/golden/`` from the unmodified reference.

TEST INFRASTRUCTURE.  Run in the development container (needs
``/root/reference``; see ``oracle/ref_shim.py``)::

    python oracle/make_golden.py

Writes
  tests/golden/boss/{model,data,cov}.npy     BOSS DR12 CMASS inputs, converted from the
                                             reference's HDF5 data files to the ``.npy``
                                             dict format the reference API reads natively
                                             (ccf_model.py:62-63) - data, not source.
  tests/golden/synth/{model,data2,cov2,data3,cov3}.npy
                                             the deterministic synthetic metric-grid inputs
                                             (SURVEY.md App. E); data vectors come from the
                                             reference's own theory at the fiducial point.
  tests/golden/ref_outputs.npz               reference outputs (theory vectors, chi2, lnL,
                                             xi(s,mu)) for every golden case, with ``simps`` standing
                                             for SciPy >= 1.11 (this image's ``simpson``).
  tests/golden/ref_outputs_avg.npz           (``--set avg``) the streaming / dispersion cases again with
                                             ``simps`` standing for SciPy < 1.11 (``even='avg'``), the
                                             convention of the reference's published notebook numbers;
                                             same input files as above.

  tests/golden/ref_outputs_more.npz          (``--set more``) the remaining shipped combinations of model, data and
                                             covariance files (fixed covariance, Patchy-mean data, anisotropic M+D
                                             covariance), with boss/{cov_fixed,cov_md_aniso,patchy_data}.npy

    python oracle/make_golden.py             # all sets
    python oracle/make_golden.py --set avg   # only the second one (inputs and ref_outputs.npz untouched)
    python oracle/make_golden.py --set more  # only the third one
    python oracle/make_golden.py --set disp  # only the fourth one: tests/golden/ref_outputs_disp.npz from
                                             # tests/golden/disp_worst_rows.json (see main_disp)
    python oracle/make_golden.py --set box   # only the fifth one: tests/golden/ref_outputs_box.npz - 48 Halton points of the
                                             # cobaya prior box (five parameters), BOSS, the four RSD models (see main_box)
"""

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402

BOSS_DIR = "data/BOSS_DR12_CMASS_data/"
BOSS_PREFIX = "CMASS_zobovVoids_reconRs10_0.43z0.7_medianRvcut_"


def halton(n, bases=(2, 3, 5, 7), skip=1):
    out = np.empty((n, len(bases)))
    for j, b in enumerate(bases):
        for i in range(n):
            k, f, x = i + skip, 1.0, 0.0
            while k > 0:
                f /= b
                x += f * (k % b)
                k //= b
            out[i, j] = x
    return out


def halton_params(n):
    """prior box of boss_cobaya_config.yaml:51-97."""
    h = halton(n)
    return [dict(fsigma8=0.05 + 1.45 * a, sigma_v=100 + 400 * b, aperp=0.8 + 0.4 * c, apar=0.8 + 0.4 * d)
            for a, b, c, d in h]


def save_dict(path, d):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.save(path, {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in d.items()}, allow_pickle=True)


# --------------------------------------------------------------------------- #
def boss_options(variant="config"):
    """Option dicts equivalent to config/boss_config.yaml (variant 'config') or the
    model/data blocks of config/boss_cobaya_config.yaml (variant 'cobaya'), pointed at
    the converted .npy inputs."""
    model = {
        "input_model_data_file": "boss/model.npy",
        "rsd_model": "streaming",
        "z_eff": 0.57,
        "cosmology": {"Omega_m": 0.31},
        "realspace_ccf": {"reconstruction": True, "beta_key": "beta", "format": "multipoles",
                          "ccf_keys": ["r", "monopole", "quadrupole"], "assume_isotropic": True},
        "matter_ccf": {"model": "template", "integrated": False, "template_keys": ["rdelta", "delta"],
                       "template_sigma8": 0.628, "bias": 1.9},
        "velocity_pdf": {"mean": {"model": "linear", "empirical_corr": False},
                         "dispersion": {"model": "template", "template_keys": ["rsv", "sigmav"]}},
    }
    if variant == "config":
        model["velocity_pdf"]["rescale_templates_independent_of_AP"] = False
    data = {
        "redshift_space_ccf": {"reconstruction": True, "data_file": "boss/data.npy", "format": "multipoles",
                               "ccf_keys": ["s", "monopole", "quadrupole"], "beta_key": None},
        "covariance_matrix": {"data_file": "boss/cov.npy", "cov_key": "covmat", "fixed_beta": False,
                              "beta_key": "beta"},
        "beta_interpolation": "datavector",
        "likelihood": {"form": "sellentin", "nmocks": 1000, "nparams": 4},
    }
    return model, data


def synth_tables(dc=-0.9, rs=38.0):
    """SURVEY.md App. E (deterministic, no RNG); (dc, rs) vary per density-split quantile (section 8d config 5)."""
    r = 1.5 + 3.0 * np.arange(40)

    def h(x, dc=dc, rs=rs, rv=45.0, a=2.2, b=7.5):
        return dc * (1 - (x / rs) ** a) / (1 + (x / rv) ** b)

    rdelta = 1.2 + 2.4 * np.arange(60)
    rsv = 3.0 + 6.0 * np.arange(25)
    model = {
        "r": r,
        "monopole": h(r),
        "quadrupole": 0.02 * (r / 40) ** 2 * np.exp(-(r / 50) ** 2),
        "hexadecapole": 0.005 * (r / 40) ** 4 * np.exp(-(r / 45) ** 2),
        "rdelta": rdelta,
        "delta": h(rdelta) / 2,
        "rsv": rsv,
        "sigmav": 385 * (1 - 0.45 * np.exp(-(rsv / 25) ** 2)),
    }
    return model


def synth_cov(s, n_ell):
    sig = 0.004 * (1 + 20 / s)
    idx = np.arange(len(s))
    T = np.outer(sig, sig) * 0.6 ** np.abs(idx[:, None] - idx[None, :])
    K = np.array([[1, .2, .05], [.2, 1, .2], [.05, .2, 1]])[:n_ell, :n_ell]
    return np.kron(K, T)


def synth_options(config):
    """config 2: isotropic xi_r, data l=0,2, template rescale independent of AP (default).
    config 3: xi_r l=0,2,4, data l=0,2,4, AP-dependent rescale."""
    aniso = config == 3
    model = {
        "input_model_data_file": "synth/model.npy",
        "rsd_model": "streaming",
        "z_eff": 0.57,
        "cosmology": {"Omega_m": 0.31},
        "realspace_ccf": {"reconstruction": False, "format": "multipoles",
                          "ccf_keys": ["r", "monopole", "quadrupole", "hexadecapole"] if aniso else ["r", "monopole"],
                          "assume_isotropic": not aniso},
        "matter_ccf": {"model": "template", "integrated": False, "template_keys": ["rdelta", "delta"],
                       "template_sigma8": 0.628},
        "velocity_pdf": {"mean": {"model": "linear"},
                         "dispersion": {"model": "template", "template_keys": ["rsv", "sigmav"]}},
    }
    if aniso:
        model["velocity_pdf"]["rescale_templates_independent_of_AP"] = False
    data = {
        "redshift_space_ccf": {"reconstruction": False, "data_file": f"synth/data{config}.npy",
                               "format": "multipoles",
                               "ccf_keys": ["s", "monopole", "quadrupole", "hexadecapole"] if aniso
                               else ["s", "monopole", "quadrupole"]},
        "covariance_matrix": {"data_file": f"synth/cov{config}.npy", "cov_key": "covmat"},
        "likelihood": {"form": "gaussian"},
    }
    return model, data


# --------------------------------------------------------------------------- #
def main_avg():
    """Reference outputs under the SciPy < 1.11 ``simps`` (even='avg'), on the inputs ``main`` wrote."""
    ref_shim.set_simpson_rule("avg")
    v = ref_shim.load()
    meta = json.loads(str(np.load(os.path.join(GOLD, "ref_outputs.npz"))["meta_json"]))
    boss_points, hp = meta["boss_points"], meta["synth_points"]
    out = {}
    model, data = boss_options("config")
    model["dir"] = data["dir"] = GOLD
    fit = v.CCFFit(model, data)
    # the converted inputs reproduce the reference's own config + HDF5 files under this rule too
    info = ref_shim.boss_config()
    fit0 = v.CCFFit(info["model"], info["data"])
    assert fit0.log_likelihood(dict(boss_points[0])) == fit.log_likelihood(dict(boss_points[0]))
    ll = [fit.log_likelihood(dict(p)) for p in boss_points]
    out["boss_config_theory"] = np.array([fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s) for p in boss_points])
    out["boss_config_lnl"] = np.array([a for a, b in ll])
    out["boss_config_chi2"] = np.array([b for a, b in ll])
    p = boss_points[0]
    nb = {"streaming": {}, "dispersion": dict(rsd_model="dispersion"), "kaiser": dict(rsd_model="kaiser"),
          "anisotropic": dict(assume_isotropic=False), "beta_likelihood": dict(beta_interpolation="likelihood")}
    for k, kw in nb.items():
        l, c = fit.log_likelihood(dict(p), **kw)
        out[f"boss_nb_{k}"] = np.array([c, l])
    out["boss_dispersion_theory"] = np.array(
        [fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, rsd_model="dispersion") for q in boss_points[:3]])
    out["boss_aniso_theory"] = np.array(
        [fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, assume_isotropic=False) for q in boss_points[:3]])
    out["boss_config_xi_smu_p0"] = fit.theory_xi(*np.meshgrid(fit.s, np.linspace(0, 1, 100)), dict(p))
    for config in (2, 3):
        model, data = synth_options(config)
        model["dir"] = data["dir"] = GOLD
        fit = v.CCFFit(model, data)
        pts = list(hp)
        if config == 3:
            pts = [{"fsigma8": 0.47, "sigma_v": 380, "aperp": 1.02, "apar": 0.97}] + pts
        pts = pts[:9]
        ll = [fit.log_likelihood(dict(q)) for q in pts]
        out[f"synth{config}_theory"] = np.array([fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s) for q in pts])
        out[f"synth{config}_lnl"] = np.array([a for a, b in ll])
        out[f"synth{config}_chi2"] = np.array([b for a, b in ll])
    out["meta_json"] = np.array(json.dumps({"simps": "scipy<1.11 even='avg'", "boss_points": boss_points,
                                            "synth_points": hp[:9]}))
    np.savez_compressed(os.path.join(GOLD, "ref_outputs_avg.npz"), **out)
    ref_shim.set_simpson_rule("simpson")
    for k in sorted(out):
        if k.startswith("boss_nb"):
            print("avg", k, np.array2string(out[k], precision=12))


# the shipped combinations of model / data / covariance files (data/BOSS_DR12_CMASS_data/README.txt) beyond the two that
# ``main`` covers; tests/cases.py::shipped_combination builds the same option dictionaries
MORE_CASES = {
    # CMASS data + PatchyMean model + the fixed (beta-independent) data-only covariance: beta-dependent data vector
    # against ONE precision matrix, no log-det term
    "fixedcov": dict(cov="boss/cov_fixed.npy", fixed_beta=True),
    # mean of the 1000 Patchy mocks as the data vector, with its covariance: the shipped file is the CMASS stack times
    # 1e-6 to within an ulp, so the test side rebuilds it from cov.npy (`cov_scale`) - chi2 of order 1e7
    "patchy": dict(data="boss/patchy_data.npy", cov="boss/cov.npy", cov_scale=1e-6),
    # measured real-space ccf with its ANISOTROPIC (M+D) covariance and the anisotropic sum
    "fromdata_aniso": dict(model="boss/measured_model.npy", from_data=True, cov="boss/cov_md_aniso.npy",
                           kwargs=dict(assume_isotropic=False)),
    # Patchy mean data + PatchyMean template + isotropic (M+D) covariance of the mean (= the CMASS one / 1000 exactly)
    "patchy_md": dict(data="boss/patchy_data.npy", cov="boss/cov_md_iso.npy", cov_scale=1e-3),
}


def more_options(case, gold, scratch):
    """(model, data, kwargs) of one MORE_CASES entry; a rescaled covariance stack is written to ``scratch``."""
    c = MORE_CASES[case]
    model, data = boss_options("config")
    model["dir"] = data["dir"] = gold
    if "model" in c:
        model["input_model_data_file"] = c["model"]
    if c.get("from_data"):
        model["realspace_ccf"]["from_data"] = True
    if "data" in c:
        data["redshift_space_ccf"]["data_file"] = c["data"]
    cov = c["cov"]
    if "cov_scale" in c:
        d = np.load(os.path.join(gold, cov), allow_pickle=True).item()
        d = dict(d, covmat=d["covmat"] * c["cov_scale"])
        cov = os.path.join(scratch, f"cov_{case}.npy")
        save_dict(cov, d)
    data["covariance_matrix"]["data_file"] = cov
    if c.get("fixed_beta"):
        data["covariance_matrix"]["fixed_beta"] = True
    return model, data, dict(c.get("kwargs", {}))


def main_more():
    """Third fixture set, ``tests/golden/ref_outputs_more.npz``: every remaining shipped combination of model, data and
    covariance files, run through the unmodified reference (default ``simps`` rule).  Leaves the first two sets untouched."""
    import tempfile
    ref_shim.set_simpson_rule("simpson")
    v = ref_shim.load()
    meta = json.loads(str(np.load(os.path.join(GOLD, "ref_outputs.npz"))["meta_json"]))
    boss_points = meta["boss_points"]
    src = os.path.join(ref_shim.REFERENCE_ROOT, BOSS_DIR, BOSS_PREFIX)
    save_dict(os.path.join(GOLD, "boss", "cov_fixed.npy"), ref_shim._h5_read(src + "fixed_D_covariance.hdf5"))
    save_dict(os.path.join(GOLD, "boss", "cov_md_aniso.npy"),
              ref_shim._h5_read(src + "variable_anisotropic_MD_covariance.hdf5"))
    patchy = src.replace("CMASS_zobovVoids", "PatchyMean_zobovVoids")
    save_dict(os.path.join(GOLD, "boss", "patchy_data.npy"), ref_shim._h5_read(patchy + "data.hdf5"))
    # how close the rebuilt Patchy covariances are to the shipped files
    for name, fac, ref_file in (("cov.npy", 1e-6, "variable_D_covariance.hdf5"),
                                ("cov_md_iso.npy", 1e-3, "variable_isotropic_MD_covariance.hdf5")):
        ours = np.load(os.path.join(GOLD, "boss", name), allow_pickle=True).item()["covmat"] * fac
        theirs = ref_shim._h5_read(patchy + ref_file)["covmat"]
        assert np.max(np.abs(ours / theirs - 1)) < 5e-16, name
    out = {}
    with tempfile.TemporaryDirectory() as scratch:
        for case in MORE_CASES:
            model, data, kw = more_options(case, GOLD, scratch)
            fit = v.CCFFit(model, data)
            out[f"{case}_theory"] = np.array([fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, **kw)
                                              for q in boss_points])
            for form in ("sellentin", "gaussian"):
                ll = [fit.log_likelihood(dict(q), likelihood={"form": form, "nmocks": 1000, "nparams": 4}, **kw)
                      for q in boss_points]
                out[f"{case}_{form}_lnl"] = np.array([a for a, b in ll])
                out[f"{case}_{form}_chi2"] = np.array([b for a, b in ll])
    out["meta_json"] = np.array(json.dumps({"boss_points": boss_points, "cases": sorted(MORE_CASES)}))
    np.savez_compressed(os.path.join(GOLD, "ref_outputs_more.npz"), **out)
    for k in sorted(out):
        if k.endswith(("chi2", "lnl")):
            print("more", k, np.array2string(out[k], precision=10))


def ulp_step(x, direction):
    """x moved by one unit in the last place (towards +inf for direction > 0, -inf otherwise)."""
    return float(np.nextafter(x, np.inf if direction > 0 else -np.inf))


def main_disp():
    """Fourth fixture set, ``tests/golden/ref_outputs_disp.npz``: the dispersion model (ccf_model.py:658-671) where it is
    ill-conditioned.  ``tests/golden/disp_worst_rows.json`` lists parameter rows, found on the GPU by
    tools/gpu_find_disp_rows.py as the rows on which the three GPU mappings disagree most (a velocity node next to r = 0 at
    mu = 1: the five fixed-point iterations amplify rounding).  The unmodified reference is run on each row AND on copies
    with one input moved by one ulp in either direction, so the fixture records the reference's own spread next to its
    outputs: ``<case>_theory`` (n, N), ``<case>_chi2``, ``<case>_lnl``, ``<case>_spread_theory`` = max over the perturbed
    copies of max|d xi_l| / max|xi_l|, ``<case>_spread_chi2`` = max relative change of chi2, ``<case>_rows`` = the inputs.
    A row on which the reference itself returns NaN (one integrand point where the iteration collapses to r = 0 and the
    Jacobian 1 + q + mu_r^2 (dq - q) cancels to exactly 0 makes xi(s, mu) infinite there, and the global bicubic fit of
    utils.py:45-56 spreads it over every multipole bin; chi2 = inf) is kept as such, spreads 0."""
    ref_shim.set_simpson_rule("simpson")
    v = ref_shim.load()
    with open(os.path.join(GOLD, "disp_worst_rows.json")) as fh:
        worst = json.load(fh)
    cases = {"synth3": (synth_options(3), {}), "boss": (boss_options("config"), {}),
             "boss_emp": (boss_options("config"), {"empirical_corr": True})}
    out, meta = {}, {}
    for name, rows in worst.items():
        (model, data), kw = cases[name]
        kw = dict(kw, rsd_model="dispersion")
        model["dir"] = data["dir"] = GOLD
        fit = v.CCFFit(model, data)
        keys = sorted(rows[0]["params"])
        th, chi, lnl, sp_t, sp_c, inputs = [], [], [], [], [], []
        for row in rows:
            p = {k: float.fromhex(x) for k, x in row["params"].items()}
            t0 = fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s, **kw)
            l0, c0 = fit.log_likelihood(dict(p), **kw)
            dt, dc = 0.0, 0.0
            for k in keys:
                for direction in (+1, -1):
                    q = dict(p)
                    q[k] = ulp_step(p[k], direction)
                    t1 = fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, **kw)
                    _, c1 = fit.log_likelihood(dict(q), **kw)
                    if np.all(np.isfinite(t0)) and np.all(np.isfinite(t1)):
                        dt = max(dt, float(np.max(np.abs(t1 - t0)) / np.max(np.abs(t0))))
                        dc = max(dc, abs(c1 / c0 - 1))
                    elif np.all(np.isfinite(t0)) != np.all(np.isfinite(t1)):
                        dt = dc = np.inf                 # finite on one side of a 1-ulp move only
            th.append(t0); chi.append(c0); lnl.append(l0); sp_t.append(dt); sp_c.append(dc)
            inputs.append([p[k] for k in keys])
            print(f"disp {name}: gpu mapping spread {row['gpu_mapping_spread']:.1e}  reference spread under 1-ulp input moves: "
                  f"theory {dt:.1e}, chi2 {dc:.1e}  (chi2 {c0:.6f})", flush=True)
        out[f"{name}_theory"], out[f"{name}_chi2"], out[f"{name}_lnl"] = np.array(th), np.array(chi), np.array(lnl)
        out[f"{name}_spread_theory"], out[f"{name}_spread_chi2"] = np.array(sp_t), np.array(sp_c)
        out[f"{name}_rows"] = np.array(inputs)
        meta[name] = {"keys": keys, "kwargs": kw, "gpu_mapping_spread": [r["gpu_mapping_spread"] for r in rows]}
    out["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLD, "ref_outputs_disp.npz"), **out)


def halton5(n):
    """tests/cases.py: halton_params(n, with_beta=True) - the prior box of boss_cobaya_config.yaml:51-97 with beta."""
    h = halton(n, bases=(2, 3, 5, 7, 11))
    return [dict(fsigma8=0.05 + 1.45 * a, sigma_v=100 + 400 * b, aperp=0.8 + 0.4 * c, apar=0.8 + 0.4 * d, beta=0.2 + 0.4 * e)
            for a, b, c, d, e in h]


def main_box():
    """The reference over the whole prior box, not a handful of hand-picked points: 48 Halton points (five parameters, beta
    sampled) on the BOSS configuration of config/boss_config.yaml, theory vector + (lnL, chi2) for the streaming, dispersion,
    kaiser and euclid_special models.  ~1 minute of the reference."""
    ref_shim.set_simpson_rule("simpson")
    v = ref_shim.load()
    model, data = boss_options("config")
    model["dir"] = data["dir"] = GOLD
    fit = v.CCFFit(model, data)
    pts = halton5(48)
    out = {"meta_json": np.array(json.dumps({"n": len(pts), "generator": "halton bases 2,3,5,7,11, skip 1 (tests/cases.py: "
                                                                          "halton_params(48, with_beta=True))"}))}
    import warnings
    for rsd in ("streaming", "dispersion", "kaiser", "euclid_special"):
        th, lnl, chi = [], [], []
        for p in pts:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                th.append(fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s, rsd_model=rsd))
                l, c = fit.log_likelihood(dict(p), rsd_model=rsd)
            lnl.append(l)
            chi.append(c)
        out[f"{rsd}_theory"] = np.array(th)
        out[f"{rsd}_lnl"] = np.array(lnl)
        out[f"{rsd}_chi2"] = np.array(chi)
        print(rsd, "chi2 range", np.nanmin(chi), np.nanmax(chi), "non-finite rows", int(np.sum(~np.isfinite(chi))))
    np.savez_compressed(os.path.join(GOLD, "ref_outputs_box.npz"), **out)


def main():
    ref_shim.set_simpson_rule("simpson")
    v = ref_shim.load()
    out = {}
    meta = {}

    # ---- BOSS inputs: HDF5 -> npy dict -------------------------------------
    src = os.path.join(ref_shim.REFERENCE_ROOT, BOSS_DIR, BOSS_PREFIX)
    save_dict(os.path.join(GOLD, "boss", "model.npy"), ref_shim._h5_read(src + "PatchyMean_model.hdf5"))
    save_dict(os.path.join(GOLD, "boss", "data.npy"), ref_shim._h5_read(src + "data.hdf5"))
    save_dict(os.path.join(GOLD, "boss", "cov.npy"), ref_shim._h5_read(src + "variable_D_covariance.hdf5"))

    beta_grid = ref_shim._h5_read(src + "data.hdf5")["beta"]
    boss_points = [
        {"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0},
        {"fsigma8": 0.55, "beta": 0.43, "sigma_v": 320, "epsilon": 1.04},
        {"fsigma8": 0.30, "beta": 0.251, "sigma_v": 450, "aperp": 0.97, "apar": 1.05},
        {"fsigma8": 0.62, "beta": float(beta_grid[12]), "sigma_v": 290, "epsilon": 0.97},   # exact grid match
        {"fsigma8": 0.41, "beta": 0.150, "sigma_v": 410, "epsilon": 1.01},                  # below grid
        {"fsigma8": 0.52, "beta": 0.660, "sigma_v": 365, "epsilon": 0.99, "alpha": 1.02},   # above grid
        {"fsigma8": 0.35, "beta": 0.58, "sigma_v": 150, "aperp": 1.1, "apar": 0.85},
        {"fsigma8": 1.20, "beta": 0.22, "sigma_v": 495, "aperp": 0.82, "apar": 1.18},
    ]
    meta["boss_points"] = boss_points

    for variant in ("config", "cobaya"):
        model, data = boss_options(variant)
        model["dir"] = data["dir"] = GOLD
        fit = v.CCFFit(model, data)
        if variant == "config":
            # sanity: the converted inputs reproduce the reference's own config + HDF5 files
            info = ref_shim.boss_config()
            fit0 = v.CCFFit(info["model"], info["data"])
            a = fit0.log_likelihood(dict(boss_points[0]))
            b = fit.log_likelihood(dict(boss_points[0]))
            assert a == b, (a, b)
            out["boss_iaH"] = np.array(fit.iaH)
            out["boss_sv_rmu"] = fit.sv_rmu
            r_ext = np.append([0.01], fit.r)
            out["boss_delta_ext"] = fit.delta(r_ext)
            out["boss_int_delta_ext"] = fit.integrated_delta(r_ext)
            out["boss_icov_12"] = fit.icov[12]
            out["boss_icov_30"] = fit.icov[30]
        th, chi, lnl = [], [], []
        for p in boss_points:
            th.append(fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s))
            l, c = fit.log_likelihood(dict(p))
            chi.append(c)
            lnl.append(l)
        out[f"boss_{variant}_theory"] = np.array(th)
        out[f"boss_{variant}_chi2"] = np.array(chi)
        out[f"boss_{variant}_lnl"] = np.array(lnl)
        if variant == "config":
            mu = np.linspace(0, 1, 100)
            out["boss_config_xi_smu_p0"] = fit.theory_xi(*np.meshgrid(fit.s, mu), dict(boss_points[0]))
            out["boss_config_xi_smu_p2"] = fit.theory_xi(*np.meshgrid(fit.s, mu), dict(boss_points[2]))
            # the five notebook variants (victor_usage_demo.ipynb:505-518)
            p = boss_points[0]
            nb = {}
            nb["streaming"] = fit.log_likelihood(dict(p))
            nb["dispersion"] = fit.log_likelihood(dict(p), rsd_model="dispersion")
            nb["kaiser"] = fit.log_likelihood(dict(p), rsd_model="kaiser")
            nb["anisotropic"] = fit.log_likelihood(dict(p), assume_isotropic=False)
            nb["beta_likelihood"] = fit.log_likelihood(dict(p), beta_interpolation="likelihood")
            for k, (l, c) in nb.items():
                out[f"boss_nb_{k}"] = np.array([c, l])
            # likelihood forms
            for form in ("gaussian", "hartlap", "percival", "sellentin"):
                l, c = fit.log_likelihood(dict(p), likelihood={"form": form, "nmocks": 1000, "nparams": 4})
                out[f"boss_form_{form}"] = np.array([c, l])
            # other rsd branches, theory vectors at two points (f1 rows)
            for rsd in ("dispersion", "kaiser", "euclid_special"):
                out[f"boss_{rsd}_theory"] = np.array(
                    [fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, rsd_model=rsd) for q in boss_points[:3]])
            out["boss_kaiser_approx_theory"] = np.array(
                [fit.theory_multipole_vector(fit.s, dict(q, M=1.1, Q=0.9), fit.poles_s, rsd_model="kaiser",
                                             kaiser_approximation=True) for q in boss_points[:3]])
            out["boss_aniso_theory"] = np.array(
                [fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, assume_isotropic=False)
                 for q in boss_points[:3]])

    # ---- synthetic metric-grid inputs ---------------------------------------
    tables = synth_tables()
    save_dict(os.path.join(GOLD, "synth", "model.npy"), tables)
    s = tables["r"].copy()
    fid = {"fsigma8": 0.45, "sigma_v": 360, "aperp": 1.0, "apar": 1.0}
    hp = halton_params(24)
    meta["synth_points"] = hp
    for config, n_ell in ((2, 2), (3, 3)):
        model, data = synth_options(config)
        model["dir"] = data["dir"] = GOLD
        cm = v.CCFModel(model)
        poles = [0, 2, 4][:n_ell]
        t_fid = cm.theory_multipole_vector(s, dict(fid), poles)
        dvec = t_fid * (1 + 0.01 * np.sin(np.arange(len(t_fid))))
        d = {"s": s}
        for i, name in enumerate(["monopole", "quadrupole", "hexadecapole"][:n_ell]):
            d[name] = dvec[i * len(s):(i + 1) * len(s)]
        save_dict(os.path.join(GOLD, "synth", f"data{config}.npy"), d)
        save_dict(os.path.join(GOLD, "synth", f"cov{config}.npy"), {"covmat": synth_cov(s, n_ell)})
        fit = v.CCFFit(model, data)
        pts = list(hp)
        if config == 3:
            pts = [{"fsigma8": 0.47, "sigma_v": 380, "aperp": 1.02, "apar": 0.97}] + pts
        th, chi, lnl = [], [], []
        for p in pts:
            th.append(fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s))
            l, c = fit.log_likelihood(dict(p))
            chi.append(c)
            lnl.append(l)
        out[f"synth{config}_theory"] = np.array(th)
        out[f"synth{config}_chi2"] = np.array(chi)
        out[f"synth{config}_lnl"] = np.array(lnl)
        out[f"synth{config}_xi_smu_p0"] = fit.theory_xi(*np.meshgrid(fit.s, np.linspace(0, 1, 100)), dict(pts[0]))

    # ---- remaining model options: linear_bias, empirical_corr, from_data + MD covariance ----
    save_dict(os.path.join(GOLD, "boss", "measured_model.npy"), ref_shim._h5_read(src + "measured_model.hdf5"))
    save_dict(os.path.join(GOLD, "boss", "cov_md_iso.npy"),
              ref_shim._h5_read(src + "variable_isotropic_MD_covariance.hdf5"))
    model, data = boss_options("config")
    model["dir"] = data["dir"] = GOLD
    fit = v.CCFFit(model, data)
    pts3 = [dict(q, bias=2.1, Av=0.7, M=1.05, Q=0.95) for q in boss_points[:3]]
    opt_cases = {
        "lb_stream": dict(matter_model="linear_bias"),
        "lb_kaiser": dict(matter_model="linear_bias", rsd_model="kaiser"),
        "lb_disp": dict(matter_model="linear_bias", rsd_model="dispersion"),
        "emp_stream": dict(empirical_corr=True),
        "emp_disp": dict(empirical_corr=True, rsd_model="dispersion"),
        "emp_kaiser": dict(empirical_corr=True, rsd_model="kaiser"),
    }
    boss_extra = dict(lb_emp_stream=dict(matter_model="linear_bias", empirical_corr=True),
                      lb_emp_disp=dict(matter_model="linear_bias", empirical_corr=True, rsd_model="dispersion"),
                      lb_emp_kaiser=dict(matter_model="linear_bias", empirical_corr=True, rsd_model="kaiser"))
    for tag, kw in dict(opt_cases, **boss_extra).items():
        out[f"opt_boss_{tag}"] = np.array([fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, **kw) for q in pts3])
    # fixed real-space input (synthetic tables): linear_bias with and without the empirical correction
    model, data = synth_options(3)
    model["dir"] = data["dir"] = GOLD
    fit = v.CCFFit(model, data)
    spts = [dict(q, beta=0.4, bias=1.7, Av=-0.5, M=1.1, Q=0.9) for q in hp[:3]]   # beta: required key, value unused
    for tag, kw in dict(opt_cases, lb_emp_stream=dict(matter_model="linear_bias", empirical_corr=True),
                        lb_emp_disp=dict(matter_model="linear_bias", empirical_corr=True, rsd_model="dispersion")).items():
        out[f"opt_synth_{tag}"] = np.array([fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, **kw) for q in spts])
    # real-space ccf measured from the data itself + the matching (M+D) covariance on its own 15-node beta grid
    model, data = boss_options("config")
    model["dir"] = data["dir"] = GOLD
    model["input_model_data_file"] = "boss/measured_model.npy"
    model["realspace_ccf"]["from_data"] = True
    data["covariance_matrix"]["data_file"] = "boss/cov_md_iso.npy"
    fit = v.CCFFit(model, data)
    fd_pts = [dict(q) for q in boss_points]
    out["opt_fromdata_theory"] = np.array([fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s) for q in fd_pts])
    ll = [fit.log_likelihood(dict(q)) for q in fd_pts]
    out["opt_fromdata_lnl"] = np.array([a for a, b in ll])
    out["opt_fromdata_chi2"] = np.array([b for a, b in ll])
    out["opt_fromdata_aniso_theory"] = np.array(
        [fit.theory_multipole_vector(fit.s, dict(q), fit.poles_s, assume_isotropic=False) for q in fd_pts[:3]])
    out["opt_fromdata_lb_kaiser"] = np.array(
        [fit.theory_multipole_vector(fit.s, dict(q, bias=2.0), fit.poles_s, matter_model="linear_bias", rsd_model="kaiser")
         for q in fd_pts[:3]])
    # host-side accessors used in the notebooks
    r_plot = np.linspace(0.01, 120, 100)
    model, data = boss_options("config")
    model["dir"] = data["dir"] = GOLD
    cm = v.CCFModel(model)
    out["acc_delta_lb"] = np.array(cm.delta_profiles(r_plot, {"beta": 0.37}, matter_model="linear_bias"))
    out["acc_delta_tmpl"] = np.array(cm.delta_profiles(r_plot, {"beta": 0.37}))
    out["acc_vterms"] = np.array(cm.velocity_terms(cm.r, {"fsigma8": 0.47, "epsilon": 1.0, "beta": 0.37}))
    out["acc_vterms_emp"] = np.array(cm.velocity_terms(cm.r, {"fsigma8": 0.47, "epsilon": 1.0, "beta": 0.37, "Av": 1},
                                                       empirical_corr=True))

    # ---- density-split style joint fit: 5 table sets sharing one parameter vector ----
    dsplit_pts = hp[:6]
    tot_chi = np.zeros(len(dsplit_pts))
    tot_lnl = np.zeros(len(dsplit_pts))
    for q in range(5):
        tq = synth_tables(dc=-0.9 + 0.4 * q, rs=38.0 + 4.0 * q)
        save_dict(os.path.join(GOLD, "dsplit", f"model_q{q}.npy"), tq)
        model, data = synth_options(3)
        model["input_model_data_file"] = f"dsplit/model_q{q}.npy"
        data["redshift_space_ccf"]["data_file"] = f"dsplit/data_q{q}.npy"
        model["dir"] = data["dir"] = GOLD
        cm = v.CCFModel(model)
        t_fid = cm.theory_multipole_vector(s, dict(fid), [0, 2, 4])
        dvec = t_fid * (1 + 0.01 * np.sin(np.arange(len(t_fid)) + q))
        save_dict(os.path.join(GOLD, "dsplit", f"data_q{q}.npy"),
                  {"s": s, "monopole": dvec[:40], "quadrupole": dvec[40:80], "hexadecapole": dvec[80:]})
        fit = v.CCFFit(model, data)
        for i, p in enumerate(dsplit_pts):
            l, c = fit.log_likelihood(dict(p))
            tot_chi[i] += c
            tot_lnl[i] += l
    out["dsplit_chi2"] = tot_chi
    out["dsplit_lnl"] = tot_lnl

    out["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(GOLD, "ref_outputs.npz"), **out)
    for k in sorted(out):
        if k.endswith(("chi2", "lnl")) or k.startswith(("boss_nb", "boss_form")):
            print(k, np.array2string(out[k], precision=12))


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", choices=["all", "default", "avg", "more", "disp", "box"], default="all")
    which = ap.parse_args().set
    if which in ("all", "default"):
        main()
    if which in ("all", "avg"):
        main_avg()
    if which in ("all", "more"):
        main_more()
    if which in ("all", "disp"):
        main_disp()
    if which in ("all", "box"):
        main_box()
