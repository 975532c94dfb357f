"""Option dictionaries and parameter sets shared by the tests, smoke() and bench.py.

The BOSS dictionaries are equivalent to the reference's ``config/boss_config.yaml`` ('config') and to the
``model``/``data`` blocks of ``config/boss_cobaya_config.yaml`` ('cobaya'), pointed at the converted inputs
under ``tests/golden/boss``; the synthetic ones are SURVEY.md App. E / section 8(d) configs 2 and 3.
"""

import copy
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def boss_options(variant="config"):
    model = {
        "dir": GOLDEN,
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
        "dir": GOLDEN,
        "redshift_space_ccf": {"reconstruction": True, "data_file": "boss/data.npy", "format": "multipoles",
                               "ccf_keys": ["s", "monopole", "quadrupole"], "beta_key": None},
        "covariance_matrix": {"data_file": "boss/cov.npy", "cov_key": "covmat", "fixed_beta": False,
                              "beta_key": "beta"},
        "beta_interpolation": "datavector",
        "likelihood": {"form": "sellentin", "nmocks": 1000, "nparams": 4},
    }
    return model, data


def synth_options(config):
    aniso = config == 3
    model = {
        "dir": GOLDEN,
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
        "dir": GOLDEN,
        "redshift_space_ccf": {"reconstruction": False, "data_file": f"synth/data{config}.npy",
                               "format": "multipoles",
                               "ccf_keys": ["s", "monopole", "quadrupole", "hexadecapole"] if aniso
                               else ["s", "monopole", "quadrupole"]},
        "covariance_matrix": {"data_file": f"synth/cov{config}.npy", "cov_key": "covmat"},
        "likelihood": {"form": "gaussian"},
    }
    return model, data


# The shipped combinations of model / data / covariance files (the reference's data/BOSS_DR12_CMASS_data/README.txt) beyond
# boss_options() and the measured model + isotropic (M+D) covariance; same table as oracle/make_golden.py::MORE_CASES.
SHIPPED_COMBINATIONS = {
    "fixedcov": dict(cov="boss/cov_fixed.npy", fixed_beta=True),
    "patchy": dict(data="boss/patchy_data.npy", cov="boss/cov.npy", cov_scale=1e-6),
    "fromdata_aniso": dict(model="boss/measured_model.npy", from_data=True, cov="boss/cov_md_aniso.npy",
                           kwargs=dict(assume_isotropic=False)),
    "patchy_md": dict(data="boss/patchy_data.npy", cov="boss/cov_md_iso.npy", cov_scale=1e-3),
}


def shipped_combination(case, scratch):
    """(model, data, call kwargs) of one SHIPPED_COMBINATIONS entry.  The Patchy-mean covariances are the CMASS stacks times
    1e-6 / 1e-3 (to an ulp / exactly): they are rebuilt from the committed stacks into ``scratch`` instead of being stored."""
    c = SHIPPED_COMBINATIONS[case]
    model, data = boss_options("config")
    if "model" in c:
        model["input_model_data_file"] = c["model"]
    if c.get("from_data"):
        model["realspace_ccf"]["from_data"] = True
    if "data" in c:
        data["redshift_space_ccf"]["data_file"] = c["data"]
    cov = c["cov"]
    if "cov_scale" in c:
        d = np.load(os.path.join(GOLDEN, cov), allow_pickle=True).item()
        d = {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in dict(d, covmat=d["covmat"] * c["cov_scale"]).items()}
        cov = os.path.join(str(scratch), f"cov_{case}.npy")
        np.save(cov, d, allow_pickle=True)
    data["covariance_matrix"]["data_file"] = cov
    if c.get("fixed_beta"):
        data["covariance_matrix"]["fixed_beta"] = True
    return model, data, dict(c.get("kwargs", {}))


def golden_outputs(simpson_even="simpson"):
    """Reference outputs: ``'simpson'`` = the reference with SciPy >= 1.11's ``simps`` (the default rule of this repo),
    ``'avg'`` = with SciPy < 1.11's (oracle/make_golden.py --set avg); same inputs.  ``'more'`` = the remaining shipped
    combinations of model, data and covariance files (SHIPPED_COMBINATIONS; default rule)."""
    name = {"simpson": "ref_outputs.npz", "avg": "ref_outputs_avg.npz", "more": "ref_outputs_more.npz",
            "box": "ref_outputs_box.npz"}[simpson_even]       # 'box': 48 Halton points of the cobaya prior box, four RSD models
    g = np.load(os.path.join(GOLDEN, name))
    meta = json.loads(str(g["meta_json"]))
    return g, meta


# (chi2, lnL) printed by the reference's notebook, notebooks/victor_usage_demo.ipynb:491-499, and the call options
NOTEBOOK_POINT = {"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0}
NOTEBOOK_PRINTED = {
    "streaming": ((65.01, 284.76), {}),
    "dispersion": ((65.03, 284.76), {"rsd_model": "dispersion"}),
    "kaiser": ((103.90, 266.81), {"rsd_model": "kaiser"}),
    "anisotropic": ((64.39, 285.06), {"assume_isotropic": False}),
    "beta_likelihood": ((64.80, 285.30), {"beta_interpolation": "likelihood"}),
}


def halton(n, bases=(2, 3, 5, 7), skip=1):
    """Deterministic low-discrepancy points in the unit cube (no RNG)."""
    idx = np.arange(skip, skip + n)
    out = np.empty((n, len(bases)))
    for j, b in enumerate(bases):
        k = idx.copy()
        f = 1.0
        x = np.zeros(n)
        while np.any(k > 0):
            f /= b
            x += f * (k % b)
            k //= b
        out[:, j] = x
    return out


def halton_params(n, with_beta=False):
    """Prior box of the reference's cobaya run (boss_cobaya_config.yaml:51-97)."""
    h = halton(n, bases=(2, 3, 5, 7, 11) if with_beta else (2, 3, 5, 7))
    p = {"fsigma8": 0.05 + 1.45 * h[:, 0], "sigma_v": 100 + 400 * h[:, 1],
         "aperp": 0.8 + 0.4 * h[:, 2], "apar": 0.8 + 0.4 * h[:, 3]}
    if with_beta:
        p["beta"] = 0.2 + 0.4 * h[:, 4]
    return p


def point(pdict, i):
    return {k: float(np.atleast_1d(v)[i] if np.ndim(v) else v) for k, v in pdict.items()}


def clone(d):
    return copy.deepcopy(d)


def dsplit_options(q):
    """Quantile q of the 5-quantile joint fit: config-3 options on its own tables."""
    model, data = synth_options(3)
    model["input_model_data_file"] = f"dsplit/model_q{q}.npy"
    data["redshift_space_ccf"]["data_file"] = f"dsplit/data_q{q}.npy"
    return model, data


def cobaya_info():
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "config", "boss_cobaya_config.yaml")) as fh:
        return yaml.full_load(fh)


def dispersion_fixture():
    """The reference's outputs on the rows where the dispersion model's fixed-point iteration is ill-conditioned, with the
    reference's own spread under 1-ulp moves of the inputs (oracle/make_golden.py --set disp).  Returns (npz, meta,
    {case: (model, data)})."""
    g = np.load(os.path.join(GOLDEN, "ref_outputs_disp.npz"))
    meta = json.loads(str(g["meta_json"]))
    options = {"synth3": synth_options(3), "boss": boss_options("config"), "boss_emp": boss_options("config")}
    return g, meta, options
