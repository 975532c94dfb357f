"""Workload definitions of the benchmark: option dictionaries and parameter sets (bench.py, __graft_entry__.smoke(), examples;
tests/cases.py re-exports them for the tests).

The BOSS dictionaries are equivalent to the reference's ``config/boss_config.yaml`` ('config') and to the
``model``/``data`` blocks of ``config/boss_cobaya_config.yaml`` ('cobaya'), pointed at the converted inputs
under ``tests/golden/boss``; the synthetic ones are SURVEY.md App. E / section 8(d) configs 2 and 3.
"""

import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(ROOT, "tests", "golden")          # the input tables are committed fixtures (data, not code)


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


def golden_outputs(simpson_even="simpson"):
    """Reference outputs: ``'simpson'`` = the reference with SciPy >= 1.11's ``simps`` (the default rule of this repo),
    ``'avg'`` = with SciPy < 1.11's (oracle/make_golden.py --set avg); same inputs.  ``'more'`` = the remaining shipped
    combinations of model, data and covariance files (SHIPPED_COMBINATIONS; default rule)."""
    name = {"simpson": "ref_outputs.npz", "avg": "ref_outputs_avg.npz", "more": "ref_outputs_more.npz",
            "box": "ref_outputs_box.npz"}[simpson_even]       # 'box': 48 Halton points of the cobaya prior box, four RSD models
    g = np.load(os.path.join(GOLDEN, name))
    meta = json.loads(str(g["meta_json"]))
    return g, meta


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


def dsplit_options(q):
    """Quantile q of the 5-quantile joint fit: config-3 options on its own tables."""
    model, data = synth_options(3)
    model["input_model_data_file"] = f"dsplit/model_q{q}.npy"
    data["redshift_space_ccf"]["data_file"] = f"dsplit/data_q{q}.npy"
    return model, data


def cobaya_info():
    import yaml
    with open(os.path.join(ROOT, "config", "boss_cobaya_config.yaml")) as fh:
        return yaml.full_load(fh)
