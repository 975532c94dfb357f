"""Option dictionaries and parameter sets of the tests: the benchmark's workloads (``workloads.py`` at the repository root,
re-exported here) plus the test-only cases - the remaining shipped file combinations, the notebook's printed values, the
dispersion-model fixture."""

import copy
import json
import os

import numpy as np

from workloads import (GOLDEN, boss_options, cobaya_info, dsplit_options, golden_outputs, halton, halton_params,  # noqa: F401
                       point, synth_options)


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


# (chi2, lnL) printed by the reference's notebook, notebooks/victor_usage_demo.ipynb:491-499, and the call options
NOTEBOOK_POINT = {"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0}
NOTEBOOK_PRINTED = {
    "streaming": ((65.01, 284.76), {}),
    "dispersion": ((65.03, 284.76), {"rsd_model": "dispersion"}),
    "kaiser": ((103.90, 266.81), {"rsd_model": "kaiser"}),
    "anisotropic": ((64.39, 285.06), {"assume_isotropic": False}),
    "beta_likelihood": ((64.80, 285.30), {"beta_interpolation": "likelihood"}),
}


def clone(d):
    return copy.deepcopy(d)


def dispersion_fixture():
    """The reference's outputs on the rows where the dispersion model's fixed-point iteration is ill-conditioned, with the
    reference's own spread under 1-ulp moves of the inputs (oracle/make_golden.py --set disp).  Returns (npz, meta,
    {case: (model, data)})."""
    g = np.load(os.path.join(GOLDEN, "ref_outputs_disp.npz"))
    meta = json.loads(str(g["meta_json"]))
    options = {"synth3": synth_options(3), "boss": boss_options("config"), "boss_emp": boss_options("config")}
    return g, meta, options
