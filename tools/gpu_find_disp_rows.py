"""Dispersion model (ccf_model.py:658-671): find the parameter rows on which the reference's fixed-point iteration is
ill-conditioned - a velocity node next to r = 0 at mu = 1, where five iterations amplify a 1-ulp difference of any input -
as the rows on which the three GPU mappings (point-major, cells, generic kernel) disagree most.  Writes them, bit-exact (hex
floats), to gpurun_out/disp_worst_rows.json; oracle/make_golden.py --set disp turns that list into a fixture made by the
reference itself (tests/golden/ref_outputs_disp.npz).  Usage: gpu_find_disp_rows.py [rows per case, default 4]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from tests import cases
from victor_amd import _native
from tools.gpu_fuzz import params

keep = int(sys.argv[1]) if len(sys.argv) > 1 else 4
out = {}
for name, opts, beta, kw, src in (
        ("synth3", cases.synth_options(3), False, {}, "halton"),
        ("synth3", cases.synth_options(3), False, {}, "wide"),
        ("boss", cases.boss_options("config"), True, {}, "halton"),
        ("boss", cases.boss_options("config"), True, {}, "wide"),
        ("boss_emp", cases.boss_options("config"), True, {"empirical_corr": True}, "halton")):
    kw = dict(kw, rsd_model="dispersion")
    fit = victor_amd.CCFFit(*opts)
    model = fit._merged(kw)
    n = 131072
    p = cases.halton_params(n, with_beta=beta) if src == "halton" else params(n, beta, 7, 1.0)
    if "empirical_corr" in kw:
        p = dict(p, Av=np.linspace(-1.0, 1.0, n))
    rows = fit._fit_rows(p, model)
    res = {}
    for mapping in ("point", "cells", "generic"):
        env = "VICTOR_HIP_FORCE_GENERIC" if mapping == "generic" else "VICTOR_HIP_MAPPING"
        _native.set_knob(env, "1" if mapping == "generic" else mapping)
        res[mapping] = fit.theory_vector_batch(rows, **kw)
        _native.set_knob(env, None)
    scale = np.max(np.abs(res["generic"]), axis=1)
    dev = np.maximum(np.max(np.abs(res["cells"] - res["generic"]), axis=1), np.max(np.abs(res["point"] - res["generic"]), axis=1)) / scale
    dev[~np.isfinite(dev)] = 0
    order = np.argsort(dev)[::-1][:keep]
    print(f"{name} ({src}): row deviation quantiles 50% {np.median(dev):.1e} 99.9% {np.quantile(dev, 0.999):.1e} max {dev.max():.1e}; "
          f"rows above 1e-9: {int((dev > 1e-9).sum())} of {n}", flush=True)
    lst = out.setdefault(name, [])
    for i in order:
        q = {k: float(np.asarray(v)[i]).hex() for k, v in p.items()}
        lst.append({"params": q, "gpu_mapping_spread": float(dev[i]), "source": src})
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/disp_worst_rows.json", "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote gpurun_out/disp_worst_rows.json")
