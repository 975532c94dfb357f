"""Fuzz of the small- and medium-batch machinery: 40000 points from a prior box much wider than the bench's, evaluated (a) in
one large batch through the two-launch path and (b) in chunks of random size 1 ... 3000 through the default path (plane parts,
cell ranges, completion counters, fused chi-square, in-place host path, captured graphs).  Prints the largest deviations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from tests import cases
from victor_amd import _native
from tools.gpu_fuzz import params

for name, opts, beta, kw in (("config3", cases.synth_options(3), False, {}), ("config2", cases.synth_options(2), False, {}),
                             ("boss", cases.boss_options("config"), True, {}),
                             ("boss anisotropic", cases.boss_options("config"), True, {"assume_isotropic": False}),
                             ("boss dispersion", cases.boss_options("config"), True, {"rsd_model": "dispersion"}),
                             ("boss sellentin->hartlap", cases.boss_options("config"), True,
                              {"likelihood": {"form": "hartlap", "nmocks": 1000}})):
    fit = victor_amd.CCFFit(*opts)
    model = fit._merged(kw)
    n = 40000
    fmax = 1.0 if kw.get("rsd_model") == "dispersion" else 2.0
    rows = fit._fit_rows(params(n, beta, 11, fmax), model)
    _native.set_knob("VICTOR_HIP_NO_FUSE", "1")
    ref_l, ref_c = fit.log_likelihood_batch(rows, **kw)
    _native.set_knob("VICTOR_HIP_NO_FUSE", None)
    rng = np.random.default_rng(3)
    got_l, got_c = np.empty(n), np.empty(n)
    i, sizes = 0, []
    while i < n:
        m = int(min(n - i, rng.choice([1, 2, 3, 5, 8, 13, 23, 24, 31, 64, 100, 127, 128, 255, 256, 500, 511, 512, 513, 1000, 2047, 2048, 2049, 3000])))
        got_l[i:i + m], got_c[i:i + m] = fit.log_likelihood_batch(rows[i:i + m], **kw)
        sizes.append(m)
        i += m
    fin = np.isfinite(ref_l) & np.isfinite(got_l)
    same_fail = int(np.sum(np.isfinite(ref_l) != np.isfinite(got_l)))
    dchi = np.abs(got_c[fin] / ref_c[fin] - 1)
    dl = np.abs(got_l[fin] - ref_l[fin]) / (np.abs(ref_l[fin]) + ref_c[fin] + 1.0)
    print(f"{name}: {len(sizes)} chunks; max rel dchi2 {dchi.max():.2e} (99.99 %: {np.quantile(dchi, 0.9999):.2e}), max scaled dlnL {dl.max():.2e}, "
          f"finite in only one of the two: {same_fail}, failed rows {int((~np.isfinite(ref_l)).sum())}", flush=True)
