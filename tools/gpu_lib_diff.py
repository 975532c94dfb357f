"""Outputs of two library builds on the same inputs (BOSS and config 3, 4099 wide-box points, 64 and 1 point): max deviation of
the theory vectors relative to max |xi_l| of the row, and of chi2.  Usage: gpu_lib_diff.py libA.so libB.so"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
import victor_amd
from tests import cases
from tools.gpu_fuzz import params
out = {}
for name, opts, beta in (("boss", cases.boss_options("config"), True), ("config3", cases.synth_options(3), False)):
    fit = victor_amd.CCFFit(*opts)
    for n in (4099, 64, 1):
        rows = fit._fit_rows(params(n, beta, 7, 2.0), fit.model)
        out[f"{name}_{n}_t"] = fit.theory_vector_batch(rows)
        out[f"{name}_{n}_c"] = fit.log_likelihood_batch(rows)[1]
        out[f"{name}_{n}_k"] = np.array(fit._get_engine().last_kernel())
np.savez(sys.argv[1], **out)
''' % ROOT
import numpy as np
res = []
for lib in sys.argv[1:3]:
    f = tempfile.mktemp(suffix=".npz")
    r = subprocess.run([sys.executable, "-c", WORKER, f], env=dict(os.environ, VICTOR_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-2000:]); sys.exit(1)
    res.append(np.load(f))
a, b = res
for k in a.files:
    if k.endswith("_t"):
        scale = np.max(np.abs(a[k]), axis=1, keepdims=True)
        ok = np.isfinite(a[k]).all(axis=1) & np.isfinite(b[k]).all(axis=1)
        c = k[:-2] + "_c"
        print(f"{k[:-2]:14s} {str(a[k[:-2] + '_k'])[10:40]:30s} rows {int(ok.sum())}/{len(ok)}: max dxi/max|xi| {np.max(np.abs(a[k][ok] - b[k][ok]) / scale[ok]):.2e}, "
              f"max rel dchi2 {np.nanmax(np.abs(a[c][ok] / b[c][ok] - 1)):.2e}, non-finite rows equal: {bool(np.array_equal(np.isfinite(a[c]), np.isfinite(b[c])))}", flush=True)
