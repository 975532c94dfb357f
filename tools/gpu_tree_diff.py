"""Outputs of two CHECKOUTS of this repository (each with its own package and its own built library) on the same inputs, bit for
bit: BOSS and config 3, every RSD model, 4099 wide-box points / 64 / 5 / 1 point (cells, point-major, polling and counter
hand-offs), theory vectors, chi2 and lnL.  For changes that must not move a bit - a rebuild in other translation units, a host-
side refactoring - across an ABI change, which tools/gpu_lib_diff.py (one package, two libraries) cannot cross.
Usage: gpu_tree_diff.py <checkout A> <checkout B>"""
import os
import subprocess
import sys
import tempfile

WORKER = r'''
import sys
root = sys.argv[1]
sys.path.insert(0, root)
import numpy as np
import victor_amd
from tests import cases
from tools.gpu_fuzz import params
out = {}
for name, opts, beta in (("boss", cases.boss_options("config"), True), ("config3", cases.synth_options(3), False)):
    fit = victor_amd.CCFFit(*opts)
    for rsd in ("streaming", "dispersion", "kaiser", "euclid_special"):
        for n in (4099, 64, 5, 1):
            rows = fit._fit_rows(params(n, beta, 7, 2.0), fit.model)
            out[f"{name}_{rsd}_{n}_t"] = fit.theory_vector_batch(rows, rsd_model=rsd)
            lnl, chi = fit.log_likelihood_batch(rows, rsd_model=rsd)[:2]
            out[f"{name}_{rsd}_{n}_c"] = chi
            out[f"{name}_{rsd}_{n}_l"] = lnl
np.savez(sys.argv[2], **out)
'''


def main():
    import numpy as np
    res = []
    for root in sys.argv[1:3]:
        f = tempfile.mktemp(suffix=".npz")
        env = {k: v for k, v in os.environ.items() if not k.startswith("VICTOR_HIP_")}
        r = subprocess.run([sys.executable, "-c", WORKER, os.path.abspath(root), f], env=env, capture_output=True, text=True, cwd=os.path.abspath(root))
        if r.returncode:
            print(r.stderr[-3000:])
            return 1
        res.append(np.load(f))
    a, b = res
    worst = 0
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True)
        if not same:
            worst += 1
            fin = np.isfinite(a[k]) & np.isfinite(b[k])
            print(f"{k}: DIFFERS, max abs {np.max(np.abs(a[k][fin] - b[k][fin])):.3e}")
    print(f"{len(a.files)} arrays compared ({sum(a[k].size for k in a.files)} doubles), {worst} differ" + ("" if worst else ": identical, bit for bit"))
    return 1 if worst else 0


if __name__ == "__main__":
    sys.exit(main())
