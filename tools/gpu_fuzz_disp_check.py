"""Worst rows of the dispersion fuzz: fast kernel, generic kernel and the CPU oracle side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import victor_amd, victor_oracle as vo
from tests import cases
from victor_amd import _native
from tools.gpu_fuzz import params  # noqa

fit = victor_amd.CCFFit(*cases.synth_options(3))
ora = vo.OracleFit(*cases.synth_options(3))
kw = {"rsd_model": "dispersion"}
model = fit._merged(kw)
p = params(131072, False, 7, 1.0)
rows = fit._fit_rows(p, model)
_native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
ref = fit.theory_vector_batch(rows, **kw)
_native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
got = fit.theory_vector_batch(rows, **kw)
scale = np.max(np.abs(ref), axis=1, keepdims=True)
dev = np.max(np.abs(got - ref) / scale, axis=1)
dev[~np.isfinite(dev)] = 0
worst = np.argsort(dev)[-3:][::-1]
print("quantiles of row deviation fast vs generic:", np.quantile(dev, [0.5, 0.99, 0.9999, 1.0]))
for i in worst:
    q = {k: float(v[i]) for k, v in p.items()}
    want = ora.theory_multipole_vector(ora.s, dict(q), ora.poles_s, **kw)
    print(i, q, "fast-generic %.2e  fast-oracle %.2e  generic-oracle %.2e" % (
        dev[i], np.max(np.abs(got[i] - want)) / np.max(np.abs(want)), np.max(np.abs(ref[i] - want)) / np.max(np.abs(want))))
