import os, subprocess, sys, json
ROOT = os.getcwd()
W = r'''
import sys, time, json
sys.path.insert(0, %r)
import numpy as np, victor_amd
from tests import cases
fit = victor_amd.CCFFit(*cases.boss_options("config"))
eng = fit._get_engine(); o = eng.make_opts(fit.model, fit.fit_options)
n = 65536
hp = cases.halton_params(n, with_beta=True)
out = {}
for label, perm in (("halton order", None), ("shuffled", np.random.default_rng(3).permutation(n))):
    rows = fit._fit_rows(hp, fit.model)
    if perm is not None: rows = np.ascontiguousarray(rows[perm])
    bufs = [eng.alloc(rows.size), eng.alloc(n), eng.alloc(n), eng.alloc(n * eng.n_data)]
    eng.upload(bufs[0], rows)
    best = 1e9
    for _ in range(3):
        for _ in range(2): eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3])
        eng.sync(); t0 = time.perf_counter()
        for _ in range(8): eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3])
        eng.sync(); best = min(best, (time.perf_counter() - t0) / 8)
    out[label] = round(best * 1e3, 3)
    for b in bufs: eng.free(b)
print(json.dumps(out))
''' % ROOT
for rnd in range(2):
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, "-c", W], env=dict(os.environ, VICTOR_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        print(rnd, os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-400:], flush=True)
