"""Same-box A/B of library builds on the models with a fixed-point coordinate iteration: kaiser, euclid_special and dispersion on
BOSS CMASS (batch 16384 resident; ms per batch, best of 3 x 20 launches) and the chi-squares of the first build as the
yardstick for the others.  Usage: gpu_rsd_ab.py lib1.so lib2.so ...   (each build runs in a process of its own)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import victor_amd
from tests import cases
batch = 16384
fit = victor_amd.CCFFit(*cases.boss_options("config"))
out = {}
chis = {}
for label, kw in (("kaiser", {"rsd_model": "kaiser"}), ("euclid_special", {"rsd_model": "euclid_special"}), ("dispersion", {"rsd_model": "dispersion"})):
    model = fit._merged(kw)
    eng = fit._get_engine(fit._engine_key(model))
    o = eng.make_opts(model, fit.fit_options)
    rows = fit._fit_rows(cases.halton_params(batch, with_beta=True), model)
    bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
    eng.upload(bufs[0], rows)
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(20 if label != "dispersion" else 6):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        best = min(best, (time.perf_counter() - t0) / (20 if label != "dispersion" else 6))
    out[label] = {"ms": best * 1e3, "kernel": eng.last_kernel()}
    chis[label] = eng.download(bufs[2], batch)
    for b in bufs:
        eng.free(b)
np.savez(sys.argv[2], **chis)
print(json.dumps(out))
'''
libs = sys.argv[1:]
ref = None
with tempfile.TemporaryDirectory() as tmp:
    for rounds in range(2):                       # two rounds, alternating: box drift shows as a difference between them
        for lib in libs:
            npz = os.path.join(tmp, os.path.basename(lib) + ".npz")
            env = dict(os.environ, VICTOR_HIP_LIB=os.path.abspath(lib))
            res = subprocess.run([sys.executable, "-c", CHILD, ROOT, npz], env=env, capture_output=True, text=True)
            if res.returncode != 0:
                print(lib, "FAILED", res.stderr[-800:])
                continue
            out = json.loads(res.stdout.strip().splitlines()[-1])
            import numpy as np
            chi = dict(np.load(npz))
            if ref is None:
                ref = chi
            dev = {k: float(np.nanmax(np.abs(chi[k] / ref[k] - 1))) for k in chi}
            print(f"round {rounds} {os.path.basename(lib):28s} " + "  ".join(f"{k} {v['ms']:7.3f} ms" for k, v in out.items())
                  + "   max rel dchi2 vs first build: " + ", ".join(f"{k} {v:.1e}" for k, v in dev.items()), flush=True)
