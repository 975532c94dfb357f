"""Same-box A/B of library builds: kernel time of the theory launch for BOSS and config 3 (cells kernel) at 65536
points.  Usage: gpu_ab_libs.py libA.so libB.so ...   (each build is timed in its own process, the round is repeated)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import victor_amd
from tests import cases
out = {}
for name, opts, beta in (("boss", cases.boss_options("config"), True), ("config3", cases.synth_options(3), False)):
    fit = victor_amd.CCFFit(*opts)
    eng = fit._get_engine()
    o = eng.make_opts(fit.model, fit.fit_options)
    n = 65536
    rows = fit._fit_rows(cases.halton_params(n, with_beta=beta), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(n), eng.alloc(n), eng.alloc(n * eng.n_data)]
    eng.upload(bufs[0], rows)
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3]); eng.sync()
    eng.timing(True); eng.read_timing(reset=True)
    for _ in range(12):
        eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3])
    eng.sync()
    k1, k2, launches = eng.read_timing(reset=True)
    out[name] = [round(k1 / launches, 3), round(k2 / launches, 3), eng.last_kernel()[10:15]]
print(json.dumps(out))
''' % ROOT
libs = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, VICTOR_HIP_LIB=os.path.abspath(lib))
        res = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True)
        print(f"round {rnd} {os.path.basename(lib):40s} {res.stdout.strip() or res.stderr[-300:]}", flush=True)
