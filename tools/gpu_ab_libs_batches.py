"""Same-box A/B of library builds over batch sizes: ms per batch (theory + likelihood, resident) for config 3 and BOSS at
1024 ... 262144 points.  Usage: gpu_ab_libs_batches.py libA.so libB.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import json, sys, time
sys.path.insert(0, %r)
import victor_amd
from tests import cases
out = {}
for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    eng = fit._get_engine()
    o = eng.make_opts(fit.model, fit.fit_options)
    nmax = 262144
    rows = fit._fit_rows(cases.halton_params(nmax, with_beta=beta), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(nmax), eng.alloc(nmax), eng.alloc(nmax * eng.n_data)]
    eng.upload(bufs[0], rows)
    for n in (1024, 4096, 16384, 65536, 262144):
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3]); eng.sync()
        reps = max(3, min(40, 400000 // n))
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3])
        eng.sync()
        out[f"{name}_{n}"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
print(json.dumps(out))
''' % ROOT
for rnd in range(2):
    for lib in sys.argv[1:]:
        res = subprocess.run([sys.executable, "-c", WORKER], env=dict(os.environ, VICTOR_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        print(f"round {rnd} {os.path.basename(lib):24s} {res.stdout.strip() or res.stderr[-300:]}", flush=True)
