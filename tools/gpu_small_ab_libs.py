"""Same-box A/B of library builds in the small-batch regime: microseconds per resident launch at 1, 2, 8, 16, 23, 64 points
(config 3 and BOSS), optionally with VICTOR_HIP_NO_ALONE=1 (argv: libs..., a trailing "noalone" adds that variant per lib)."""
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
    rows = fit._fit_rows(cases.halton_params(64, with_beta=beta), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(64), eng.alloc(64), eng.alloc(64 * eng.n_data)]
    eng.upload(bufs[0], rows)
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        eng.eval_device_async(o, bufs[0], 1, bufs[1], bufs[2], bufs[3]); eng.sync()
    for n in (1, 2, 8, 16, 23, 64):
        best = 1e9
        for _ in range(3):
            for _ in range(50):
                eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3])
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(400):
                eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3])
            eng.sync()
            best = min(best, (time.perf_counter() - t0) / 400)
        out[f"{name}_{n}"] = round(best * 1e6, 2)
print(json.dumps(out))
''' % ROOT
args = sys.argv[1:]
noalone = args and args[-1] == "noalone"
libs = args[:-1] if noalone else args
for rnd in range(2):
    for lib in libs:
        for knob in ((None, "1") if noalone else (None,)):
            env = dict(os.environ, VICTOR_HIP_LIB=os.path.abspath(lib))
            if knob:
                env["VICTOR_HIP_NO_ALONE"] = "1"
            res = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True)
            print(f"round {rnd} {os.path.basename(lib):26s} {'NO_ALONE' if knob else '        '} {res.stdout.strip() or res.stderr[-300:]}", flush=True)
