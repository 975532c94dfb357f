"""BOSS CMASS, cells kernel, resident: evals/s against batch size, one launch vs the same points in launches of 16384."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases

fit = victor_amd.CCFFit(*cases.boss_options("config"))
eng = fit._get_engine()
o = eng.make_opts(fit.model, fit.fit_options)
nmax = 262144
rows = fit._fit_rows(cases.halton_params(nmax, with_beta=True), fit.model)
bufs = [eng.alloc(rows.size), eng.alloc(nmax), eng.alloc(nmax), eng.alloc(nmax * eng.n_data)]
eng.upload(bufs[0], rows)


def run(n, chunk, reps):
    for _ in range(reps):
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            eng.eval_device_async(o, bufs[0] + lo * 96, m, bufs[1] + lo * 8, bufs[2] + lo * 8, bufs[3] + lo * eng.n_data * 8)
    eng.sync()


for n in (4096, 8192, 16384, 32768, 65536, 131072, 262144):
    line = f"boss batch {n:6d}:"
    for chunk in (n, 16384, 8192):
        if chunk > n:
            continue
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            run(n, chunk, 1)
        reps = max(2, int(0.5 / (n / 3.9e6)))
        t0 = time.perf_counter(); run(n, chunk, reps); dt = (time.perf_counter() - t0) / reps
        line += f"  launches of {chunk:6d}: {n/dt/1e6:6.3f} M/s"
    print(line, flush=True)
