"""Resident throughput vs batch size (config 3 and BOSS): looks for dips at the kernel-selection boundaries."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases
for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    eng = fit._get_engine()
    o = eng.make_opts(fit.model, fit.fit_options)
    for batch in (64, 128, 256, 512, 1000, 1024, 2000, 3000, 4500, 6000, 6900, 7000, 8192, 10000, 14000, 20000, 30000, 50000, 100000):
        rows = fit._fit_rows(cases.halton_params(batch, with_beta=beta), fit.model)
        bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
        eng.upload(bufs[0], rows)
        reps = max(3, min(200, 200000 // batch))
        for _ in range(3):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3]); eng.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f"{name} batch {batch:6d}: {dt*1e3:8.3f} ms -> {batch/dt:9.0f} evals/s ({eng.last_kernel()})", flush=True)
        for b in bufs: eng.free(b)
