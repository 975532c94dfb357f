"""kaiser / euclid_special on the cells kernel with several consecutive points per workgroup (VICTOR_HIP_CELLS_PPW; the workgroup
keeps its LDS image): resident launches of the BOSS configuration, same box, the settings interleaved over several rounds;
results must be bit-identical to one point per workgroup.  Usage: python tools/gpu_kaiser_ppw_sweep.py [batch ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import victor_amd
import workloads as cases
from victor_amd import _native

batches = [int(a) for a in sys.argv[1:]] or [2048, 16384, 65536]
fit = victor_amd.CCFFit(*cases.boss_options("config"))
for rsd in ("kaiser", "euclid_special"):
    model = fit._merged({"rsd_model": rsd})
    eng = fit._get_engine(fit._engine_key(model))
    o = eng.make_opts(model, fit.fit_options)
    for batch in batches:
        rows = fit._fit_rows(cases.halton_params(batch, with_beta=True), model)
        bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
        eng.upload(bufs[0], rows)
        launch = lambda: eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])      # noqa: E731
        best, ref = {}, None
        for rnd in range(4):
            for ppw in (1, 2, 3, 4, 6, 8, 16):
                _native.set_knob("VICTOR_HIP_CELLS_PPW", str(ppw))
                for _ in range(5):
                    launch()
                eng.sync()
                t0 = time.perf_counter()
                for _ in range(20):
                    launch()
                eng.sync()
                dt = (time.perf_counter() - t0) / 20
                best[ppw] = min(best.get(ppw, 1e9), dt)
                out = np.concatenate([eng.download(bufs[1], batch), eng.download(bufs[2], batch)])
                if ref is None:
                    ref = out
                assert np.array_equal(out, ref), (rsd, batch, ppw)
        _native.set_knob("VICTOR_HIP_CELLS_PPW", None)
        print(f"{rsd:15s} batch {batch:6d}: " + "  ".join(f"ppw {p}: {1e3 * t:.4f} ms ({best[1] / t:.2f} x)" for p, t in best.items()),
              flush=True)
        for b in bufs:
            eng.free(b)
