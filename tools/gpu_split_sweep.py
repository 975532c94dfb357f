"""Point-major kernel: work split (s bins per workgroup, waves per s bin) vs batch size, resident, config 3 and BOSS."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases
from victor_amd import _native
_native.set_knob("VICTOR_HIP_MAPPING", "point")
for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    eng = fit._get_engine()
    o = eng.make_opts(fit.model, fit.fit_options)
    bufs = [eng.alloc(700 * 12), eng.alloc(700), eng.alloc(700), eng.alloc(700 * eng.n_data)]
    for batch in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
        rows = fit._fit_rows(cases.halton_params(batch, with_beta=beta), fit.model)
        eng.upload(bufs[0], rows)
        line = f"{name} batch {batch:4d}:"
        for split in ("default", "1,4", "1,2", "1,1", "2,1", "4,1", "10,1", "40,1"):
            if split == "default":
                os.environ.pop("VICTOR_HIP_SPLIT", None)
            else:
                _native.set_knob("VICTOR_HIP_SPLIT", split)
            for _ in range(30):
                eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
            eng.sync()
            t0 = time.perf_counter()
            for _ in range(300):
                eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
            eng.sync()
            dt = (time.perf_counter() - t0) / 300
            line += f"  [{split}] {dt*1e6:6.1f}us"
        print(line, flush=True)
