"""A/B of the lanes kernel's work split (config 3, 65536 and 131072 points, resident): one item per wave (default) against a
workgroup per 64-point chunk taking all 40 s bins (VICTOR_HIP_LANES_BY_CHUNK=1) - the mapping a chi-square fused into the lanes
kernel would need.  Prints K1 / K2 kernel milliseconds per launch, three rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _devlib import use_dev_library
use_dev_library()          # the lanes kernel lives in the development build of the library only
import victor_amd
from tests import cases
from victor_amd import _native

fit = victor_amd.CCFFit(*cases.synth_options(3))
eng = fit._get_engine()
o = eng.make_opts(fit.model, fit.fit_options)
for n in (65536, 131072):
    rows = fit._fit_rows(cases.halton_params(n), fit.model)
    bufs = [eng.alloc(rows.size), eng.alloc(n), eng.alloc(n), eng.alloc(n * eng.n_data)]
    eng.upload(bufs[0], rows)
    for rnd in range(3):
        for knob in (None, "1"):
            _native.set_knob("VICTOR_HIP_LANES_BY_CHUNK", knob)
            t_end = time.perf_counter() + 0.3
            while time.perf_counter() < t_end:
                eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3]); eng.sync()
            eng.timing(True); eng.read_timing(reset=True)
            for _ in range(10):
                eng.eval_device_async(o, bufs[0], n, bufs[1], bufs[2], bufs[3])
            eng.sync()
            k1, k2, launches = eng.read_timing(reset=True)
            eng.timing(False)
            print(f"n={n} round {rnd} {'workgroup per chunk' if knob else 'one item per wave   '}: K1 {k1 / launches:.3f} ms  K2 {k2 / launches:.3f} ms  ({eng.last_kernel()})", flush=True)
    _native.set_knob("VICTOR_HIP_LANES_BY_CHUNK", None)
    for b in bufs:
        eng.free(b)
