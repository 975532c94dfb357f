"""Point-major kernel: work split "s bins per workgroup, waves per s bin, workgroups per (mu, v) plane", workgroups per CU in
the launch (POINT_CAP) and fused / separate chi-square vs batch size; resident buffers, config 3 and BOSS.
Usage: gpu_split_sweep.py [config3|boss] [batches...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases
from victor_amd import _native

which = sys.argv[1] if len(sys.argv) > 1 else "config3"
batches = [int(b) for b in sys.argv[2:]] or [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
opts, beta = (cases.boss_options("config"), True) if which == "boss" else (cases.synth_options(3), False)
fit = victor_amd.CCFFit(*opts)
eng = fit._get_engine()
o = eng.make_opts(fit.model, fit.fit_options)
bufs = [eng.alloc(700 * 12), eng.alloc(700), eng.alloc(700), eng.alloc(700 * eng.n_data)]


def timed(batch, reps=300):
    for _ in range(40):
        eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
    eng.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e6


t_end = time.perf_counter() + 0.4          # the runtime's one-off post-allocation stall must not land in a timed window
while time.perf_counter() < t_end:
    timed(1, 20)
_native.set_knob("VICTOR_HIP_MAPPING", "point")
splits = ["default", "1,4,1", "1,4,2", "1,4,4", "1,4,8", "1,2,1", "1,2,2", "1,2,4", "1,1,1", "1,1,2", "2,1,1", "2,1,2", "4,1,1", "4,1,2", "10,1,1"]
for batch in batches:
    rows = fit._fit_rows(cases.halton_params(max(batch, 2), with_beta=beta), fit.model)[:batch]
    eng.upload(bufs[0], rows)
    for extra in ({}, {"VICTOR_HIP_NO_FUSE": "1"}, {"VICTOR_HIP_POINT_CAP": "5"}, {"VICTOR_HIP_POINT_CAP": "10"}):
        for k, v in extra.items():
            _native.set_knob(k, v)
        line = f"{which} batch {batch:4d} {str(extra):36s}:"
        for split in splits:
            _native.set_knob("VICTOR_HIP_SPLIT", None if split == "default" else split)
            line += f" [{split}] {timed(batch):6.1f}"
        _native.set_knob("VICTOR_HIP_SPLIT", None)
        for k in extra:
            _native.set_knob(k, None)
        print(line, flush=True)
