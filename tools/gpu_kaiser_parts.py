import os, sys, time
sys.path.insert(0, "/root/repo")
import victor_amd
from victor_amd import _native
from tests import cases
batch = 16384
fit = victor_amd.CCFFit(*cases.boss_options("config"))
def timed(kw, knobs, like=True):
    model = fit._merged(kw)
    eng = fit._get_engine(fit._engine_key(model))
    o = eng.make_opts(model, fit.fit_options)
    rows = fit._fit_rows(cases.halton_params(batch, with_beta=True), model)
    bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
    eng.upload(bufs[0], rows)
    for k, v in knobs.items(): _native.set_knob(k, v)
    best = 1e9
    import ctypes as C
    def go():
        if like: eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        else: eng._check(eng._lib.vk_eval_batch_device_async(eng._ctx, C.byref(o), bufs[0], batch, None, None, bufs[3]))
    for _ in range(3):
        for _ in range(5): go()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(20): go()
        eng.sync()
        best = min(best, (time.perf_counter() - t0) / 20)
    for k in knobs: _native.set_knob(k, None)
    for b in bufs: eng.free(b)
    return best * 1e3
kw = {"rsd_model": "kaiser"}
print("fused            %.3f ms" % timed(kw, {}))
print("two launches     %.3f ms" % timed(kw, {"VICTOR_HIP_NO_FUSE": "1"}))
print("theory only      %.3f ms" % timed(kw, {}, like=False))
print("theory only, no shift %.3f ms" % timed({"rsd_model": "kaiser", "kaiser_coord_shift": False}, {}, like=False))
for p in ("1", "2", "3"):
    print("fused, parts", p, " %.3f ms" % timed(kw, {"VICTOR_HIP_CELLS_PARTS": p}))
