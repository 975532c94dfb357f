"""kaiser / euclid_special on BOSS CMASS, batch 16384 resident: the cells kernel (fused chi-square) against the generic kernel +
K2 on the same box, and the cells kernel without the coordinate shift (what is left is set-up, final evaluation, projection
and the per-point work).  ms per batch, best of 3 rounds of 20 launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from victor_amd import _native
from tests import cases

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
fit = victor_amd.CCFFit(*cases.boss_options("config"))


def timed(kw, generic):
    model = fit._merged(kw)
    eng = fit._get_engine(fit._engine_key(model))
    o = eng.make_opts(model, fit.fit_options)
    rows = fit._fit_rows(cases.halton_params(batch, with_beta=True), model)
    bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
    eng.upload(bufs[0], rows)
    _native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1" if generic else None)
    best = 1e9
    for _ in range(3):
        for _ in range(5):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        best = min(best, (time.perf_counter() - t0) / 20)
    name = eng.last_kernel()
    _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
    for b in bufs:
        eng.free(b)
    return best * 1e3, name


for label, kw in (("kaiser", {"rsd_model": "kaiser"}), ("euclid_special", {"rsd_model": "euclid_special"}),
                  ("kaiser, no coordinate shift", {"rsd_model": "kaiser", "kaiser_coord_shift": False}),
                  ("kaiser, niter 2", {"rsd_model": "kaiser", "niter": 2})):
    g, gk = timed(kw, True)
    c, ck = timed(kw, False)
    print(f"{label:30s} generic {g:7.3f} ms ({gk})   cells {c:7.3f} ms ({ck})   x{g / c:.2f}", flush=True)
