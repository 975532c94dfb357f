"""A few launches of the BOSS CMASS configuration under another RSD model / option set, buffers resident in HBM: the workload
behind profiles/r04/{dispersion,kaiser,euclid}_*.  Usage: gpu_model_loop.py <rsd_model> [launches] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases

rsd = sys.argv[1]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
fit = victor_amd.CCFFit(*cases.boss_options("config"))
model = fit._merged({"rsd_model": rsd})
eng = fit._get_engine(fit._engine_key(model))
o = eng.make_opts(model, fit.fit_options)
rows = fit._fit_rows(cases.halton_params(batch, with_beta=True), model)
bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
eng.upload(bufs[0], rows)
for _ in range(launches):
    eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
eng.sync()
print(eng.last_kernel(), "fused" if eng.last_fused() else "two launches")
