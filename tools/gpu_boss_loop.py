"""BOSS CMASS configuration, 65536 points resident in HBM, a few launches: the workload behind the cells-kernel profile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases

fit = victor_amd.CCFFit(*cases.boss_options("config"))
eng = fit._get_engine()
o = eng.make_opts(fit.model, fit.fit_options)
batch = 65536
rows = fit._fit_rows(cases.halton_params(batch, with_beta=True), fit.model)
bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
eng.upload(bufs[0], rows)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
eng.sync()
print(eng.last_kernel())
