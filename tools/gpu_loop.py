"""A few launches of one workload on buffers resident in HBM: the loop behind the PMC comparisons of library builds.
Usage: gpu_loop.py {3|2|boss} [launches] [batch]      (VICTOR_HIP_LIB selects the build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd
from tests import cases

which = sys.argv[1]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
boss = which == "boss"
fit = victor_amd.CCFFit(*(cases.boss_options("config") if boss else cases.synth_options(int(which))))
eng = fit._get_engine()
o = eng.make_opts(fit.model, fit.fit_options)
rows = fit._fit_rows(cases.halton_params(batch, with_beta=boss), fit.model)
bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
eng.upload(bufs[0], rows)
for _ in range(launches):
    eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
eng.sync()
print(eng.last_kernel())
