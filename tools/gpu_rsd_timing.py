"""Throughput of the other RSD branches (generic kernel) on the config-3 and BOSS tables."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import victor_amd
from tests import cases
from victor_amd import _native

def run(fit, label, batch, beta, **kw):
    model = fit._merged(kw)
    eng = fit._get_engine(fit._engine_key(model))
    o = eng.make_opts(model, fit.fit_options)
    hp = cases.halton_params(batch, with_beta=beta)
    if not beta:
        hp = dict(hp, beta=0.4)          # required key for linear_bias on fixed tables, value unused
    rows = fit._fit_rows(hp, model)
    bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
    eng.upload(bufs[0], rows)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3]); eng.sync()
    t0 = time.perf_counter()
    for _ in range(4):
        eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
    eng.sync()
    dt = (time.perf_counter() - t0) / 4
    print(f"{label}: {dt*1e3:.2f} ms/batch {batch/dt:.0f} evals/s ({eng.last_kernel()})", flush=True)
    for b in bufs: eng.free(b)

m, d = cases.boss_options("config")
m["input_model_data_file"] = "boss/measured_model.npy"
m["realspace_ccf"]["from_data"] = True
d["covariance_matrix"]["data_file"] = "boss/cov_md_iso.npy"
fd = victor_amd.CCFFit(m, d)
run(fd, "boss from_data (measured model + MD covariance)", 16384, True)
_native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
run(fd, "boss from_data, generic kernel", 16384, True)
_native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
for name, opts, beta in (("config3", cases.synth_options(3), False), ("boss", cases.boss_options("config"), True)):
    fit = victor_amd.CCFFit(*opts)
    for kw in ({}, {"rsd_model": "dispersion"}, {"rsd_model": "kaiser"}, {"rsd_model": "euclid_special"},
               {"empirical_corr": True}, {"matter_model": "linear_bias"}):
        run(fit, f"{name} {kw}", 16384, beta, **kw)

# the beta-dependent velocity profile (linear_bias on the reconstructed BOSS ccf) in the combinations that used to be generic-only
boss = victor_amd.CCFFit(*cases.boss_options("config"))
for kw in ({"matter_model": "linear_bias", "rsd_model": "dispersion"}, {"matter_model": "linear_bias", "empirical_corr": True},
           {"matter_model": "linear_bias", "empirical_corr": True, "rsd_model": "dispersion"}):
    run(boss, f"boss {kw}", 16384, True, **kw)
    _native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
    run(boss, f"boss {kw}, generic kernel", 16384, True, **kw)
    _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
