"""Where a small-batch launch of the point-major kernel spends its time: wall_clock64() marks (100 MHz) per workgroup from the
profiling build (python -m victor_amd.build --phases).  Usage: VICTOR_HIP_LIB=victor_amd/csrc/libvictor_hip_phases.so gpu_phases.py {3|boss} BATCH [api]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import victor_amd  # noqa: E402
from tests import cases  # noqa: E402
from victor_amd import _native  # noqa: E402

which, batch = sys.argv[1], int(sys.argv[2])
via_api = len(sys.argv) > 3 and sys.argv[3] == "api"      # one point through CCFFit.log_likelihood: parameters in host-mapped memory
fit = victor_amd.CCFFit(*(cases.boss_options("config") if which == "boss" else cases.synth_options(int(which))))
hp = cases.halton_params(max(batch, 2), with_beta=which == "boss")
rows = fit._fit_rows(hp, fit.model)[:batch]
eng = fit._get_engine()
opts = eng.make_opts(fit.model, fit.fit_options)
bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
eng.upload(bufs[0], rows)
lib = _native.load()
stamps = np.zeros((4096, 16), dtype=np.int64)
acc = []
one = cases.point(hp, 0)
for rep in range(60):
    if via_api:
        fit.log_likelihood(one)
    else:
        eng.eval_device_async(opts, bufs[0], batch, bufs[1], bufs[2], bufs[3])
    eng.sync()
    if rep < 20:
        continue
    assert lib.vk_debug_read_stamps(stamps.ctypes.data_as(C.c_void_p)) == 0
    live = stamps[:, 0] > 0
    s = stamps[live].astype(float)
    t0 = s[:, 0].min()
    us = (s - t0) / 100.0
    us[s == 0] = np.nan
    last = np.nanargmax(us[:, 5]) if np.any(~np.isnan(us[:, 5])) else None
    acc.append([live.sum(), np.nanmax(us[:, 0]), np.nanmedian(us[:, 1] - us[:, 0]), np.nanmedian(us[:, 2] - us[:, 1]),
                np.nanmedian(us[:, 3] - us[:, 2]), np.nanmedian(us[:, 4] - us[:, 3]), np.nanmax(us[:, 4]),
                np.nanmax(us[:, 5] - us[:, 4]), np.nanmax(us[:, 5]), np.nanmax(us[:, 6] - us[:, 4]) if np.any(s[:, 6] > 0) else np.nan])
# distribution over the workgroups of the last repetition: start, integrand phase, end of the integrand phase
pc = lambda v: " / ".join(f"{x:.1f}" for x in np.nanpercentile(v, [0, 10, 50, 90, 100]))   # noqa: E731
print(f"  workgroup start {pc(us[:, 0])}; integrand phase {pc(us[:, 3] - us[:, 2])}; integrand done at {pc(us[:, 3])}  (min / p10 / median / p90 / max, us)")
fin = int(np.nanargmax(us[:, 5]))
marks = us[fin, [4, 6, 8, 9, 10, 11, 12, 5]]
print("  finishing workgroup, us after its counter mark: partial sums gathered %.2f | chi2 entry %.2f | beta searches done %.2f | residual in LDS %.2f | "
      "quadratic form %.2f | block sum %.2f | results stored %.2f" % tuple(marks[1:] - marks[0]))
a = np.median(np.array(acc), axis=0)
print(f"{which} batch {batch}: workgroups {a[0]:.0f}; last workgroup starts at {a[1]:.2f} us; per workgroup (median): staging {a[2]:.2f}, "
      f"point set-up {a[3]:.2f}, integrand + projection {a[4]:.2f}, completion counter {a[5]:.2f}; all counters done at {a[6]:.2f}; "
      f"tail (gather + chi2) {a[7]:.2f} (of which the gather of the partial sums {a[9]:.2f}); kernel end (last mark) at {a[8]:.2f} us")
