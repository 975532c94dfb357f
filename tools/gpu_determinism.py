"""Repeat the same batches many times through every kernel family and require bit-identical outputs (race detector
for the workgroup-level synchronisation of the per-point kernels and the wave-level reductions)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _devlib import use_dev_library
use_dev_library()          # the lanes kernel lives in the development build of the library only
import numpy as np
import victor_amd
from tests import cases

def repeat(fit, label, rows, reps, **kw):
    first = None
    for i in range(reps):
        out = fit.theory_vector_batch(rows, **kw)
        l, c = fit.log_likelihood_batch(rows, **kw)
        if first is None:
            first = (out.copy(), l.copy(), c.copy())
            kern = fit._get_engine(fit._engine_key(fit._merged(kw))).last_kernel()
        else:
            assert np.array_equal(out, first[0], equal_nan=True), (label, i, "theory")
            assert np.array_equal(l, first[1], equal_nan=True) and np.array_equal(c, first[2], equal_nan=True), (label, i)
    print(f"{label}: {reps} repeats of {len(rows)} points bit-identical ({kern})", flush=True)

f3 = victor_amd.CCFFit(*cases.synth_options(3))
boss = victor_amd.CCFFit(*cases.boss_options("config"))
r3 = lambda n: f3._fit_rows(cases.halton_params(n), f3.model)
rb = lambda n: boss._fit_rows(cases.halton_params(n, with_beta=True), boss.model)
repeat(f3, "config3 lanes", r3(16384), 60)
repeat(f3, "config3 cells", r3(2000), 100)
repeat(f3, "config3 point-major (4 s-bins per workgroup)", r3(300), 200)
repeat(f3, "config3 point-major (teams of 4 waves)", r3(3), 300)
repeat(boss, "boss cells", rb(4096), 100)
repeat(boss, "boss point-major", rb(77), 300)
repeat(boss, "boss dispersion cells", rb(2048), 60, rsd_model="dispersion")
repeat(boss, "boss empirical cells", rb(2048), 60, empirical_corr=True)
repeat(boss, "boss kaiser generic", rb(2048), 100, rsd_model="kaiser")
