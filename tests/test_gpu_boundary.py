"""Round-4 boundary behaviour on the GPU: development knobs are inert in a production environment, quadrature rules with zero
weights, the scratch semantics of the device-resident theory workspace."""

import ctypes as C
import os

import numpy as np
import pytest

import victor_amd
from victor_amd import _native
from tests import cases
from tests.devlib import KERNEL_OF, mapped
from tests.tolerances import assert_same_chi2, assert_same_theory, chi2_bound

pytestmark = pytest.mark.gpu


def test_inherited_knobs_are_ignored_without_the_development_switch():
    """A VICTOR_HIP_* tuning variable that a production process merely inherits must not change anything: the library reads
    the knobs only under VICTOR_HIP_DEV=1 (victor_hip.hip: load_knobs)."""
    fit = victor_amd.CCFFit(*cases.synth_options(3))
    hp = cases.halton_params(8192)
    base = fit.log_likelihood_batch(hp)
    eng = fit._get_engine()
    assert eng.last_kernel() == "vk_theory_cells_kernel"
    saved = os.environ.pop("VICTOR_HIP_DEV", None)
    try:
        os.environ["VICTOR_HIP_MAPPING"] = "lanes"
        os.environ["VICTOR_HIP_FORCE_GENERIC"] = "1"
        os.environ["VICTOR_HIP_NO_FUSE"] = "1"
        _native.load().vk_knobs_refresh()
        got = fit.log_likelihood_batch(hp)
        assert eng.last_kernel() == "vk_theory_cells_kernel"           # not the lanes kernel, not the generic one
        assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1])
        one = fit.log_likelihood(cases.point(hp, 3))
        assert eng.last_fused()                                        # NO_FUSE ignored as well
        assert_same_chi2(one[1], base[1][3], chi2_bound(fit, {k: v[3:4] for k, v in hp.items()}), what="single point vs batch row")
        os.environ["VICTOR_HIP_DEV"] = "1"                             # the switch on: the same variables now act
        os.environ["VICTOR_HIP_MAPPING"] = "point"
        _native.load().vk_knobs_refresh()
        fit.log_likelihood_batch(hp)
        assert eng.last_kernel() == "vk_theory_kernel"                 # FORCE_GENERIC
        os.environ.pop("VICTOR_HIP_FORCE_GENERIC")
        _native.load().vk_knobs_refresh()
        fit.log_likelihood_batch(hp)
        assert eng.last_kernel() == "vk_theory_fast_kernel"            # MAPPING=point
    finally:
        for k in ("VICTOR_HIP_MAPPING", "VICTOR_HIP_FORCE_GENERIC", "VICTOR_HIP_NO_FUSE"):
            os.environ.pop(k, None)
        if saved is not None:
            os.environ["VICTOR_HIP_DEV"] = saved
        _native.load().vk_knobs_refresh()


def test_quadrature_rule_with_zero_weights(monkeypatch):
    """A caller's velocity rule may contain nodes of weight zero: the kernels whose node loop runs over weight groups leave them
    out (a zero weight could not close its group, vk_common.h: load_node) and agree with the generic kernel, which multiplies
    every node by its weight."""
    from victor_amd import tables as T
    real = T.simpson_weights

    def with_zeros(n, rule=None):
        w = real(n, rule).copy()
        w[[0, 7, 20, 21, n - 1]] = 0.0
        return w

    hp = cases.halton_params(300)
    plain = victor_amd.CCFFit(*cases.synth_options(3)).log_likelihood_batch(hp)
    monkeypatch.setattr(T, "simpson_weights", with_zeros)
    fit = victor_amd.CCFFit(*cases.synth_options(3))
    out = {}
    for mapping in ("generic", "point", "cells", "lanes"):
        with mapped(fit, mapping) as f:                  # "lanes": the development build of the library (tests/devlib.py)
            out[mapping] = f.log_likelihood_batch(hp)
            assert f._get_engine().last_kernel() == KERNEL_OF[mapping]
    bound = chi2_bound(fit, hp, ulps=1024)                           # against the generic kernel: another arithmetic
    for mapping in ("point", "cells", "lanes"):
        assert_same_chi2(out[mapping][1], out["generic"][1], bound, what=f"zero-weight rule, {mapping} vs generic")
    assert np.max(np.abs(out["generic"][1] / plain[1] - 1)) > 1e-6        # the rule really was another one


def test_non_finite_or_denormal_weights_are_refused(monkeypatch):
    from victor_amd import tables as T
    real = T.simpson_weights
    for bad in (np.nan, np.inf, 1e-310):
        def rule(n, r=None, bad=bad):
            w = real(n, r).copy()
            w[3] = bad
            return w
        monkeypatch.setattr(T, "simpson_weights", rule)
        fit = victor_amd.CCFFit(*cases.synth_options(2))
        with pytest.raises(_native.NativeError, match="finite"):
            fit.log_likelihood({"fsigma8": 0.5, "sigma_v": 380.0})


def test_theory_workspace_is_scratch_when_the_likelihood_is_fused():
    """vk_eval_batch_device_async: with lnL / chi2 requested the workspace is scratch - a fused cells launch never writes the
    theory vectors to HBM (a sentinel survives) - and with d_lnl = d_chi2 = NULL it returns the theory vectors."""
    fit = victor_amd.CCFFit(*cases.boss_options("config"))
    eng = fit._get_engine()
    opts = eng.make_opts(fit.model, fit.fit_options)
    n = 8192                                                           # BOSS, blended covariance: fused from 8192 points on
    rows = fit._fit_rows(cases.halton_params(n, with_beta=True), fit.model)
    d_rows, d_lnl, d_chi, d_ws = eng.alloc(rows.size), eng.alloc(n), eng.alloc(n), eng.alloc(n * eng.n_data)
    try:
        eng.upload(d_rows, rows)
        sentinel = np.full(n * eng.n_data, -12345.0)
        eng.upload(d_ws, sentinel)
        eng.eval_device_async(opts, d_rows, n, d_lnl, d_chi, d_ws)
        eng.sync()
        assert eng.last_kernel() == "vk_theory_cells_kernel" and eng.last_fused()
        assert np.array_equal(eng.download(d_ws, n * eng.n_data), sentinel)       # nothing was stored
        chi2 = eng.download(d_chi, n)
        lib = _native.load()
        eng._check(lib.vk_eval_batch_device_async(eng._ctx, C.byref(opts), d_rows, n, None, None, d_ws))
        eng.sync()
        theory = eng.download(d_ws, n * eng.n_data).reshape(n, eng.n_data)
        assert not np.any(theory == -12345.0)
        assert_same_theory(theory[:64], fit.theory_vector_batch(rows[:64]), what="device-resident theory vs host call")
        assert_same_chi2(fit.log_likelihood_batch(rows[:64])[1], chi2[:64], chi2_bound(fit, rows[:64]), what="fused 8192 vs 64 on their own")
    finally:
        for p in (d_rows, d_lnl, d_chi, d_ws):
            eng.free(p)


_LEDGER_CHILD = r'''
import json, sys, time
root, n_ctx, hold = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
sys.path.insert(0, root)
import ctypes as C
import numpy as np
import victor_amd
from tests import cases
from victor_amd.engine import Engine
fit = victor_amd.CCFFit(*cases.boss_options("config"))
first = fit._get_engine()
key = fit._engine_key(fit._merged({}))
engines = [first] + [Engine(fit, fit, matter_model=key, simpson_even=first.simpson_even) for _ in range(n_ctx - 1)]
opts = first.make_opts(fit.model, fit.fit_options)
rows = fit._fit_rows(cases.halton_params(5, with_beta=True), fit.model)      # five BOSS points: planes split in two, one waiter each
polled, out = [], []
for e in engines:
    out.append(np.concatenate(e.eval_batch(opts, rows)[:2]).tolist())
    polled.append(e.last_polled())
others, mine = C.c_int32(), C.c_int32()
rc = first._lib.vk_poll_device_reserved(first._ctx, C.byref(others), C.byref(mine))
print(json.dumps({"polled": polled, "others": others.value, "mine": mine.value, "rc": rc, "same": all(o == out[0] for o in out)}), flush=True)
sys.stdin.readline() if hold else None
'''


def test_polling_reservations_are_kept_device_wide_across_processes():
    """The ledger of reserved waiters (victor_amd/csrc/vk_ledger.cpp; one file per GPU in /dev/shm): a process reserves 5 per
    context for five-point BOSS launches up to its own budget of 32 (six contexts, the seventh hands over through the
    counters); a second process sees those 30 and gets its own 30; a third finds 60 taken and is granted nothing (63 is the
    bound) - all of them return the same bits; when the first two have gone their slots no longer count."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def start(n_ctx, hold):
        return subprocess.Popen([sys.executable, "-c", _LEDGER_CHILD, root, str(n_ctx), "1" if hold else "0"], stdin=subprocess.PIPE,
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def first_line(p):
        line = p.stdout.readline()
        assert line, p.stderr.read()[-2000:]
        return json.loads(line)

    held = []
    try:
        a = start(7, True)
        ra = first_line(a)
        held.append(a)
        base = ra["others"]                                   # whatever other processes of this user hold on the GPU already (normally 0)
        assert ra["rc"] == 0 and ra["polled"] == [True] * 6 + [False] and ra["mine"] == 30 and ra["same"], ra
        b = start(6, True)
        rb = first_line(b)
        held.append(b)
        assert rb["polled"] == [True] * 6 and rb["mine"] == 30 and rb["others"] == base + 30 and rb["same"], rb
        c = start(2, False)
        rc_ = first_line(c)
        c.wait(timeout=60)
        # 63 - 60 = 3 waiters left on the device: a five-point launch gets none of them
        assert rc_["polled"] == [False, False] and rc_["mine"] == 0 and rc_["others"] == base + 60 and rc_["same"], rc_
        assert rc_["same"] and ra["same"]
    finally:
        for p in held:
            try:
                p.stdin.write("\n")
                p.stdin.flush()
            except OSError:
                pass
            p.wait(timeout=60)
    d = start(2, False)
    rd = first_line(d)
    d.wait(timeout=60)
    assert rd["polled"] == [True, True] and rd["others"] == base and rd["mine"] == 10, rd      # the dead processes' slots do not count
