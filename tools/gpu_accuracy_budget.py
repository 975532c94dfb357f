"""End-to-end accuracy budget of the fast theory kernels: over the wide prior box of tools/gpu_fuzz.py,
every fast mapping against the generic kernel (third-order refinements, library-grade exp) - max |d xi_l| / max |xi_l| of the
theory vectors and max relative d chi2 - and against the CPU oracle on a sample.  The budget is 1e-10 on both (the contract is
1e-6, the parity tests hold 1e-9).  Usage: gpu_accuracy_budget.py [points per case, default 65536] [oracle points, default 64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _devlib import use_dev_library
use_dev_library()          # the lanes kernel lives in the development build of the library only
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import victor_amd
from tests import cases
from victor_amd import _native
from tools.gpu_fuzz import params


def budget(fit, label, beta, n, n_oracle, ofit=None, **kw):
    model = fit._merged(kw)
    rows = fit._fit_rows(params(n, beta, 7, 2.0), model)
    _native.set_knob("VICTOR_HIP_FORCE_GENERIC", "1")
    ref_t = fit.theory_vector_batch(rows, **kw)
    ref_l, ref_c = fit.log_likelihood_batch(rows, **kw)
    _native.set_knob("VICTOR_HIP_FORCE_GENERIC", None)
    scale = np.max(np.abs(ref_t), axis=1, keepdims=True)
    worst = {}
    for mapping in ("lanes", "cells", "point"):
        _native.set_knob("VICTOR_HIP_MAPPING", mapping)
        got_t = fit.theory_vector_batch(rows, **kw)
        got_l, got_c = fit.log_likelihood_batch(rows, **kw)
        kern = fit._get_engine(fit._engine_key(model)).last_kernel()
        _native.set_knob("VICTOR_HIP_MAPPING", None)
        ok = np.isfinite(ref_c) & np.isfinite(got_c)
        dxi = np.max(np.abs(got_t[ok] - ref_t[ok]) / scale[ok])
        dchi = np.max(np.abs(got_c[ok] / ref_c[ok] - 1))
        worst[mapping] = (dxi, dchi)
        print(f"{label} [{mapping} -> {kern[10:]}] vs generic kernel, {int(ok.sum())} rows: max dxi/max|xi| {dxi:.2e}, max rel dchi2 {dchi:.2e} "
              f"(median {np.median(np.abs(got_c[ok] / ref_c[ok] - 1)):.1e})", flush=True)
    if ofit is not None and n_oracle > 0:
        idx = np.linspace(0, n - 1, n_oracle).astype(int)
        p = params(n, beta, 7, 2.0)
        got_l, got_c = fit.log_likelihood_batch(rows[idx], **kw)
        want = np.array([ofit.log_likelihood(cases.point(p, int(i)), **kw)[1] for i in idx])
        ok = np.isfinite(want) & np.isfinite(got_c)
        print(f"{label} [default path] vs CPU oracle, {int(ok.sum())} rows: max rel dchi2 {np.max(np.abs(got_c[ok] / want[ok] - 1)):.2e}", flush=True)
    return worst


def main():
    import victor_oracle as vo
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    n_or = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    out = {}
    for cfg in (3, 2):
        opts = cases.synth_options(cfg)
        out[f"config{cfg}"] = budget(victor_amd.CCFFit(*opts), f"config{cfg}", False, n, n_or, vo.OracleFit(*opts))
    opts = cases.boss_options("config")
    boss, oboss = victor_amd.CCFFit(*opts), vo.OracleFit(*opts)
    out["boss"] = budget(boss, "boss", True, n, n_or, oboss)
    out["boss aniso"] = budget(boss, "boss anisotropic", True, n, 0, assume_isotropic=False)
    out["boss linear_bias"] = budget(boss, "boss linear_bias", True, n, 0, matter_model="linear_bias")
    worst_xi = max(v[0] for c in out.values() for v in c.values())
    worst_chi = max(v[1] for c in out.values() for v in c.values())
    print(f"WORST over every case and mapping: dxi/max|xi| {worst_xi:.2e}, rel dchi2 {worst_chi:.2e}  (budget 1e-10)")


if __name__ == "__main__":
    main()
