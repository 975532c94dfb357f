"""Joint fit of several data vectors that share one parameter vector (density-split quantiles).

The reference mentions density-split centres only in passing (``ccf_model.py:28-30``) and has no joint-fit
class; BASELINE config "density-split 5-quantile joint fit" is defined as Q independent ``CCFFit`` blocks with a
block-diagonal covariance, so chi-square and log-likelihood add.  Each block keeps its own context (tables in
HBM) on the same GPU; one batch of parameter rows is run through every block.
"""

import numpy as np


import ctypes as C

from . import _native as N


class JointFit:
    def __init__(self, fits):
        self.fits = list(fits)
        if not self.fits:
            raise ValueError("need at least one fit")
        self._buffers = None

    # ------------------------------------------------------------------ device-resident joint evaluation
    def _plan(self, kwargs):
        """Engines, option block and row builder when every block can share ONE parameter upload and ONE option block
        (the normal case: same options in every block); None otherwise."""
        engines, blobs, rowsig = [], [], []
        opts = None
        for fit in self.fits:
            model = fit._merged(kwargs)
            fit._check_supported(model)
            fo = fit._merged_fit(kwargs)
            if fo["beta_interpolation"] == "likelihood" and not fit.fixed_data:
                return None
            eng = fit._get_engine(fit._engine_key(model), model["simpson_even"])
            o = eng.make_opts(model, fo)
            engines.append(eng)
            blobs.append(bytes(o))
            rowsig.append((fit._needs_beta(model) or not fit.fixed_data, fit._needs_fsigma8(model), model["bias"]))
            opts = o
        if len(set(blobs)) != 1 or len(set(rowsig)) != 1 or len({e.device for e in engines}) != 1:
            return None
        return engines, opts

    def _device_buffers(self, engines, n):
        lead = engines[0]
        ctxs = (C.c_void_p * len(engines))(*[e._ctx for e in engines])
        need = lead._lib.vk_joint_workspace_doubles(ctxs, len(engines), n)
        b = self._buffers
        if b is None or b["n"] < n or b["lead"] is not lead:
            if b is not None:
                for ptr in b["ptrs"]:
                    b["lead"].free(ptr)
            ptrs = [lead.alloc(n * N.VK_NPAR), lead.alloc(2 * n), lead.alloc(need)]
            b = self._buffers = {"n": n, "lead": lead, "ptrs": ptrs}
        return ctxs, b["ptrs"]

    def eval_device_async(self, engines, opts, d_rows, n, d_lnl, d_chi2, d_ws):
        """Enqueue the joint evaluation on buffers already in HBM (``bench.py``); ``engines[0].sync()`` waits for it."""
        lead = engines[0]
        ctxs = (C.c_void_p * len(engines))(*[e._ctx for e in engines])
        lead._check(lead._lib.vk_joint_eval_device_async(ctxs, len(engines), C.byref(opts), d_rows, int(n), d_lnl, d_chi2, d_ws))

    def log_likelihood_batch(self, params, **kwargs):
        """(lnL[n], chi2[n]) summed over the blocks: one parameter upload, every block's kernels enqueued without a host
        synchronisation in between, the sums taken on the device (``vk_joint_eval_device_async``)."""
        plan = self._plan(kwargs)
        if plan is None:
            return self._sequential(params, kwargs)
        engines, opts = plan
        fit = self.fits[0]
        rows = fit._fit_rows(params, fit._merged(kwargs))
        n = len(rows)
        if n == 0:
            return np.empty(0), np.empty(0)
        ctxs, (d_rows, d_out, d_ws) = self._device_buffers(engines, n)
        lead = engines[0]
        lead.upload(d_rows, rows)
        d_chi = C.c_void_p(d_out + 8 * n)
        lead._check(lead._lib.vk_joint_eval_device_async(ctxs, len(engines), C.byref(opts), d_rows, n, d_out, d_chi, d_ws))
        out = lead.download(d_out, 2 * n)
        return out[:n].copy(), out[n:].copy()

    def _sequential(self, params, kwargs):
        """Blocks with different options: one call per block, sums on the host."""
        lnl = chi2 = None
        for fit in self.fits:
            a, b = fit.log_likelihood_batch(params, **kwargs)
            lnl = a if lnl is None else lnl + a
            chi2 = b if chi2 is None else chi2 + b
        bad = ~np.isfinite(lnl)
        lnl[bad], chi2[bad] = -np.inf, np.inf
        return lnl, chi2

    def log_likelihood(self, params, **kwargs):
        lnl, chi2 = self.log_likelihood_batch(params, **kwargs)
        return float(lnl[0]), float(chi2[0])

    @property
    def n_data(self):
        return sum(len(f.s) * len(f.poles_s) for f in self.fits)
