"""``CCFFit``: chi-square and log-likelihood against measured multipoles, evaluated on the GPU.

Drop-in for ``victor.CCFFit`` (reference: ``victor/ccf_fit.py:10-483``): same constructor, attributes
(``s``, ``poles_s``, ``covmat``, ``icov``, ``fit_options`` ...) and method signatures.  ``chi_squared`` and
``log_likelihood`` run one parameter point through the same kernels that ``log_likelihood_batch``
uses for thousands (a batch of one is not special-cased).
"""

import ctypes as C
import os

import numpy as np

from . import _native as N
from . import tables as T
from . import utils
from .ccf_model import CCFModel
from .utils import InputError


_SCALARS = (float, int, np.float64)     # the common types of a sampler's parameter values: no np.ndim() call needed


class CCFFit(CCFModel):
    """Fits of the CCF model to measured redshift-space multipoles."""

    def __init__(self, model, data, device=0, broker=None):
        """``broker``: name of a GPU owner process to send plain single-point ``log_likelihood`` calls to
        (:mod:`victor_amd.broker`), ``"auto"`` to start / share one per job, ``False`` for never; ``None`` (default) reads the
        environment variable ``VICTOR_HIP_BROKER``.  The reference's constructor has the first two arguments
        (``ccf_fit.py:15``)."""
        super().__init__(model, device=device)
        if broker is None:
            broker = os.environ.get("VICTOR_HIP_BROKER", "") or False
        self._broker_spec = broker
        self._broker_client = None
        if broker:
            import copy
            self._broker_blocks = (copy.deepcopy(model), copy.deepcopy(data))
        base_dir = data.get("dir", "")
        data_fn = os.path.join(base_dir, data["redshift_space_ccf"].get("data_file"))
        cov_fn = os.path.join(base_dir, data["covariance_matrix"].get("data_file"))
        for fn in (data_fn, cov_fn):
            if not os.path.isfile(fn):
                raise InputError(f"Data file {fn} not found")
        self._load_redshiftspace_ccf(data["redshift_space_ccf"], data_fn)
        self._load_covariance_matrix(data["covariance_matrix"], cov_fn)
        self.fit_options = {"beta_interpolation": data.get("beta_interpolation", "datavector"),
                            "likelihood": data.get("likelihood", {"form": "Gaussian"})}
        if broker:
            import copy
            self._broker_pristine = (copy.deepcopy(self.model), copy.deepcopy(self.fit_options))

    # ------------------------------------------------------------------ set-up (host) -----
    def _load_redshiftspace_ccf(self, ccf, input_fn):
        """Data multipoles (reference: ccf_fit.py:44-114)."""
        input_data = utils.read_input_file(input_fn, self.extensions)
        isim = ccf.get("simulation_number", None)
        if isim is not None and not isinstance(isim, int):
            raise InputError("If provided, simulation_number must be an integer")
        self.fixed_data = not ccf.get("reconstruction", False)
        if not self.fixed_data:
            beta_key = ccf.get("beta_key", None)
            if beta_key and beta_key in input_data:
                self.beta_ccf = np.asarray(input_data[beta_key], dtype=float)
                if not np.all(np.diff(self.beta_ccf) > 0):
                    raise InputError("Redshift-space beta grid must be strictly monotonically increasing")
            elif self.fixed_real_input:
                raise InputError("Reconstruction beta information required for redshift-space ccf but not found")
            else:
                self.beta_ccf = self.beta
        fmt = ccf.get("format", "multipoles")
        ccf_keys = np.atleast_1d(ccf["ccf_keys"])
        if (fmt == "multipoles" and len(ccf_keys) < 2) or (fmt == "rmu" and len(ccf_keys) != 3):
            raise InputError(f"Wrong number of redshift-space ccf keys provided for format {fmt}")
        for key in ccf_keys:
            if key not in input_data:
                raise InputError(f"Key {key} not found in file {input_fn}")
        if fmt != "multipoles":
            raise InputError("Currently only multipole format is supported for redshift-space ccf data and covmat")
        self.s = np.asarray(input_data[ccf_keys[0]], dtype=float)
        names = ["monopole", "quadrupole", "hexadecapole"][: len(ccf_keys) - 1]
        self.poles_s = np.atleast_1d([0, 2, 4][: len(ccf_keys) - 1])
        self.redshift_multipoles = {}
        for i, ell in enumerate(self.poles_s):
            a = np.asarray(input_data[ccf_keys[i + 1]], dtype=float)
            self.redshift_multipoles[f"{ell}"] = a if isim is None else a[isim]
        want = self.s.shape if self.fixed_data else (len(self.beta_ccf), len(self.s))
        for i, ell in enumerate(self.poles_s):
            got = self.redshift_multipoles[f"{ell}"].shape
            if got != want:
                raise InputError(f"Shape of redshift ccf {names[i]} is {got}, expected {want}")

    def _load_covariance_matrix(self, covariance, input_fn):
        """Covariance and its inverse (reference: ccf_fit.py:116-164)."""
        input_data = utils.read_input_file(input_fn, self.extensions)
        if not self.fixed_data:
            self.fixed_covmat = covariance.get("fixed_beta", True)
            if not self.fixed_covmat:
                beta_key = covariance.get("beta_key", None)
                if beta_key and beta_key in input_data:
                    self.beta_covmat = np.asarray(input_data[beta_key], dtype=float)
                    if not np.all(np.diff(self.beta_covmat) > 0):
                        raise InputError("Covariance beta grid must be strictly monotonically increasing")
                else:
                    self.beta_covmat = self.beta_ccf
        else:
            self.fixed_covmat = True
        cov_key = covariance["cov_key"]
        if cov_key not in input_data:
            raise InputError(f"Key {cov_key} not found in file {input_fn}")
        covmat = np.asarray(input_data[cov_key], dtype=float)
        n = len(self.s) * len(self.poles_s)
        if self.fixed_covmat:
            if covmat.shape != (n, n):
                raise InputError("Unexpected shape of (fixed) covariance matrix")
        elif covmat.shape != (len(self.beta_covmat), n, n):
            raise InputError("Unexpected shape of (beta-varying) covariance matrix")
        self.covmat = covmat
        self.icov = np.linalg.inv(self.covmat)

    # ------------------------------------------------------------------ host-side accessors ---
    def get_interpolated_redshift_multipoles(self, beta=None):
        """Data multipoles at ``beta`` (reference: ccf_fit.py:166-193)."""
        stack = np.array([self.redshift_multipoles[f"{ell}"] for ell in self.poles_s])
        if self.fixed_data:
            return np.atleast_2d(stack)
        if beta is None:
            raise InputError("Need to supply a valid value of beta for interpolation")
        return np.atleast_2d(T.pchip(self.beta_ccf, np.moveaxis(stack, 1, 0))(beta))

    def _bracket(self, beta):
        """(low index, weight of the LAST grid entry) used for covariance and precision
        (reference: ccf_fit.py:213-228 - note the upper bracket is the last grid point)."""
        g = self.beta_covmat
        if beta < g.min():
            return 0, 0.0
        if beta > g.max():
            return len(g) - 1, 0.0
        if beta in g:
            return int(np.where(g == beta)[0][0]), 0.0
        lo = int(np.where(g < beta)[0][-1])
        hi = int(np.where(g >= beta)[0][-1])
        return lo, (beta - g[lo]) / (g[hi] - g[lo])

    def _interp_stack(self, stack, beta):
        if self.fixed_covmat:
            return stack
        if beta is None:
            raise InputError("Need to supply a valid value of beta for interpolation")
        lo, t = self._bracket(beta)
        if t == 0.0:
            return stack[lo]
        return (1 - t) * stack[lo] + t * stack[-1]

    def get_interpolated_covariance(self, beta=None):
        """Reference: ccf_fit.py:195-228."""
        return self._interp_stack(self.covmat, beta)

    def get_interpolated_precision(self, beta=None):
        """Reference: ccf_fit.py:230-260."""
        return self._interp_stack(self.icov, beta)

    def correlation_matrix(self, beta=None):
        """Reference: ccf_fit.py:262-284."""
        cov = self.get_interpolated_covariance(beta)
        d = np.sqrt(np.diag(cov))
        denom = np.outer(d, d)
        out = np.zeros_like(cov)
        np.divide(cov, denom, out=out, where=denom != 0)
        return out

    def diagonal_errors(self, beta=None):
        """Reference: ccf_fit.py:286-304."""
        cov = self.get_interpolated_covariance(beta)
        return np.sqrt(np.diag(cov)).reshape((len(self.poles_s), len(self.s)))

    def multipole_datavector(self, beta=None):
        """Reference: ccf_fit.py:306-323."""
        return self.get_interpolated_redshift_multipoles(beta).reshape(len(self.poles_s) * len(self.s))

    # ------------------------------------------------------------------ device plumbing -------
    def _fit_side(self):
        return self

    def _fit_rows(self, params, model):
        need_beta = self._needs_beta(model) or not self.fixed_data
        if not isinstance(params, np.ndarray) and not self.fixed_data and params.get("beta", None) is None:
            raise InputError("Need to supply a valid value of beta for interpolation")   # ccf_fit.py:188-189
        return self._param_rows(params, need_beta, self._needs_fsigma8(model))

    def _merged_fit(self, kwargs):
        fit_options = dict(self.fit_options)
        fit_options.update(kwargs)
        return fit_options

    def _run(self, params, kwargs, want_theory=False):
        model = self._merged(kwargs)
        self._check_supported(model)
        fit_options = self._merged_fit(kwargs)
        eng = self._get_engine(self._engine_key(model), model["simpson_even"])
        opts = eng.make_opts(model, fit_options)
        rows = self._fit_rows(params, model)
        if fit_options["beta_interpolation"] == "likelihood" and not self.fixed_data:
            return self._run_likelihood_interp(eng, opts, rows)
        return eng.eval_batch(opts, rows, want_theory=want_theory)

    def _run_likelihood_interp(self, eng, opts, rows):
        """beta_interpolation='likelihood' (reference: ccf_fit.py:383-440): evaluate at the two grid betas that
        bracket the input and blend lnL and chi2 linearly."""
        g = self.beta_ccf
        beta = rows[:, N.P_BETA]
        lo = np.array([np.where(g < b)[0][-1] for b in beta])       # IndexError outside the grid, as the reference
        hi = np.array([np.where(g >= b)[0][0] for b in beta])
        t = (beta - g[lo]) / (g[hi] - g[lo])
        both = np.concatenate([rows, rows])
        both[: len(rows), N.P_BETA] = g[lo]
        both[len(rows):, N.P_BETA] = g[hi]
        lnl, chi2, _ = eng.eval_batch(opts, both)
        n = len(rows)
        bad = ~np.isfinite(lnl[:n]) | ~np.isfinite(lnl[n:])          # singular at either end fails both (:402-410)
        out_l = (1 - t) * lnl[:n] + t * lnl[n:]
        out_c = (1 - t) * chi2[:n] + t * chi2[n:]
        out_l[bad] = -np.inf
        out_c[bad] = np.inf
        return out_l, out_c, None

    # ------------------------------------------------------------------ likelihood (device) ---
    def chi_squared(self, params, **kwargs):
        """chi-square of the theory against the data at one point; also returns the covariance used
        (reference: ccf_fit.py:325-354)."""
        kw = dict(kwargs)
        kw["beta_interpolation"] = "datavector"
        _, chi2, _ = self._run(params, kw)
        cov = self.get_interpolated_covariance(params.get("beta", None))
        return float(chi2[0]), cov

    def _single_point_plan(self):
        """(engine, opts, need_beta, need_fsigma8) of a plain ``log_likelihood(params)`` call, cached until the option
        dictionaries are changed (they are public attributes in the reference and consulted on every call there)."""
        plan = getattr(self, "_plan", None)
        if plan is not None and plan[0] == self.model and plan[1] == self.fit_options:
            return plan[2]
        model = self._merged({})
        self._check_supported(model)
        fit_options = self._merged_fit({})
        if fit_options["beta_interpolation"] == "likelihood" and not self.fixed_data:
            made = None                                         # two evaluations per point: the general path
        elif self._broker_spec and (self.model, self.fit_options) == self._broker_pristine:
            # the options are the ones the configuration blocks gave - what the job's broker evaluates: the point goes to its
            # mailbox and this process never creates a GPU context for it (victor_amd/broker.py)
            if self._broker_client is None:
                from . import broker as B
                self._broker_client = B.connect(*self._broker_blocks, self._broker_spec)
            made = (None, None, self._needs_beta(model) or not self.fixed_data, self._needs_fsigma8(model), None)
        else:
            eng = self._get_engine(self._engine_key(model), model["simpson_even"])
            opts = eng.make_opts(model, fit_options)
            made = (eng, C.byref(opts), self._needs_beta(model) or not self.fixed_data, self._needs_fsigma8(model), opts)
        import copy
        self._plan = (copy.deepcopy(self.model), copy.deepcopy(self.fit_options), made)
        return made

    def log_likelihood(self, params, **kwargs):
        """(lnL, chi2) at one parameter point (reference: ccf_fit.py:356-483)."""
        plan = self._single_point_plan() if (not kwargs and type(params) is dict) else None
        row = None
        if plan is not None:
            if not self.fixed_data and params.get("beta", None) is None:
                raise InputError("Need to supply a valid value of beta for interpolation")   # ccf_fit.py:188-189
            try:
                # one point given as scalars (cobaya hands over all ~25 declared inputs per call: only the ones that enter the row
                # are looked at); an array-valued entry fails the float() inside and takes the general path
                row = self._scalar_row(params, plan[2], plan[3])
            except TypeError:
                row = None
        if row is not None:
            eng = plan[0]
            lnl, chi2 = eng.eval_point(plan[1], row) if eng is not None else self._broker_client.eval_point(row)
        else:
            lnl, chi2, _ = self._run(params, kwargs)
            lnl, chi2 = float(lnl[0]), float(chi2[0])
        if lnl == -np.inf and chi2 == np.inf:
            print(f"Likelihood evaluation failed (singular covariance or NaN). Parameters at fail point: {params}")
        return lnl, chi2

    def log_likelihood_batch(self, params, **kwargs):
        """(lnL[n], chi2[n]) for a batch: ``params`` is a dict of equal-length arrays (scalars broadcast) or an
        ``(n, VK_NPAR)`` array of rows in the column order of ``include/victor_hip.h``."""
        plan = self._single_point_plan() if not kwargs else None
        if plan is not None and plan[0] is not None:
            # plain call (a sampler's step): the cached (engine, option block) pair of log_likelihood, no option merging
            eng, _, need_beta, need_fs8, opts = plan
            if not isinstance(params, np.ndarray) and not self.fixed_data and params.get("beta", None) is None:
                raise InputError("Need to supply a valid value of beta for interpolation")   # ccf_fit.py:188-189
            lnl, chi2, _ = eng.eval_batch(opts, self._param_rows(params, need_beta, need_fs8))
            return lnl, chi2
        lnl, chi2, _ = self._run(params, kwargs)
        return lnl, chi2

    def theory_vector_batch(self, params, **kwargs):
        """Theory vectors (n, N) on the data's own s grid and multipoles."""
        model = self._merged(kwargs)
        self._check_supported(model)
        eng = self._get_engine(self._engine_key(model), model["simpson_even"])
        return eng.theory_vector_batch(eng.make_opts(model), self._fit_rows(params, model))
