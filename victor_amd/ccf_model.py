"""``CCFModel``: redshift-space CCF theory, evaluated on the GPU.

Drop-in for ``victor.CCFModel`` (reference: ``victor/ccf_model.py:24-860``) with the same
constructor, option dictionary, attributes and method signatures.  Construction happens on the
host (tables are read and every spline is compiled to coefficient arrays); every evaluation -
``theory_xi``, ``theory_multipoles``, ``theory_multipole_vector`` and their ``*_batch`` forms - runs in
the HIP kernels behind ``libvictor_hip.so``.  Option combinations the kernels do not implement raise
:class:`InputError`; nothing is ever computed on the CPU instead.
"""

import os

import numpy as np

from . import _native as N
from . import tables as T
from . import utils
from . import velocity_tables
from .cosmology import BackgroundCosmology
from .utils import InputError

EXTENSIONS = {"npy": [".npy"],
              "hdf5": [".hdf", ".h4", ".hdf4", ".he2", ".h5", ".hdf5", ".he5", ".h5py"]}


def _ez(z, cosmology):
    """E(z) of a LambdaCDM background without radiation (reference: cosmology.py:27-45)."""
    return BackgroundCosmology(cosmology).Ez(z)


_SCALARS = (float, int, np.float64, np.int64)


class CCFModel:
    """Model calculations for void-galaxy / density-split cross-correlation functions."""

    def __init__(self, model, device=0):
        self.z_eff = model["z_eff"]
        self.iaH = (1 + self.z_eff) / (100 * _ez(self.z_eff, model.get("cosmology")))   # ccf_model.py:43-45
        self.extensions = EXTENSIONS
        self._device = device
        self._engine = None
        self._unsupported = []

        base_dir = model.get("dir", "")
        input_fn = os.path.join(base_dir, model["input_model_data_file"])
        if not os.path.isfile(input_fn):
            raise InputError(f"File {input_fn} containing input model data not found")
        input_data = utils.read_input_file(input_fn, self.extensions)

        self._load_realspace_ccf(model["realspace_ccf"], input_data)
        self.matter_model = model["matter_ccf"].get("model", "linear_bias")
        self.realspace_ccf_from_data = model["realspace_ccf"].get("from_data", False)
        self.template_sigma8 = model["matter_ccf"].get("template_sigma8", None)
        if self.matter_model == "linear_bias" and not self.realspace_ccf_from_data and not self.template_sigma8:
            raise InputError("When using linear bias for the matter ccf and the real-space ccf is from a template, "
                             "template_sigma8 must be provided")
        if self.matter_model == "template":
            self._set_matter_ccf_template(model["matter_ccf"], input_data)
        self._set_velocity_pdf(model["velocity_pdf"], input_data)
        del input_data
        # defaults used in automated evaluations; any key can be overridden per call (ccf_model.py:85-97)
        self.model = {
            "rsd_model": model.get("rsd_model", "streaming"),
            "kaiser_approximation": model.get("kaiser_approximation", False),
            "kaiser_coord_shift": model.get("kaiser_coord_shift", True),
            "assume_isotropic": model["realspace_ccf"].get("assume_isotropic", True),
            "realspace_ccf_from_data": self.realspace_ccf_from_data,
            "matter_model": self.matter_model,
            "excursion_set_options": model["matter_ccf"].get("excursion_set_options", {}),
            "bias": model["matter_ccf"].get("bias", 1.9),
            "mean_model": model["velocity_pdf"]["mean"].get("model", "linear"),
            "pdf_form": model["velocity_pdf"].get("form", "gaussian"),
            "empirical_corr": model["velocity_pdf"]["mean"].get("empirical_corr", False),
            "velocity_independent_of_AP": model["velocity_pdf"].get("rescale_templates_independent_of_AP", True),
            # not a reference key: which SciPy's even-N Simpson rule the velocity integral of ccf_model.py:690 follows
            # ('simpson' = SciPy >= 1.11, the default; 'avg' = SciPy < 1.11) - see tables.simpson_weights, README.md
            "simpson_even": self._simpson_rule((model.get("numerics") or {}).get("simpson_even")),
        }

    # ------------------------------------------------------------------ set-up (host) -----
    def _load_realspace_ccf(self, realspace_ccf, input_data):
        """Read the real-space CCF multipoles (reference: ccf_model.py:99-181)."""
        fmt = realspace_ccf.get("format", "multipoles")
        self.fixed_real_input = not realspace_ccf.get("reconstruction", False)
        ccf_keys = np.atleast_1d(realspace_ccf["ccf_keys"])
        if not self.fixed_real_input:
            beta_key = realspace_ccf.get("beta_key", None)
            if beta_key is None:
                raise InputError("Reconstruction specified for realspace ccf but no beta key provided")
            if beta_key not in input_data:
                raise InputError(f"Key {beta_key} not found in input model data file")
            self.beta = np.asarray(input_data[beta_key], dtype=float)
            if not np.all(np.diff(self.beta) > 0):
                raise InputError("Realspace beta grid must be strictly monotonically increasing")
        if (fmt == "multipoles" and len(ccf_keys) < 2) or (fmt == "rmu" and len(ccf_keys) != 3):
            raise InputError(f"Wrong number of ccf keys provided for ccf format {fmt}")
        for key in ccf_keys:
            if key not in input_data:
                raise InputError(f"Key {key} not found in input model data file")
        isim = realspace_ccf.get("simulation_number", None)
        if isim is not None and not isinstance(isim, int):
            raise InputError("If provided, simulation_number must be an integer")

        def select(a):
            a = np.asarray(a, dtype=float)
            return a if isim is None else a[isim]

        if fmt == "multipoles":
            self.r = np.asarray(input_data[ccf_keys[0]], dtype=float)
            names = ["monopole", "quadrupole", "hexadecapole"][: len(ccf_keys) - 1]
            self.poles_r = np.atleast_1d([0, 2, 4][: len(ccf_keys) - 1])
            self.real_multipoles = {}
            for i, ell in enumerate(self.poles_r):
                self.real_multipoles[f"{ell}"] = select(input_data[ccf_keys[i + 1]])
            want = self.r.shape if self.fixed_real_input else (len(self.beta), len(self.r))
            for i, ell in enumerate(self.poles_r):
                got = self.real_multipoles[f"{ell}"].shape
                if got != want:
                    raise InputError(f"Shape of real ccf {names[i]} is {got}, expected {want}")
        elif fmt == "rmu":
            # xi(r, mu) tabulated: bilinear interpolant projected onto l = 0, 2, 4 at the r nodes
            self.r = np.asarray(input_data[ccf_keys[0]], dtype=float)
            mu = np.asarray(input_data[ccf_keys[1]], dtype=float)
            real_ccf = select(input_data[ccf_keys[2]])
            self.poles_r = np.array([0, 2, 4])

            def project(table):
                return utils.multipoles_from_fn(utils.bilinear_on_grid(self.r, mu, table.T), self.r, self.poles_r)

            if self.fixed_real_input:
                if real_ccf.shape != (len(self.r), len(mu)):
                    raise InputError(f"Shape of real ccf is {real_ccf.shape}, expected ({len(self.r)}, {len(mu)})")
                self.real_multipoles = project(real_ccf)
            else:
                if real_ccf.shape != (len(self.beta), len(self.r), len(mu)):
                    raise InputError(f"Shape of real ccf is {real_ccf.shape}, expected "
                                     f"({len(self.beta)}, {len(self.r)}, {len(mu)})")
                self.real_multipoles = {f"{l}": np.zeros((len(self.beta), len(self.r))) for l in self.poles_r}
                for i in range(len(self.beta)):
                    tmp = project(real_ccf[i])
                    for l in self.poles_r:
                        self.real_multipoles[f"{l}"][i] = tmp[f"{l}"]
        else:
            raise InputError(f"Wrong number of ccf keys provided for ccf format {fmt}")
        if len(self.r) < 4:
            raise InputError("Real-space ccf needs at least 4 radial bins for cubic interpolation")

    def _set_matter_ccf_template(self, matter_ccf, input_data):
        """delta(r) and its volume average Delta(r) (reference: ccf_model.py:183-220)."""
        self.template_sigma8 = matter_ccf.get("template_sigma8", None)
        if not self.template_sigma8:
            raise InputError("When using template model for the matter ccf, template_sigma8 must be provided")
        template_keys = np.atleast_1d(matter_ccf.get("template_keys"))
        if len(template_keys) != 2:
            raise InputError("Wrong number of matter ccf template keys provided: expected 2 "
                             "(radial distance and monopole)")
        for key in template_keys:
            if key not in input_data:
                raise InputError(f"Key {key} not found in input model data file")
        r_for_delta = np.asarray(input_data[template_keys[0]], dtype=float)
        delta = np.asarray(input_data[template_keys[1]], dtype=float)
        if len(r_for_delta) != len(delta):
            raise InputError(f"Shape of matter ccf template is {len(delta)}, expected {len(r_for_delta)}")
        r = np.linspace(r_for_delta.min(), r_for_delta.max())
        if matter_ccf.get("integrated", False):
            self.integrated_delta = T.notaknot(r_for_delta, delta)
            derivative = np.gradient(self.integrated_delta(r), r)
            self.delta = T.notaknot(r, self.integrated_delta(r) + r * derivative / 3)
        else:
            from scipy.integrate import quad
            self.delta = T.notaknot(r_for_delta, delta)
            integral = np.zeros_like(r)
            dfun = self.delta.scalar_function()
            for i in range(len(r)):   # same adaptive quadrature as the reference so Delta agrees to round-off
                integral[i] = quad(lambda x: 3 * dfun(x) * x ** 2 / r[i] ** 3, 0, r[i], full_output=1)[0]
            self.integrated_delta = T.notaknot(r, integral)

    def _set_velocity_pdf(self, velocity_pdf, input_data):
        """Mean and dispersion of the Gaussian velocity pdf (reference: ccf_model.py:222-297)."""
        mean_model = velocity_pdf["mean"].get("model", "linear")
        if mean_model == "template":
            self.template_fsigma8 = velocity_pdf["mean"].get("template_fsigma8")
            if not self.template_fsigma8:
                raise InputError("When using template model for the mean of the velocity pdf, a value for "
                                 "template_fsigma8 must be provided")
            z_sim = velocity_pdf["mean"].get("z_sim", None)
            ratio = velocity_pdf["mean"].get("template_hubble_ratio", None)
            self.z_sim = self.z_eff if z_sim is None else z_sim
            self.template_hubble_ratio = 1 if ratio is None else ratio
            template_keys = np.atleast_1d(velocity_pdf["mean"].get("template_keys"))
            if len(template_keys) != 2:
                raise InputError(f"{len(template_keys)} velocity mean template keys provided, require 2")
            for key in template_keys:
                if key not in input_data:
                    raise InputError(f"Key {key} not found in input model data file")
            r_for_v = np.asarray(input_data[template_keys[0]], dtype=float)
            vr = np.asarray(input_data[template_keys[1]], dtype=float)
            if len(r_for_v) != len(vr):
                raise InputError(f"Shape of mean velocity template is {len(vr)}, expected {len(r_for_v)}")
            self.radial_velocity = T.notaknot(r_for_v, vr)
            self.has_velocity_template = True
        else:
            self.has_velocity_template = False
        if mean_model == "nonlinear" and not self.matter_model == "excursion_set":
            raise InputError("Cannot have nonlinear mean velocity model unless using excursion_set matter model")

        dispersion = velocity_pdf.get("dispersion", {})
        disp_model = dispersion.get("model", "constant")
        if disp_model == "template":
            template_keys = np.atleast_1d(dispersion.get("template_keys"))
            if len(template_keys) < 2 or len(template_keys) > 3:
                raise InputError(f"{len(template_keys)} velocity dispersion template keys provided, require 2 or 3")
            for key in template_keys:
                if key not in input_data:
                    raise InputError(f"Key {key} not found in input model data file")
            self.r_for_sv = np.asarray(input_data[template_keys[0]], dtype=float)
            sv = np.asarray(input_data[template_keys[-1]], dtype=float)
            self._sv_isotropic = len(template_keys) == 2
            if self._sv_isotropic:
                self.mu_for_sv = np.linspace(0, 1)
                sv = (np.ones((len(self.mu_for_sv), len(self.r_for_sv))) * sv).T
            else:
                self.mu_for_sv = np.asarray(input_data[template_keys[1]], dtype=float)
            if sv.shape != (len(self.r_for_sv), len(self.mu_for_sv)):
                raise InputError(f"Dispersion template shape {sv.shape} does not match expected "
                                 f"({len(self.r_for_sv)}, {len(self.mu_for_sv)})")
            if dispersion.get("filter", True):
                from scipy.signal import savgol_filter
                window = dispersion.get("filter_window", 3)
                polyorder = dispersion.get("filter_order", 1)
                sv = np.array([savgol_filter(sv[:, i], window, polyorder) for i in range(sv.shape[1])]).T
        elif disp_model == "constant":
            # The documented default, but the reference cannot run it: it sets self.sv_rmu and then reads an
            # undefined local (ccf_model.py:284-291 -> UnboundLocalError).  There is no reference behaviour to
            # reproduce, so refuse it explicitly.
            raise InputError("dispersion model 'constant' fails inside the reference (ccf_model.py:284-291); "
                             "supply a dispersion template instead")
        else:
            raise InputError(f"Bad choice '{disp_model}' for dispersion model, options are 'constant' or 'template'")
        if sv.shape[0] == len(self.r_for_sv):
            sv = sv.T                                                     # (n_mu, n_r)
        # normalise by the monopole at the largest r (ccf_model.py:295-297; interp2d default = bilinear)
        mono = utils.multipoles_from_fn(utils.bilinear_on_grid(self.r_for_sv, self.mu_for_sv, sv),
                                        self.r_for_sv[-1:], ell=[0])
        self.sv_rmu = sv / mono["0"][-1]
        if not self._sv_isotropic and len(self.mu_for_sv) < 4:
            raise InputError("Anisotropic dispersion template needs at least 4 mu values for cubic interpolation")
        if len(self.r_for_sv) < 4:
            raise InputError("Dispersion template needs at least 4 radial bins for cubic interpolation")

    @staticmethod
    def _simpson_rule(name):
        try:
            return T.simpson_even_rule(name)
        except ValueError as exc:
            raise InputError(str(exc))

    # ------------------------------------------------------------------ small host helpers ---
    def get_interpolated_real_multipoles(self, beta=None):
        """Real-space multipoles at ``beta`` (reference: ccf_model.py:299-326).  Host helper for inspection;
        the kernels evaluate the same PCHIP pieces on the device."""
        stack = np.array([self.real_multipoles[f"{ell}"] for ell in self.poles_r])
        if self.fixed_real_input:
            return np.atleast_2d(stack)
        if beta is None:
            raise InputError("Need to supply a valid value of beta for interpolation")
        return np.atleast_2d(T.pchip(self.beta, np.moveaxis(stack, 1, 0))(beta))

    # ------------------------------------------------------------------ matter / velocity profiles ---
    def _matter_nodal(self, matter_model, r_nodes, beta=None):
        return velocity_tables.matter_nodal(self, matter_model, r_nodes, beta)

    def delta_profiles(self, r, params, **kwargs):
        """delta(r) and its volume average Delta(r) (reference: ccf_model.py:328-383).  Host-side accessor (used
        for plots and set-up); the likelihood kernels use the precompiled tables instead."""
        model = self._merged(kwargs)
        r = np.asarray(r, dtype=float)
        d, D = self._matter_nodal(model["matter_model"], r, params.get("beta", None))
        if model["matter_model"] == "linear_bias":
            bias = params.get("bias", model["bias"])
            return d / bias, D / bias
        return d, D

    def velocity_terms(self, r, params, **kwargs):
        """Mean radial velocity v_r(r) and its derivative (reference: ccf_model.py:385-492).  Host-side accessor
        for the linear mean model; the kernels evaluate the same profile from tables with per-point amplitudes."""
        model = self._merged(kwargs)
        self._check_supported(model)
        r = np.asarray(r, dtype=float)
        if "epsilon" in params:
            apar = params.get("alpha", 1) * params["epsilon"] ** (-2 / 3)
        else:
            apar = params.get("apar", 1)
        iaH_true = self.iaH * apar
        delta_r, int_delta_r = self.delta_profiles(r, params, **kwargs)
        if model["mean_model"] == "template":
            growth = (params["fsigma8"] / self.template_fsigma8) * self.template_hubble_ratio * \
                (1 + self.z_sim) / (1 + self.z_eff) / apar
            rg = np.linspace(0.1, self.r.max(), 100)
            vr = self.radial_velocity(r) * growth
            return vr, T.notaknot(rg, np.gradient(self.radial_velocity(rg) * growth, rg))(r)
        if model["matter_model"] == "linear_bias" and model["realspace_ccf_from_data"]:
            growth = params["beta"] * params.get("bias", model["bias"])
        else:
            growth = params["fsigma8"] / self.template_sigma8
        if not model["empirical_corr"]:
            vr = -growth * r * int_delta_r / (3 * iaH_true)
            dvr = -growth * (delta_r - 2 * int_delta_r / 3) / iaH_true
        else:
            Av = params.get("Av", 0)
            vr = -growth * r * int_delta_r * (1 + Av * delta_r) / (3 * iaH_true)
            rg = np.linspace(0.1, self.r.max(), 100)
            Ds = T.notaknot(r, int_delta_r)(rg)
            ds = T.notaknot(r, delta_r)(rg)
            vg = -growth * rg * Ds * (1 + Av * ds) / (3 * iaH_true)
            dvr = T.notaknot(rg, np.gradient(vg, rg))(r)
        return vr, dvr

    # ------------------------------------------------------------------ device plumbing -------
    def _merged(self, kwargs):
        model = dict(self.model)
        model.update(kwargs)
        return model

    def _get_engine(self, matter_model=None, simpson_even=None):
        """One device context per (matter model, Simpson convention) - their tables differ; created on first use."""
        matter_model = matter_model or self.matter_model
        rule = self._simpson_rule(simpson_even if simpson_even is not None else self.model["simpson_even"])
        if self._engine is None:
            self._engine = {}
        key = matter_model if rule == T.SIMPSON_EVEN_DEFAULT else (matter_model, rule)
        if key not in self._engine:
            from .engine import Engine
            self._engine[key] = Engine(self, self._fit_side(), device=self._device, matter_model=matter_model,
                                       simpson_even=rule, lib=getattr(self, "_native_lib", None))
        return self._engine[key]

    def _fit_side(self):
        return None

    def _check_supported(self, model):
        problems = list(self._unsupported)
        if model["matter_model"] == "excursion_set":
            problems.append("matter_model 'excursion_set'")
        elif model["matter_model"] not in ("template", "linear_bias"):
            raise InputError(f"Invalid choice of matter_model {model['matter_model']}")
        if model["matter_model"] == "template" and not hasattr(self, "delta"):
            raise InputError("matter_model 'template' requested but no matter ccf template was loaded")
        if model["matter_model"] == "linear_bias" and not model["realspace_ccf_from_data"] and not self.template_sigma8:
            raise InputError("template_sigma8 must be provided for the linear_bias matter model")
        if model["mean_model"] == "template":
            if not self.has_velocity_template:
                raise InputError("velocity_terms: Cannot use template option as no template has been supplied.")
        elif model["mean_model"] != "linear":
            problems.append(f"velocity mean model '{model['mean_model']}'")
        if model["rsd_model"] not in N.RSD:
            raise InputError(f"theory_xi: Unrecognised choice of model {model['rsd_model']}")
        if problems:
            raise InputError("not implemented on the HIP path (and there is no CPU fallback): " + ", ".join(problems))

    def _param_rows(self, params, need_beta, need_fsigma8=True):
        """dict of scalars or equal-length arrays -> (n, VK_NPAR) rows (reference: ccf_model.py:583-613,638)."""
        if isinstance(params, np.ndarray):
            rows = N.f64(params)
            if rows.ndim != 2 or rows.shape[1] != N.VK_NPAR:
                raise InputError(f"parameter array must have shape (n, {N.VK_NPAR})")
            return rows
        arrays, scalars = {}, {}
        for key, v in params.items():                 # one pass; np.ndim only for what is neither a number nor an ndarray
            if type(v) in _SCALARS or v is None:
                scalars[key] = v
            elif (v.ndim if isinstance(v, np.ndarray) else np.ndim(v)) > 0:
                arrays[key] = v
            else:
                scalars[key] = v
        lengths = {len(v) for v in arrays.values()}
        if len(lengths) > 1:
            raise InputError(f"parameter arrays have different lengths: {sorted(lengths)}")
        n = next(iter(lengths)) if arrays else 1      # an empty batch (length 0) is legal
        rows = np.empty((n, N.VK_NPAR))

        if not arrays:
            # one point given as scalars (the reference's calling convention, one call per MCMC step): plain Python
            # floats, one assignment into the row
            rows[0] = self._scalar_row(params, need_beta, need_fsigma8)
            return rows

        # Batches: the scalar entries (fixed parameters of a sampler, defaults) fill every row at once from the same Python
        # floats the single-point path uses; only the parameters given as arrays are assigned column by column (a walker
        # ensemble's step spends as long here as on the GPU, so the numpy calls are counted)

        def col(name, default=None):                  # the parameter as a float or a length-n float array
            if name in arrays:
                v = np.asarray(arrays[name], dtype=float)
                if v.ndim > 1:
                    raise InputError("parameter arrays must be one-dimensional")
                return v
            v = scalars.get(name, default)
            return None if v is None else float(v)

        if need_fsigma8 and "fsigma8" not in params:
            raise KeyError("fsigma8")                                # as ccf_model.py:432-435
        if need_beta and "beta" not in params:
            raise KeyError("beta")                                   # as ccf_model.py:587
        if "epsilon" in params:
            eps = col("epsilon")
            if isinstance(eps, float):
                apar = col("alpha", 1) * eps ** (-2 / 3)
                aperp = eps * apar
            else:                                  # arrays: the library's routine (libm's pow, as the scalar path and the reference)
                alpha = col("alpha", 1)
                if isinstance(alpha, float):
                    aperp, apar = N.epsilon_to_ap(eps, alpha)
                else:
                    aperp, apar = N.epsilon_to_ap(eps, 1.0)
                    apar = alpha * apar
                    aperp = eps * apar
        else:
            aperp = col("aperp", 1)
            apar = col("apar", 1)
            eps = aperp / apar
        beta = col("beta")
        cols = (col("fsigma8", 0.0), col("sigma_v", 380), aperp, apar, eps, 0.40 if beta is None else beta, col("astar", 1),
                col("M", 1.0), col("Q", 1.0), col("bias", self.model["bias"]), col("Av", 0), 0.0)
        rows[:] = [c if isinstance(c, float) else 0.0 for c in cols]
        for k, c in enumerate(cols):
            if not isinstance(c, float):
                rows[:, k] = c
        return rows

    def _scalar_row(self, params, need_beta, need_fsigma8=True):
        """One parameter point given as scalars -> list of VK_NPAR floats in the column order of include/victor_hip.h
        (reference: ccf_model.py:583-613, 638, 695-696)."""
        get = params.get
        fs8 = float(params["fsigma8"]) if need_fsigma8 else float(get("fsigma8", 0.0))   # KeyError as ccf_model.py:432-435
        if "epsilon" in params:
            eps = float(params["epsilon"])
            apar = float(get("alpha", 1)) * eps ** (-2 / 3)
            aperp = eps * apar
        else:
            aperp = float(get("aperp", 1))
            apar = float(get("apar", 1))
            eps = aperp / apar
        if need_beta:
            beta = float(params["beta"])                           # KeyError if absent, as ccf_model.py:587
        else:
            beta = get("beta", None)
            beta = 0.40 if beta is None else float(beta)
        return [fs8, float(get("sigma_v", 380)), aperp, apar, eps, beta, float(get("astar", 1)), float(get("M", 1.0)),
                float(get("Q", 1.0)), float(get("bias", self.model["bias"])), float(get("Av", 0)), 0.0]

    def _needs_beta(self, model):
        return not (self.fixed_real_input and model["matter_model"] != "linear_bias")

    def _needs_fsigma8(self, model):
        # growth term: beta*bias for linear_bias on a measured real-space ccf, fsigma8/sigma8 otherwise
        return not (model["matter_model"] == "linear_bias" and model["realspace_ccf_from_data"])

    def _engine_key(self, model):
        """Which table set a call needs: the velocity template replaces the matter-model based profile."""
        return "velocity_template" if model["mean_model"] == "template" else model["matter_model"]

    def _prepare(self, params, model):
        """(engine, opts, rows) for one call with merged options ``model``."""
        self._check_supported(model)
        eng = self._get_engine(self._engine_key(model), model["simpson_even"])
        opts = eng.make_opts(model)
        rows = self._param_rows(params, self._needs_beta(model), self._needs_fsigma8(model))
        return eng, opts, rows

    # ------------------------------------------------------------------ theory (device) --------
    def theory_xi(self, s, mu, params, **kwargs):
        """xi^s(s, mu) (reference: ccf_model.py:538-789).

        ``s`` and ``mu`` are 1-D arrays or a ``np.meshgrid`` pair; returns shape ``(n_mu, n_s)``.
        """
        out = self.theory_xi_batch(s, mu, params, **kwargs)
        return out[0]

    def theory_xi_batch(self, s, mu, params, **kwargs):
        model = self._merged(kwargs)
        s = np.atleast_1d(s)
        mu = np.atleast_1d(mu)
        if np.ndim(s) == 2 and np.ndim(mu) == 2:
            if s.shape != mu.shape:
                raise InputError("theory_xi: If arguments s and mu are 2D arrays they must have same shape")
            s, mu = np.unique(s), np.unique(mu)
        elif not (np.ndim(s) == 1 and np.ndim(mu) == 1):
            raise InputError("theory_xi: arguments s and mu have incompatible dimensions")
        eng, opts, rows = self._prepare(params, model)
        return eng.xi_smu_batch(opts, rows, s, mu)

    def theory_multipoles(self, s, params, poles=[0, 2], **kwargs):
        """Legendre multipoles of xi^s at ``s`` as a dict keyed '0', '2', ... (reference: ccf_model.py:791-827)."""
        poles = np.atleast_1d(poles)
        out = self.theory_multipoles_batch(s, params, poles, **kwargs)
        return {f"{ell}": out[0, i] for i, ell in enumerate(poles)}

    def theory_multipoles_batch(self, s, params, poles=[0, 2], **kwargs):
        """Batched form: returns an array of shape (n_points, n_poles, n_s)."""
        model = self._merged(kwargs)
        poles = np.atleast_1d(poles)
        if len(poles) > 3 or np.any(poles > 4) or np.any(poles < 0):
            raise InputError("at most three multipoles with 0 <= ell <= 4 are supported")
        eng, opts, rows = self._prepare(params, model)
        return eng.theory_batch(opts, rows, np.asarray(s, dtype=float), poles)

    # ------------------------------------------------------------------ 2-D model grids (notebook helpers) ----
    @staticmethod
    def _sigma_pi_grid(rmax):
        """(s_perp, s_par) nodes and the (s, mu) of every node (reference: ccf_model.py:883-888, 923-928)."""
        sperp = np.linspace(0.01, rmax)
        spar = np.linspace(-rmax, rmax)
        sigma, pi = np.meshgrid(sperp, spar)
        s = np.sqrt(sigma ** 2 + pi ** 2)
        return sperp, spar, s, pi / s

    def theory_xi_2D(self, params, rmax=85, **kwargs):
        """xi^s(s_perp, s_par) on the reference's 50 x 50 grid, as a bilinear interpolant with the call convention of
        the ``interp2d`` object the reference returns (ccf_model.py:862-894).  One launch per s_par row (the product
        grid of the row's s and mu values, diagonal kept) instead of 2500 scalar evaluations."""
        sperp, spar, s, mu = self._sigma_pi_grid(rmax)
        xi = np.empty_like(s)
        for j in range(s.shape[0]):
            xi[j] = np.diagonal(self.theory_xi(s[j], mu[j], params, **kwargs))
        return utils.GridInterpolant(sperp, spar, xi)

    def xi_2D_from_multipoles(self, params, rmax=85, **kwargs):
        """sum_l xi_l(s) P_l(mu) on the same grid from the model multipoles l = 0, 2, 4, splined in s
        (ccf_model.py:896-934)."""
        s1 = np.linspace(0.01, rmax)
        poles = self.theory_multipoles(s1, params, poles=[0, 2, 4], **kwargs)
        sperp, spar, s, mu = self._sigma_pi_grid(rmax)
        grid = np.zeros_like(s)
        for ell in (0, 2, 4):
            spline = T.notaknot(s1, poles[f"{ell}"])
            grid = grid + spline(np.clip(s, s1[0], s1[-1])) * T.legendre_values(ell, mu)
        return utils.GridInterpolant(sperp, spar, grid)

    def theory_multipole_vector(self, s, params, poles=[0, 2], **kwargs):
        """Concatenated multipoles [xi_l0(s), xi_l1(s), ...] (reference: ccf_model.py:829-860)."""
        poles = np.atleast_1d(poles)
        out = self.theory_multipoles_batch(s, params, poles, **kwargs)
        return out[0].reshape(len(poles) * len(np.atleast_1d(s)))
