"""CPU oracle: a NumPy/SciPy restatement of victor's per-step likelihood path.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; nothing under
``victor_amd/`` does, and the product path raises when the HIP library is
missing rather than falling back to anything here.

Parity status: PINNED.  The restatement is checked (tests/test_oracle_vs_reference.py,
in the development container where ``/root/reference`` exists) against the
unmodified reference run through ``oracle/ref_shim.py``, and (everywhere)
against golden vectors under ``tests/golden/`` that were produced from the
reference by ``oracle/make_golden.py``.  The reference's only published
numbers for the path - the five (chi2, lnL) pairs printed in
``notebooks/victor_usage_demo.ipynb:491-499`` - are asserted as well.

The arithmetic of the reference lives in SciPy/NumPy (``setup.py:31`` pins only
``numpy>=1.17.2, scipy>=1.6.3``; no lock file).  The oracle calls the *same*
SciPy primitives the reference calls (FITPACK ``InterpolatedUnivariateSpline``
/ ``RectBivariateSpline``, ``PchipInterpolator``, ``simpson``, ``norm.pdf``,
``legendre``, ``quad``, ``savgol_filter``) from this image's numpy 2.2 / scipy
1.15.  One primitive is SciPy-version dependent in a way that matters (2e-4 relative on the
BOSS anisotropic chi2): the reference calls ``simps`` with its default ``even=`` on 50 nodes
(ccf_model.py:690), which is the 'avg' rule in SciPy < 1.11 and the corrected Simpson rule
from 1.11 on.  Both are "the reference" under its pin, so the oracle carries both:
``simpson_even='simpson'`` (default, SciPy's own ``simpson`` of this image) and ``'avg'``
(:func:`simps_legacy`, a restatement of the documented SciPy < 1.11 rule, under which all five
published notebook pairs reproduce at printed precision).

Every function cites the reference lines (paths relative to the reference
root) it restates.  The code is written independently of the reference
source: same mathematics, own structure (explicit function arguments instead
of one large class, no plotting, no excursion-set model).
"""

import numpy as np
import scipy.interpolate as si
from scipy.integrate import quad, simpson
from scipy.special import legendre
from scipy.stats import norm

_trapz = getattr(np, "trapezoid", None) or np.trapz


class OracleInputError(Exception):
    """Bad input (the reference raises ``victor.utils.InputError``, utils.py:5)."""


def simps_legacy(y, x, axis=-1, even="avg"):
    """``scipy.integrate.simps`` as documented for SciPy < 1.11 (the SciPy behind the reference's published numbers).

    Composite Simpson for samples at (possibly unequal) abscissae ``x`` - same shape as ``y`` or 1-D along ``axis``.
    For an even number of samples the composite rule covers N-2 of the N-1 intervals and ``even`` says what happens
    to the remaining one: 'first' = Simpson on the first N-2 intervals + trapezoid on the last interval, 'last' =
    trapezoid on the first interval + Simpson on the last N-2, 'avg' (the default the reference gets,
    ccf_model.py:690) = the mean of the two.
    """
    y = np.moveaxis(np.asarray(y, dtype=float), axis, -1)
    x = np.asarray(x, dtype=float)
    x = np.moveaxis(x, axis, -1) if x.ndim == y.ndim else x
    x = np.broadcast_to(x, y.shape)

    def composite(yy, xx):          # odd number of samples: one parabola per pair of intervals
        h = np.diff(xx, axis=-1)
        ha, hb = h[..., 0::2], h[..., 1::2]
        both = ha + hb
        left = yy[..., 0:-2:2] * (2 - hb / ha)
        mid = yy[..., 1:-1:2] * (both * both / (ha * hb))
        right = yy[..., 2::2] * (2 - ha / hb)
        return np.sum(both / 6.0 * (left + mid + right), axis=-1)

    n = y.shape[-1]
    if n % 2 == 1:
        return composite(y, x)
    if even not in ("avg", "first", "last"):
        raise ValueError("even must be 'avg', 'first' or 'last'")
    total, parts = 0.0, 0
    if even in ("avg", "first"):
        total = total + composite(y[..., :-1], x[..., :-1]) + 0.5 * (x[..., -1] - x[..., -2]) * (y[..., -1] + y[..., -2])
        parts += 1
    if even in ("avg", "last"):
        total = total + composite(y[..., 1:], x[..., 1:]) + 0.5 * (x[..., 1] - x[..., 0]) * (y[..., 0] + y[..., 1])
        parts += 1
    return total / parts


def velocity_integral(y, x, axis, rule):
    """The ``simps(..., x=v_par, axis=2)`` of ccf_model.py:690 under the named SciPy convention."""
    if rule in ("simpson", "scipy>=1.11"):
        return simpson(y, x=x, axis=axis)
    if rule in ("avg", "scipy<1.11"):
        return simps_legacy(y, x, axis=axis, even="avg")
    if rule in ("first", "last"):
        return simps_legacy(y, x, axis=axis, even=rule)
    raise OracleInputError(f"unknown simpson_even rule '{rule}'")


def _ius(x, y):
    # ccf_model.py:17 - every 1-D spline is InterpolatedUnivariateSpline(k=3) with ext=3
    return si.InterpolatedUnivariateSpline(x, y, ext=3)


# --------------------------------------------------------------------------- #
# file input
# --------------------------------------------------------------------------- #
def load_input(path):
    """ccf_model.py:54-68 / ccf_fit.py:48-57: ``.npy`` pickled dict of arrays."""
    if path.endswith(".npy"):
        return np.load(path, allow_pickle=True).item()
    raise OracleInputError("oracle reads .npy dict inputs only (got %s)" % path)


def inverse_aH(z_eff, cosmology):
    """cosmology.py:16-45 with astropy's Tcmb0=0 default; ccf_model.py:43-45."""
    cosmology = cosmology or {}
    om = cosmology.get("Omega_m", 0.31)
    ok = cosmology.get("Omega_K", 0)
    ol = 1 - om - ok
    ez = np.sqrt(om * (1 + z_eff) ** 3 + ok * (1 + z_eff) ** 2 + ol)
    return (1 + z_eff) / (100 * ez)


def multipoles_from_fn(fn, r, ell, even=True, npts=200):
    """utils.py:9-58: (2l+1) * trapz_mu f(r_j, mu) P_l(mu) on 200 points."""
    ell = np.atleast_1d(ell)
    out = {f"{l}": np.zeros(len(r)) for l in ell}
    if even:
        mu = np.linspace(0.0, 1.0, npts)
        factors = [2 * l + 1 for l in ell]
    else:
        mu = np.linspace(-1, 1, npts)
        factors = [(2 * l + 1) / 2 for l in ell]
    for i, l in enumerate(ell):
        lmu = legendre(l)(mu)
        for j in range(len(r)):
            y = fn(r[j], mu)
            out[f"{l}"][j] = factors[i] * _trapz(y * lmu, mu)
    return out


# --------------------------------------------------------------------------- #
class OracleModel:
    """Restates ``CCFModel`` (ccf_model.py:24-860) for the in-scope options."""

    def __init__(self, model, input_data=None):
        self.z_eff = model["z_eff"]
        self.iaH = inverse_aH(self.z_eff, model.get("cosmology"))
        if input_data is None:
            import os
            input_data = load_input(os.path.join(model.get("dir", ""), model["input_model_data_file"]))
        self._load_realspace(model["realspace_ccf"], input_data)
        self.matter_model = model["matter_ccf"].get("model", "linear_bias")
        self.from_data = model["realspace_ccf"].get("from_data", False)
        self.template_sigma8 = model["matter_ccf"].get("template_sigma8", None)
        if self.matter_model == "linear_bias" and not self.from_data and not self.template_sigma8:
            raise OracleInputError("template_sigma8 required")  # ccf_model.py:73-77
        if self.matter_model == "template":
            self._set_matter_template(model["matter_ccf"], input_data)
        elif self.matter_model != "linear_bias":
            raise OracleInputError("oracle covers matter_model template / linear_bias only")
        self._set_velocity_pdf(model["velocity_pdf"], input_data)
        # ccf_model.py:85-97
        self.model = {
            "rsd_model": model.get("rsd_model", "streaming"),
            "kaiser_approximation": model.get("kaiser_approximation", False),
            "kaiser_coord_shift": model.get("kaiser_coord_shift", True),
            "assume_isotropic": model["realspace_ccf"].get("assume_isotropic", True),
            "realspace_ccf_from_data": self.from_data,
            "matter_model": self.matter_model,
            "bias": model["matter_ccf"].get("bias", 1.9),
            "mean_model": model["velocity_pdf"]["mean"].get("model", "linear"),
            "empirical_corr": model["velocity_pdf"]["mean"].get("empirical_corr", False),
            "velocity_independent_of_AP": model["velocity_pdf"].get("rescale_templates_independent_of_AP", True),
            # which SciPy's even-N Simpson rule ccf_model.py:690 runs under (not a reference key)
            "simpson_even": (model.get("numerics") or {}).get("simpson_even", "simpson"),
        }

    # ----- init tables ------------------------------------------------------ #
    def _load_realspace(self, opts, data):
        """ccf_model.py:99-181."""
        fmt = opts.get("format", "multipoles")
        self.fixed_real_input = not opts.get("reconstruction", False)
        keys = np.atleast_1d(opts["ccf_keys"])
        if not self.fixed_real_input:
            bkey = opts.get("beta_key", None)
            if bkey is None or bkey not in data:
                raise OracleInputError("beta key missing")
            self.beta = np.asarray(data[bkey])
            if not np.all(np.diff(self.beta) > 0):
                raise OracleInputError("beta grid not increasing")
        isim = opts.get("simulation_number", None)
        pick = (lambda a: a) if isim is None else (lambda a: a[isim])
        if fmt == "multipoles":
            if len(keys) < 2:
                raise OracleInputError("wrong number of ccf keys")
            self.r = np.asarray(data[keys[0]])
            self.poles_r = np.atleast_1d([0, 2, 4][: len(keys) - 1])
            self.real_multipoles = {f"{l}": pick(np.asarray(data[keys[i + 1]]))
                                    for i, l in enumerate(self.poles_r)}
        elif fmt == "rmu":
            # ccf_model.py:154-181: bilinear interp2d of xi(r, mu) -> multipoles at the r nodes
            if len(keys) != 3:
                raise OracleInputError("wrong number of ccf keys")
            self.r = np.asarray(data[keys[0]])
            mu = np.asarray(data[keys[1]])
            ccf = pick(np.asarray(data[keys[2]]))
            self.poles_r = np.array([0, 2, 4])

            def project(table):  # table (n_r, n_mu)
                spl = si.RectBivariateSpline(self.r, mu, table, kx=1, ky=1, s=0)
                return multipoles_from_fn(lambda rj, m: spl(rj, m)[0], self.r, self.poles_r)

            if self.fixed_real_input:
                self.real_multipoles = project(ccf)
            else:
                self.real_multipoles = {f"{l}": np.zeros((len(self.beta), len(self.r))) for l in self.poles_r}
                for i in range(len(self.beta)):
                    tmp = project(ccf[i])
                    for l in self.poles_r:
                        self.real_multipoles[f"{l}"][i] = tmp[f"{l}"]
        else:
            raise OracleInputError("bad realspace format")

    def _set_matter_template(self, opts, data):
        """ccf_model.py:183-220."""
        keys = np.atleast_1d(opts.get("template_keys"))
        rd = np.asarray(data[keys[0]])
        dl = np.asarray(data[keys[1]])
        r = np.linspace(rd.min(), rd.max())
        if opts.get("integrated", False):
            self.integrated_delta = _ius(rd, dl)
            deriv = np.gradient(self.integrated_delta(r), r)
            self.delta = _ius(r, self.integrated_delta(r) + r * deriv / 3)
        else:
            self.delta = _ius(rd, dl)
            integral = np.zeros_like(r)
            for i in range(len(r)):
                integral[i] = quad(lambda x: 3 * self.delta(x) * x ** 2 / r[i] ** 3, 0, r[i], full_output=1)[0]
            self.integrated_delta = _ius(r, integral)

    def _set_velocity_pdf(self, opts, data):
        """ccf_model.py:222-297 (mean: linear or template; dispersion: template or constant*)."""
        mean_model = opts["mean"].get("model", "linear")
        self.has_velocity_template = False
        if mean_model == "template":
            self.template_fsigma8 = opts["mean"].get("template_fsigma8")
            self.z_sim = opts["mean"].get("z_sim", self.z_eff)
            self.template_hubble_ratio = opts["mean"].get("template_hubble_ratio", 1)
            keys = np.atleast_1d(opts["mean"].get("template_keys"))
            self.radial_velocity = _ius(np.asarray(data[keys[0]]), np.asarray(data[keys[1]]))
            self.has_velocity_template = True
        disp = opts.get("dispersion", {})
        dmodel = disp.get("model", "constant")
        if dmodel != "template":
            # ccf_model.py:284-291: 'constant' leaves ``sv`` undefined and the reference
            # dies with UnboundLocalError; nothing to restate.
            raise OracleInputError("dispersion model 'constant' crashes in the reference")
        keys = np.atleast_1d(disp.get("template_keys"))
        if len(keys) < 2 or len(keys) > 3:
            raise OracleInputError("need 2 or 3 dispersion template keys")
        self.r_for_sv = np.asarray(data[keys[0]])
        sv = np.asarray(data[keys[-1]])
        if len(keys) == 2:
            self.mu_for_sv = np.linspace(0, 1)
            sv = (np.ones((len(self.mu_for_sv), len(self.r_for_sv))) * sv).T
        else:
            self.mu_for_sv = np.asarray(data[keys[1]])
        if sv.shape != (len(self.r_for_sv), len(self.mu_for_sv)):
            raise OracleInputError("bad dispersion template shape")
        if disp.get("filter", True):
            from scipy.signal import savgol_filter
            w = disp.get("filter_window", 3)
            o = disp.get("filter_order", 1)
            sv = np.array([savgol_filter(sv[:, i], w, o) for i in range(sv.shape[1])]).T
        if sv.shape[0] == len(self.r_for_sv):
            sv = sv.T  # -> (n_mu, n_r)
        # ccf_model.py:295-297: interp2d default kind is *linear*; monopole at the last r node
        spl = si.RectBivariateSpline(self.r_for_sv, self.mu_for_sv, sv.T, kx=1, ky=1, s=0)
        mono = multipoles_from_fn(lambda rj, m: spl(rj, m)[0], self.r_for_sv, [0])
        self.sv_rmu = sv / mono["0"][-1]

    # ----- per-point pieces ------------------------------------------------- #
    def real_multipoles_at(self, beta):
        """ccf_model.py:299-326."""
        stack = np.array([self.real_multipoles[f"{l}"] for l in self.poles_r])
        if self.fixed_real_input:
            return np.atleast_2d(stack)
        if beta is None:
            raise OracleInputError("beta required")
        return np.atleast_2d(si.PchipInterpolator(self.beta, stack, axis=1)(beta))

    def delta_profiles(self, r, params, model):
        """ccf_model.py:328-383 (template and linear_bias branches)."""
        if model["matter_model"] == "linear_bias":
            bias = params.get("bias", model["bias"])
            xir = _ius(self.r, self.real_multipoles_at(params.get("beta", None))[0])
            integral = np.zeros_like(r)
            for i in range(len(r)):
                rr = np.linspace(0, r[i], 100)
                integral[i] = _trapz(xir(rr) * rr ** 2, rr)
            return xir(r) / bias, 3 * integral / (bias * r ** 3)
        if model["matter_model"] == "template":
            return self.delta(r), self.integrated_delta(r)
        raise OracleInputError("matter model")

    def velocity_terms(self, r, params, model):
        """ccf_model.py:385-492 (linear mean with/without empirical_corr; template mean)."""
        if "epsilon" in params:
            apar = params.get("alpha", 1) * params["epsilon"] ** (-2 / 3)
        else:
            apar = params.get("apar", 1)
        iaH_true = self.iaH * apar
        d_r, D_r = self.delta_profiles(r, params, model)
        delta = _ius(r, d_r)
        int_delta = _ius(r, D_r)
        if model["matter_model"] == "linear_bias":
            if model["realspace_ccf_from_data"]:
                growth = params["beta"] * params.get("bias", model["bias"])
            else:
                growth = params["fsigma8"] / self.template_sigma8
        else:
            growth = params["fsigma8"] / self.template_sigma8
        if model["mean_model"] == "template":
            shift = (1 + self.z_sim) / (1 + self.z_eff)
            growth = (params["fsigma8"] / self.template_fsigma8) * self.template_hubble_ratio * shift / apar
        if model["mean_model"] == "linear":
            if not model["empirical_corr"]:
                vr = -growth * r * int_delta(r) / (3 * iaH_true)
                dvr = -growth * (delta(r) - 2 * int_delta(r) / 3) / iaH_true
            else:
                Av = params.get("Av", 0)
                vr = -growth * r * int_delta(r) * (1 + Av * delta(r)) / (3 * iaH_true)
                rg = np.linspace(0.1, self.r.max(), 100)
                vg = -growth * rg * int_delta(rg) * (1 + Av * delta(rg)) / (3 * iaH_true)
                dvr = _ius(rg, np.gradient(vg, rg))(r)
        elif model["mean_model"] == "template":
            vr = self.radial_velocity(r) * growth
            rg = np.linspace(0.1, self.r.max(), 100)
            dvr = _ius(rg, np.gradient(self.radial_velocity(rg) * growth, rg))(r)
        else:
            raise OracleInputError("mean model")
        return vr, dvr

    def theory_xi(self, s, mu, params, **kwargs):
        """ccf_model.py:538-789.  ``s``, ``mu`` 1-D; returns (n_mu, n_s)."""
        model = dict(self.model)
        model.update(kwargs)
        rsd = model["rsd_model"]
        x = np.linspace(-6, 6) if rsd in ("streaming", "dispersion") else 0
        s = np.atleast_1d(s)
        mu = np.atleast_1d(mu)
        if s.ndim == 2 and mu.ndim == 2:
            S, Mu, X = np.meshgrid(np.unique(s), np.unique(mu), x)
        else:
            S, Mu, X = np.meshgrid(s, mu, x)
        if self.fixed_real_input and model["matter_model"] != "linear_bias":
            beta = 0.40
        else:
            beta = params["beta"]
        if "epsilon" in params:
            eps = params["epsilon"]
            apar = params.get("alpha", 1) * eps ** (-2 / 3)
            aperp = eps * apar
        else:
            aperp = params.get("aperp", 1)
            apar = params.get("apar", 1)
            eps = aperp / apar
        iaH_true = self.iaH * apar
        if model["velocity_independent_of_AP"]:
            c = params.get("astar", 1)
        else:
            m = np.linspace(1e-10, 1)
            c = _trapz(apar * np.sqrt(1 + (1 - m ** 2) * (eps ** 2 - 1)), m)
        r0 = self.r
        rc = r0 * c
        mult = self.real_multipoles_at(beta)
        xi_r = {}
        for i, l in enumerate(self.poles_r):
            xi_r[l] = _ius(r0 if model["realspace_ccf_from_data"] else rc, mult[i])
        r_ext = np.append([0.01], r0)
        vr, dvr = self.velocity_terms(r_ext, params, model)
        vr_i = _ius(np.append([0.01 * c], rc), vr)
        dvr_i = _ius(np.append([0.01 * c], rc), dvr / c)
        s_perp = S * np.sqrt(1 - Mu ** 2) * aperp
        s_par = S * Mu * apar
        s_true = np.sqrt(s_par ** 2 + s_perp ** 2)

        def xi_real(r, mu_r):
            if model["realspace_ccf_from_data"]:
                rp, rt = r_par / apar, s_perp / aperp
                r = np.sqrt(rp ** 2 + rt ** 2)
                mu_r = rp / r
            if model["assume_isotropic"]:
                return xi_r[0](r) * legendre(0)(mu_r)
            tot = np.zeros_like(r)
            for l in self.poles_r:
                tot = tot + xi_r[l](r) * legendre(l)(mu_r)
            return tot

        if rsd in ("streaming", "dispersion"):
            sigma_v = params.get("sigma_v", 380)
            v_par = X * sigma_v
            sv_spl = si.RectBivariateSpline(self.r_for_sv * c, self.mu_for_sv, self.sv_rmu.T)
            if rsd == "streaming":
                r_par = s_par - v_par * iaH_true
                r = np.sqrt(s_perp ** 2 + r_par ** 2)
                mu_r = r_par / r
                sv = sigma_v * sv_spl.ev(r, mu_r)
                pdf = norm.pdf(v_par, loc=vr_i(r) * mu_r, scale=sv)
                jac = 1
            else:
                r_par = (s_par - v_par * iaH_true) / (1 + iaH_true * vr_i(s_true) / s_true)
                for _ in range(model.get("niter", 5)):
                    r = np.sqrt(s_perp ** 2 + r_par ** 2)
                    r_par = (s_par - v_par * iaH_true) / (1 + iaH_true * vr_i(r) / r)
                r = np.sqrt(s_perp ** 2 + r_par ** 2)
                mu_r = r_par / r
                sv = sigma_v * sv_spl.ev(r, mu_r)
                pdf = norm.pdf(v_par, loc=0, scale=sv)
                jac = 1 / (1 + vr_i(r) * iaH_true / r + iaH_true * mu_r ** 2 * (dvr_i(r) - vr_i(r) / r))
            return velocity_integral((1 + xi_real(r, mu_r)) * jac * pdf, v_par, 2, model["simpson_even"]) - 1

        if rsd in ("kaiser", "euclid_special"):
            M = params.get("M", 1.0)
            Q = params.get("Q", 1.0)
            if model.get("kaiser_coord_shift", True):
                r_par = s_par / (1 + M * iaH_true * vr_i(s_true) / s_true)
                for _ in range(model.get("niter", 5)):
                    r = np.sqrt(s_perp ** 2 + r_par ** 2)
                    r_par = s_par / (1 + M * iaH_true * vr_i(r) / r)
            else:
                r_par = s_par
            r = np.sqrt(s_perp ** 2 + r_par ** 2)
            mu_r = r_par / r
            a, b = (1, 1) if rsd == "kaiser" else (3, 2)
            J = a * M * vr_i(r) * iaH_true / r + b * M * Q * mu_r ** 2 * iaH_true * (dvr_i(r) - vr_i(r) / r)
            xr = xi_real(r, mu_r)
            if rsd == "euclid_special":
                out = M * xr - J
            elif not model.get("kaiser_approximation", False):
                out = (1 + M * xr) / (1 + J) - 1
            else:
                out = M * xr - J
            return out[:, :, 0]
        raise OracleInputError(f"unknown rsd_model {rsd}")

    def theory_multipoles(self, s, params, poles=(0, 2), **kwargs):
        """ccf_model.py:791-827: 100 mu nodes, bicubic interp2d, utils.multipoles_from_fn."""
        poles = np.atleast_1d(poles)
        even = not np.any(poles % 2)
        mu = np.linspace(0, 1, 100) if even else np.linspace(-1, 1, 100)
        s = np.asarray(s, dtype=float)
        xi = self.theory_xi(s, mu, params, **kwargs)
        spl = si.RectBivariateSpline(s, mu, xi.T, kx=3, ky=3, s=0)
        return multipoles_from_fn(lambda sj, m: spl(sj, m)[0], s, poles, even=even), xi

    def theory_multipole_vector(self, s, params, poles=(0, 2), **kwargs):
        """ccf_model.py:829-860."""
        mp, _ = self.theory_multipoles(s, params, poles, **kwargs)
        return np.concatenate([mp[f"{l}"] for l in np.atleast_1d(poles)])


# --------------------------------------------------------------------------- #
class OracleFit(OracleModel):
    """Restates ``CCFFit`` (ccf_fit.py:10-483)."""

    def __init__(self, model, data, model_input=None, data_input=None, cov_input=None):
        super().__init__(model, model_input)
        import os
        base = data.get("dir", "")
        if data_input is None:
            data_input = load_input(os.path.join(base, data["redshift_space_ccf"]["data_file"]))
        if cov_input is None:
            cov_input = load_input(os.path.join(base, data["covariance_matrix"]["data_file"]))
        self._load_data(data["redshift_space_ccf"], data_input)
        self._load_cov(data["covariance_matrix"], cov_input)
        self.fit_options = {"beta_interpolation": data.get("beta_interpolation", "datavector"),
                            "likelihood": data.get("likelihood", {"form": "Gaussian"})}

    def _load_data(self, opts, d):
        """ccf_fit.py:44-114."""
        isim = opts.get("simulation_number", None)
        pick = (lambda a: a) if isim is None else (lambda a: a[isim])
        self.fixed_data = not opts.get("reconstruction", False)
        if not self.fixed_data:
            bkey = opts.get("beta_key", None)
            if bkey and bkey in d:
                self.beta_ccf = np.asarray(d[bkey])
            elif self.fixed_real_input:
                raise OracleInputError("beta info missing")
            else:
                self.beta_ccf = self.beta
        keys = np.atleast_1d(opts["ccf_keys"])
        if opts.get("format", "multipoles") != "multipoles" or len(keys) < 2:
            raise OracleInputError("data format")
        self.s = np.asarray(d[keys[0]])
        self.poles_s = np.atleast_1d([0, 2, 4][: len(keys) - 1])
        self.redshift_multipoles = {f"{l}": pick(np.asarray(d[keys[i + 1]])) for i, l in enumerate(self.poles_s)}

    def _load_cov(self, opts, d):
        """ccf_fit.py:116-164."""
        if not self.fixed_data:
            self.fixed_covmat = opts.get("fixed_beta", True)
            if not self.fixed_covmat:
                bkey = opts.get("beta_key", None)
                self.beta_covmat = np.asarray(d[bkey]) if (bkey and bkey in d) else self.beta_ccf
        else:
            self.fixed_covmat = True
        self.covmat = np.asarray(d[opts["cov_key"]])
        n = len(self.s) * len(self.poles_s)
        want = (n, n) if self.fixed_covmat else (len(self.beta_covmat), n, n)
        if self.covmat.shape != want:
            raise OracleInputError("covariance shape")
        self.icov = np.linalg.inv(self.covmat)

    def data_vector(self, beta=None):
        """ccf_fit.py:166-193, 306-323."""
        stack = np.array([self.redshift_multipoles[f"{l}"] for l in self.poles_s])
        if not self.fixed_data:
            if beta is None:
                raise OracleInputError("beta required")
            stack = si.PchipInterpolator(self.beta_ccf, stack, axis=1)(beta)
        return np.atleast_2d(stack).reshape(len(self.poles_s) * len(self.s))

    def _interp_stack(self, stack, beta):
        """ccf_fit.py:195-260, including the last-index upper bracket."""
        if self.fixed_covmat:
            return stack
        if beta is None:
            raise OracleInputError("beta required")
        g = self.beta_covmat
        if beta < g.min():
            return stack[0]
        if beta > g.max():
            return stack[-1]
        if beta in g:
            return stack[np.where(g == beta)[0][0]]
        lo = np.where(g < beta)[0][-1]
        hi = np.where(g >= beta)[0][-1]
        t = (beta - g[lo]) / (g[hi] - g[lo])
        return (1 - t) * stack[lo] + t * stack[hi]

    def chi_squared(self, params, **kwargs):
        """ccf_fit.py:325-354."""
        mkw = {k: v for k, v in kwargs.items()}
        t = self.theory_multipole_vector(self.s, params, self.poles_s, **mkw)
        d = self.data_vector(params.get("beta", None))
        cov = self._interp_stack(self.covmat, params.get("beta", None))
        icov = self._interp_stack(self.icov, params.get("beta", None))
        return np.dot(np.dot(t - d, icov), t - d), cov

    def _form(self, opts, chisq, factor):
        """ccf_fit.py:415-437 / 455-473."""
        form = opts["form"].lower()
        n = opts.get("nmocks", 1)
        nd = len(self.s) * len(self.poles_s)
        if form == "sellentin":
            return -n * np.log(1 + chisq / (n - 1)) / 2 + factor
        if form == "hartlap":
            return -0.5 * chisq * (n - nd - 2) / (n - 1) + factor
        if form == "percival":
            npar = opts["nparams"]
            B = (n - nd - 2) / ((n - nd - 1) * (n - nd - 4))
            m = npar + 2 + (n - 1 + B * (nd - npar)) / (1 + B * (nd - npar))
            return -m * np.log(1 + chisq / (n - 1)) / 2 + factor
        if form == "gaussian":
            return -0.5 * chisq + factor
        raise OracleInputError("Unrecognised likelihood form")

    def log_likelihood(self, params, **kwargs):
        """ccf_fit.py:356-483."""
        fo = dict(self.fit_options)
        fo.update(kwargs)
        like = fo["likelihood"]
        if fo["beta_interpolation"] == "likelihood" and not self.fixed_data:
            beta = params["beta"]
            g = self.beta_ccf
            lo = np.where(g < beta)[0][-1]
            hi = np.where(g >= beta)[0][0]
            t = (beta - g[lo]) / (g[hi] - g[lo])
            plo = dict(params, beta=g[lo])
            phi = dict(params, beta=g[hi])
            c_lo, cov_lo = self.chi_squared(plo, **kwargs)
            c_hi, cov_hi = self.chi_squared(phi, **kwargs)
            f_lo = f_hi = 0
            if not self.fixed_covmat:
                for cov, which in ((cov_lo, "lo"), (cov_hi, "hi")):
                    sign, ld = np.linalg.slogdet(cov)
                    if sign != 1:
                        return -np.inf, np.inf
                    if which == "lo":
                        f_lo = -0.5 * ld
                    else:
                        f_hi = -0.5 * ld
            lnlike = (1 - t) * self._form(like, c_lo, f_lo) + t * self._form(like, c_hi, f_hi)
            chisq = (1 - t) * c_lo + t * c_hi
        else:
            chisq, cov = self.chi_squared(params, **kwargs)
            factor = 0
            if not self.fixed_covmat:
                sign, ld = np.linalg.slogdet(cov)
                if sign != 1:
                    return -np.inf, np.inf
                factor = -0.5 * ld
            lnlike = self._form(like, chisq, factor)
        if np.isnan(lnlike):
            return -np.inf, np.inf
        return lnlike, chisq
