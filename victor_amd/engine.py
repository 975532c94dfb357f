"""Device engine: uploads the compiled tables and runs batches through the C ABI.

One :class:`Engine` owns one ``vk_ctx`` (one GPU, one stream).  It is created by
:class:`victor_amd.ccf_model.CCFModel` / :class:`victor_amd.ccf_fit.CCFFit` from their host
tables; users normally do not touch it, except for the device-resident batch API used by
``bench.py`` and :mod:`victor_amd.sharding`.
"""

import ctypes as C

import numpy as np

from . import _native as N
from . import tables as T
from . import velocity_tables as VT
from .utils import InputError


def _pp(knots, coef, lead=0):
    """Build a vk_pp (plus the arrays that must stay alive while it is used)."""
    knots = N.f64(knots)
    coef = N.f64(coef)
    pp = N.vk_pp()
    pp.n_int = len(knots) - 1
    h = T.uniform_spacing(knots, lead=lead)
    if h is not None:
        pp.lead, pp.inv_h = lead, 1.0 / h                    # exactly uniform
    else:
        hm = T.near_uniform_spacing(knots, lead=lead)
        if hm is not None:
            pp.lead, pp.inv_h = lead, -1.0 / hm              # nearly uniform: estimate + one-step correction
        else:
            pp.lead, pp.inv_h = 0, 0.0                       # arbitrary knots: binary search
    pp.knots = N.as_dp(knots)
    pp.coef = N.as_dp(coef)
    return pp, (knots, coef)


def build_tables(model, fit=None, matter_model=None, simpson_even=None):
    """Compile a ``CCFModel`` (and optionally the data side of a ``CCFFit``) into vk_tables.

    Only plain attributes are read (see :mod:`victor_amd.velocity_tables`), so ``model`` / ``fit`` may equally be
    objects of the reference package.  ``simpson_even`` names the even-N Simpson convention of the velocity integral
    (:func:`victor_amd.tables.simpson_weights`; ``None`` = the default, SciPy >= 1.11).  Returns ``(tables, keepalive)``.
    """
    keep = []

    def arr(a):
        a = N.f64(a)
        keep.append(a)
        return a

    t = N.vk_tables()
    if fit is not None:
        s = arr(fit.s)
        poles = np.atleast_1d(fit.poles_s)
    else:
        s = arr(model.r)
        poles = np.array([0, 2])
    mu = arr(T.mu_nodes_for(poles))
    w = arr(T.projection_weights(mu, poles))
    n_x = 50                                   # ccf_model.py:570 (np.linspace default num)
    x = arr(np.linspace(-6, 6, n_x))
    try:
        w_unit = T.simpson_weights(n_x, simpson_even)          # ccf_model.py:690 (simps, default even=)
    except ValueError as exc:
        raise InputError(str(exc))
    w_x = arr(w_unit * (12.0 / (n_x - 1)) / np.sqrt(2 * np.pi))
    t.n_s, t.n_mu, t.n_x, t.n_ell = len(s), len(mu), n_x, len(poles)
    t.s, t.mu, t.w_ell, t.x, t.w_x = map(N.as_dp, (s, mu, w, x, w_x))

    # real-space multipoles ---------------------------------------------------------------
    r = N.f64(model.r)
    n_ell_r = len(model.poles_r)
    if n_ell_r > 3:
        raise InputError("at most three real-space multipoles are supported")
    if model.fixed_real_input:
        coef = np.stack([T.notaknot_coefficients(r, model.real_multipoles[f"{l}"]) for l in model.poles_r])
        t.n_beta_r = 0
        t.beta_r = None
    else:
        beta = arr(model.beta)
        coef = np.stack([T.beta_dependent_spline_table(r, beta, model.real_multipoles[f"{l}"])
                         for l in model.poles_r])
        t.n_beta_r = len(beta)
        t.beta_r = N.as_dp(beta)
    t.n_ell_r = n_ell_r
    t.xi, k = _pp(r, coef)
    keep.append(k)

    # velocity profile tables on r_ext = [0.01, r...] (ccf_model.py:625, 449-459) for the chosen matter model
    matter_model = matter_model or model.matter_model
    if matter_model == "excursion_set":
        raise InputError("matter_model 'excursion_set' is not implemented on the HIP path")
    if matter_model not in N.MATTER:
        raise InputError(f"matter_model '{matter_model}' is not implemented on the HIP path")
    r_ext = np.append([0.01], r)
    vr_coef, vr_beta_dep = VT.velocity_tables(model, matter_model)
    t.matter_model = N.MATTER[matter_model]
    t.vr_beta_dep = 1 if vr_beta_dep else 0
    t.vr, k = _pp(r_ext, vr_coef, lead=1)
    keep.append(k)
    if vr_beta_dep:
        vr_emp = arr(VT.empirical_beta_tables(model))
        t.vr_emp = N.as_dp(vr_emp)

    if matter_model == "velocity_template":
        t.vt_amp = float(model.template_hubble_ratio * (1 + model.z_sim) / (1 + model.z_eff) / model.template_fsigma8)

    # dispersion template.  Isotropic: the bicubic RectBivariateSpline of ccf_model.py:654 through mu-independent
    # data is the 1-D not-a-knot spline in r.  Anisotropic (3 keys): bicubic patches, evaluated from global memory.
    if VT.dispersion_is_isotropic(model):
        t.sv, k = _pp(model.r_for_sv, T.notaknot_coefficients(model.r_for_sv, model.sv_rmu[0])[None])
        keep.append(k)
        t.sv_n_mu = 0
    else:
        patches = arr(T.bicubic_patches(model.r_for_sv, model.mu_for_sv, model.sv_rmu.T))
        mu_sv = arr(model.mu_for_sv)
        t.sv, k = _pp(model.r_for_sv, patches)          # coef pointer unused by the kernels in this mode
        keep.append(k)
        t.sv_n_mu = len(mu_sv)
        hmu = T.uniform_spacing(mu_sv)
        t.sv_mu_inv_h = 1.0 / hmu if hmu is not None else 0.0
        t.sv_mu = N.as_dp(mu_sv)
        t.sv2d = N.as_dp(patches)

    # unified tables for the fast kernels (fixed velocity tables, isotropic sigma_v template).  Uniform, commensurate
    # r and r_sv grids: the lattice form (index arithmetic).  Anything else: the union-grid form (look-up table).
    t.uni_n = 0
    t.uni_lut_n = 0
    # (an anisotropic sigma_v(r, mu) template keeps its bicubic patches in `sv2d`; the sigma_v half of the unified records then
    # holds the mu = mu_0 row and is not read by the kernels that take the patches)
    if r[0] > r_ext[0] and model.r_for_sv[0] >= r_ext[0]:
        grid = None
        if t.xi.inv_h > 0 and t.sv.inv_h > 0 and t.vr.inv_h > 0:
            cr = T.common_refinement(r, model.r_for_sv)
            if cr is not None and cr[2] + int(np.ceil(cr[0] / cr[1])) <= 640:
                u0, h, n = cr
                # extend the grid down to u <= 0 so that the index-unit coordinate t = r/(c h) - u0/h is never
                # negative and the leading V interval [0.01, r_0] is part of the same records (vk_kernel_fast.h)
                k0 = int(np.ceil(u0 / h - 1e-9))
                u0, n = u0 - k0 * h, n + k0
                if abs(u0) < 1e-9 * h:
                    u0 = 0.0
                grid = (u0 + h * np.arange(n), np.full(n, h))
                t.uni_n, t.uni_u0, t.uni_inv_h = n, u0, 1.0 / h
        if grid is None:
            ug = T.union_grid(r_ext, model.r_for_sv)
            if ug is not None and len(ug[0]) - 1 <= 512:
                U, lut, inv_g = ug
                grid = (U[:-1], np.diff(U))
                lut = np.ascontiguousarray(lut, dtype=np.uint16)
                knots_u = arr(U)
                keep.append(lut)
                t.uni_n, t.uni_u0, t.uni_inv_h = len(U) - 1, float(U[0]), 0.0
                t.uni_lut_n, t.uni_lut_inv_g = len(lut), float(inv_g)
                t.uni_lut = lut.ctypes.data_as(C.POINTER(C.c_uint16))
                t.uni_knots = N.as_dp(knots_u)
        if grid is not None:
            left, width = grid
            sv_ref = T.refine_pp_on(model.r_for_sv, T.notaknot_coefficients(model.r_for_sv, model.sv_rmu[0]), left, width)
            if vr_beta_dep:       # V1 as beta polynomials, rebuilt per point like xi^r (vr_coef[0]: (n_beta-1, n_int, 4, 4))
                vb = T.refine_pp_on(r_ext, np.moveaxis(vr_coef[0], 0, -1), left, width)    # (n, 4, 4, n_beta-1)
                uni_vb = arr(np.moveaxis(vb, -1, 0))                                       # (n_beta-1, n, 4, 4)
                t.uni_vb = N.as_dp(uni_vb)
                v_ref = np.zeros_like(sv_ref)
                # Da likewise (dispersion model), and the degree-6 beta polynomials of the empirical_corr branch (V2, Ge1, Ge2)
                dab = T.refine_pp_on(r_ext, np.moveaxis(vr_coef[1], 0, -1), left, width)
                uni_dab = arr(np.moveaxis(dab, -1, 0))
                t.uni_dab = N.as_dp(uni_dab)
                uni_empb = arr(np.stack([np.moveaxis(T.refine_pp_on(r_ext, np.moveaxis(vr_emp[v], 0, -1), left, width), -1, 0)
                                         for v in range(3)]))                               # (3, n_beta-1, n, 4, 7)
                t.uni_empb = N.as_dp(uni_empb)
            else:
                v_ref = T.refine_pp_on(r_ext, vr_coef[0], left, width)
                uni_v2 = arr(T.refine_pp_on(r_ext, vr_coef[2], left, width))              # empirical_corr: V = V1 + Av V2
                t.uni_v2 = N.as_dp(uni_v2)
                uni_da = arr(T.refine_pp_on(r_ext, vr_coef[1], left, width))              # dispersion model: v_r'(r)
                t.uni_da = N.as_dp(uni_da)
                uni_ge = arr(np.stack([T.refine_pp_on(r_ext, vr_coef[3], left, width),
                                       T.refine_pp_on(r_ext, vr_coef[4], left, width)]))   # empirical_corr: v_r'(r)
                t.uni_ge = N.as_dp(uni_ge)
            uni_sv_v = arr(np.stack([sv_ref, v_ref], axis=1))                         # (n, 2, 4)
            if model.fixed_real_input:
                uni_xi = arr(np.stack([T.refine_pp_on(r, coef[l], left, width) for l in range(n_ell_r)]))   # (L, n, 4)
            else:
                parts = []
                for l in range(n_ell_r):                                              # coef[l]: (n_beta-1, n_int, 4, 4)
                    ref = T.refine_pp_on(r, np.moveaxis(coef[l], 0, -1), left, width)  # (n, 4, 4, n_beta-1)
                    parts.append(np.moveaxis(ref, -1, 0))                              # (n_beta-1, n, 4, 4)
                uni_xi = arr(np.stack(parts))
            # Legendre sum regrouped in powers of mu_r^2 (exact: linear combinations of the coefficient sets)
            if n_ell_r == 1:
                uni_xic = uni_xi
            else:
                x0, x2 = uni_xi[0], uni_xi[1]
                x4 = uni_xi[2] if n_ell_r == 3 else np.zeros_like(x0)
                comb = [x0 - 0.5 * x2 + 0.375 * x4, 1.5 * x2 - 3.75 * x4, 4.375 * x4]
                uni_xic = arr(np.stack(comb[:n_ell_r]))
            t.uni_sv_v, t.uni_xi, t.uni_xic = N.as_dp(uni_sv_v), N.as_dp(uni_xi), N.as_dp(uni_xic)

    t.iaH = float(model.iaH)
    # not used when the growth term is beta*bias (linear_bias on a measured real-space ccf)
    t.template_sigma8 = float(model.template_sigma8) if model.template_sigma8 else 1.0

    # data side -----------------------------------------------------------------------------
    if fit is not None:
        Nd = len(s) * len(poles)
        stack = np.array([fit.redshift_multipoles[f"{l}"] for l in poles])   # (n_ell, [n_beta,] n_s)
        if fit.fixed_data:
            t.n_beta_d = 0
            d = arr(stack.reshape(Nd))
        else:
            bd = arr(fit.beta_ccf)
            vals = np.transpose(stack, (1, 0, 2)).reshape(len(bd), Nd)     # (n_beta, N)
            pc = T.pchip_coefficients(bd, vals)                            # (n_beta-1, 4, N)
            d = arr(np.transpose(pc, (0, 2, 1)))                           # (n_beta-1, N, 4)
            t.n_beta_d = len(bd)
            t.beta_d = N.as_dp(bd)
        t.data = N.as_dp(d)
        prec = arr(fit.icov)
        t.prec = N.as_dp(prec)
        if fit.fixed_covmat:
            t.n_beta_c = 0
        else:
            import scipy.linalg as sl
            bc = arr(fit.beta_covmat)
            nb = len(bc)
            logdet = np.empty(nb)
            eig = np.ones((nb, Nd))
            for kk in range(nb):
                sign, ld = np.linalg.slogdet(fit.covmat[kk])
                logdet[kk] = ld if sign == 1 else np.nan
                if kk < nb - 1 and sign == 1:
                    # cov[last] v = lambda cov[k] v  =>  det((1-t) cov[k] + t cov[last]) = det(cov[k]) prod(1-t+t lambda)
                    try:
                        eig[kk] = sl.eigh(fit.covmat[-1], fit.covmat[kk], eigvals_only=True)
                    except np.linalg.LinAlgError:
                        # slice k has a positive determinant but is not positive definite (an even number of negative
                        # eigenvalues): the identity above does not hold through eigh.  The reference constructs such a
                        # fit and decides point by point from slogdet of the blend (ccf_fit.py:447-450).  Here the slice
                        # keeps its own log det - a point ON this grid value (t = 0, no blend) evaluates as in the
                        # reference - and NaN factors mark every BLENDED evaluation that starts from it as failed
                        # (-inf, inf): a covariance matrix that is not positive definite is an input error in any case.
                        import warnings
                        warnings.warn("covariance slice %d (beta = %g) has a positive determinant but is not positive "
                                      "definite: likelihood evaluations that interpolate from it (beta between this grid "
                                      "value and the next) will report -inf" % (kk, float(bc[kk])), RuntimeWarning)
                        eig[kk] = np.nan
            logdet = arr(logdet)
            eig = arr(eig)
            t.n_beta_c = nb
            t.beta_c = N.as_dp(bc)
            t.logdet = N.as_dp(logdet)
            t.eig = N.as_dp(eig)
    return t, keep


def table_array_lengths(t):
    """Element counts of every array of a ``vk_tables`` (the shapes documented in include/victor_hip.h)."""
    N_ = t.n_ell * t.n_s
    nb = t.n_beta_r
    per_beta = lambda fixed, dep: dep if nb > 0 else fixed   # noqa: E731
    out = {
        "s": t.n_s, "mu": t.n_mu, "w_ell": t.n_ell * t.n_mu, "x": t.n_x, "w_x": t.n_x, "beta_r": nb,
        "xi.knots": t.xi.n_int + 1,
        "xi.coef": per_beta(t.n_ell_r * t.xi.n_int * 4, t.n_ell_r * (nb - 1) * t.xi.n_int * 16),
        "vr.knots": t.vr.n_int + 1,
        "vr.coef": 2 * (nb - 1) * t.vr.n_int * 16 if t.vr_beta_dep else 5 * t.vr.n_int * 4,
        "vr_emp": 3 * (nb - 1) * t.vr.n_int * 28 if (t.vr_beta_dep and t.vr_emp) else 0,
        "sv.knots": t.sv.n_int + 1, "sv.coef": 4 if t.sv_n_mu else t.sv.n_int * 4,
        "sv_mu": t.sv_n_mu, "sv2d": t.sv.n_int * (t.sv_n_mu - 1) * 16 if t.sv_n_mu else 0,
        "uni_sv_v": t.uni_n * 8,
        "uni_xi": per_beta(t.n_ell_r * t.uni_n * 4, t.n_ell_r * (nb - 1) * t.uni_n * 16) if t.uni_n else 0,
        "uni_xic": per_beta(t.n_ell_r * t.uni_n * 4, t.n_ell_r * (nb - 1) * t.uni_n * 16) if t.uni_n else 0,
        "uni_vb": (nb - 1) * t.uni_n * 16 if (t.uni_n and t.vr_beta_dep and t.uni_vb) else 0,
        "uni_v2": t.uni_n * 4 if (t.uni_n and t.uni_v2) else 0,
        "uni_da": t.uni_n * 4 if (t.uni_n and t.uni_da) else 0,
        "uni_ge": t.uni_n * 8 if (t.uni_n and t.uni_ge) else 0,
        "uni_dab": (nb - 1) * t.uni_n * 16 if (t.uni_n and t.vr_beta_dep and t.uni_dab) else 0,
        "uni_empb": 3 * (nb - 1) * t.uni_n * 28 if (t.uni_n and t.vr_beta_dep and t.uni_empb) else 0,
        "uni_lut": t.uni_lut_n, "uni_knots": t.uni_n + 1 if t.uni_lut_n else 0,
        "beta_d": t.n_beta_d, "data": (t.n_beta_d - 1) * N_ * 4 if t.n_beta_d else (N_ if t.data else 0),
        "beta_c": t.n_beta_c, "prec": (max(t.n_beta_c, 1) * N_ * N_) if t.prec else 0,
        "logdet": t.n_beta_c, "eig": t.n_beta_c * N_,
    }
    return out


def dump_tables(t, path):
    """Write a ``vk_tables`` as a flat record stream (name[24], kind, count, payload) that a client in any language can
    load into the C struct - see examples/c_abi_client.c."""
    import struct
    lengths = table_array_lengths(t)

    def walk(obj, prefix=""):
        for name, ctype in obj._fields_:
            val = getattr(obj, name)
            full = prefix + name
            if isinstance(val, C.Structure):
                yield from walk(val, full + ".")
            elif ctype is C.c_int32:
                yield full, 0, int(val)
            elif ctype is C.c_double:
                yield full, 1, float(val)
            else:
                n = lengths[full] if val else 0
                kind = 3 if ctype._type_ is C.c_uint16 else 2
                dtype = np.uint16 if kind == 3 else np.float64
                arr = np.ctypeslib.as_array(val, shape=(n,)).astype(dtype, copy=True) if n else np.empty(0, dtype)
                yield full, kind, arr

    with open(path, "wb") as f:
        f.write(b"VKTB1\0\0\0")
        for name, kind, val in walk(t):
            head = name.encode().ljust(24, b"\0")
            if kind == 0:
                f.write(head + struct.pack("<iq", 0, 1) + struct.pack("<q", val))
            elif kind == 1:
                f.write(head + struct.pack("<iq", 1, 1) + struct.pack("<d", val))
            else:
                raw = val.tobytes()
                raw += b"\0" * (-len(raw) % 8)
                f.write(head + struct.pack("<iq", kind, len(val)) + raw)
        f.write(b"END".ljust(24, b"\0") + struct.pack("<iq", 0, 0))


class Engine:
    def __init__(self, model, fit=None, device=0, matter_model=None, simpson_even=None, lib=None):
        self._lib = lib or N.load()        # `lib`: another build of the library (development runs: tests/devlib.py)
        if self._lib.vk_device_count() <= 0:
            raise N.NativeError("no HIP device visible; victor_amd has no CPU fallback")
        tables, keep = build_tables(model, fit, matter_model, simpson_even)
        self.simpson_even = T.simpson_even_rule(simpson_even)
        err = C.create_string_buffer(512)
        self._ctx = self._lib.vk_create(C.byref(tables), int(device), err, len(err))
        del keep
        if not self._ctx:
            raise N.NativeError("vk_create failed: " + err.value.decode())
        self.device = int(device)
        self.n_data = tables.n_ell * tables.n_s
        self.has_data = fit is not None

    # -- plumbing ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.vk_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if getattr(self, "_dead", None):
            raise N.CommInitTimeout(self._dead)
        if rc != 0:
            msg = self._lib.vk_last_error(self._ctx)
            msg = msg.decode() if msg else "unknown error"
            if rc == -1:
                raise InputError(msg)
            raise N.NativeError(f"libvictor_hip error {rc}: {msg}")

    def make_opts(self, model_opts, fit_opts=None):
        """Translate the reference's option dicts (ccf_model.py:85-97, ccf_fit.py:41-42)."""
        o = N.vk_eval_opts()
        self._lib.vk_default_opts(C.byref(o))
        rsd = model_opts["rsd_model"]
        if rsd not in N.RSD:
            raise InputError(f"theory_xi: Unrecognised choice of model {rsd}")   # ccf_model.py:787
        o.rsd_model = N.RSD[rsd]
        o.assume_isotropic = 1 if model_opts["assume_isotropic"] else 0
        o.rescale_from_ap = 0 if model_opts["velocity_independent_of_AP"] else 1
        o.kaiser_approx = 1 if model_opts.get("kaiser_approximation", False) else 0
        o.kaiser_coord_shift = 1 if model_opts.get("kaiser_coord_shift", True) else 0
        o.niter = int(model_opts.get("niter", 5))
        o.from_data = 1 if model_opts.get("realspace_ccf_from_data", False) else 0
        o.empirical_corr = 1 if model_opts.get("empirical_corr", False) else 0
        if fit_opts is not None:
            like = fit_opts["likelihood"]
            form = like["form"].lower()
            if form not in N.LIKE:
                raise InputError("Unrecognised likelihood form")                 # ccf_fit.py:473
            o.like_form = N.LIKE[form]
            o.nmocks = float(like.get("nmocks", 1))
            if form == "percival":
                o.nparams = float(like["nparams"])                                # KeyError as in ccf_fit.py:465
        return o

    # -- host-buffer entry points -----------------------------------------------------------
    def eval_batch(self, opts, rows, want_theory=False):
        rows = N.f64(rows).reshape(-1, N.VK_NPAR)
        n = len(rows)
        lnl = np.empty(n)
        chi2 = np.empty(n)
        theory = np.empty((n, self.n_data)) if want_theory else None
        self._check(self._lib.vk_eval_batch(self._ctx, C.byref(opts), N.as_dp(rows), n, N.as_dp(lnl),
                                            N.as_dp(chi2), N.as_dp(theory) if want_theory else None))
        return lnl, chi2, theory

    def eval_point(self, opts, row):
        """One point (a list of VK_NPAR floats) through ``vk_eval_batch`` with preallocated buffers: the single-point calls
        of an MCMC driver spend as long in Python as on the GPU, so nothing is allocated or converted per call."""
        one = getattr(self, "_one", None)
        if one is None:
            buf = np.empty((1, N.VK_NPAR))
            out = np.empty(2)
            one = self._one = (buf, out, N.as_dp(buf), N.as_dp(out[0:1]), N.as_dp(out[1:2]))
        buf, out, p_rows, p_lnl, p_chi = one
        buf[0] = row
        rc = self._lib.vk_eval_batch(self._ctx, opts, p_rows, 1, p_lnl, p_chi, None)
        if rc != 0:
            self._check(rc)
        return float(out[0]), float(out[1])

    def theory_vector_batch(self, opts, rows):
        rows = N.f64(rows).reshape(-1, N.VK_NPAR)
        n = len(rows)
        theory = np.empty((n, self.n_data))
        self._check(self._lib.vk_eval_batch(self._ctx, C.byref(opts), N.as_dp(rows), n, None, None,
                                            N.as_dp(theory)))
        return theory

    def theory_batch(self, opts, rows, s, poles):
        rows = N.f64(rows).reshape(-1, N.VK_NPAR)
        s = N.f64(np.atleast_1d(s))
        poles = np.atleast_1d(poles)
        mu = N.f64(T.mu_nodes_for(poles))
        w = N.f64(T.projection_weights(mu, poles))
        out = np.empty((len(rows), len(poles), len(s)))
        self._check(self._lib.vk_theory_batch(self._ctx, C.byref(opts), N.as_dp(rows), len(rows), N.as_dp(s),
                                              len(s), N.as_dp(mu), len(mu), N.as_dp(w), len(poles), N.as_dp(out)))
        return out

    def xi_smu_batch(self, opts, rows, s, mu):
        rows = N.f64(rows).reshape(-1, N.VK_NPAR)
        s = N.f64(np.atleast_1d(s))
        mu = N.f64(np.atleast_1d(mu))
        out = np.empty((len(rows), len(mu), len(s)))
        self._check(self._lib.vk_xi_smu_batch(self._ctx, C.byref(opts), N.as_dp(rows), len(rows), N.as_dp(s),
                                              len(s), N.as_dp(mu), len(mu), N.as_dp(out)))
        return out

    # -- device-resident API (bench, sharding) ------------------------------------------------
    def alloc(self, n_doubles):
        p = self._lib.vk_device_alloc(self._ctx, int(n_doubles) * 8)
        if not p:
            raise N.NativeError("device allocation failed")
        return p

    def free(self, ptr):
        self._lib.vk_device_free(self._ctx, ptr)

    def upload(self, ptr, host):
        host = N.f64(host)
        self._check(self._lib.vk_memcpy_h2d(self._ctx, ptr, host.ctypes.data, host.nbytes))

    def download(self, ptr, n_doubles):
        out = np.empty(int(n_doubles))
        self._check(self._lib.vk_memcpy_d2h(self._ctx, out.ctypes.data, ptr, out.nbytes))
        return out

    def eval_device_async(self, opts, d_rows, n, d_lnl, d_chi2, d_theory_ws):
        self._check(self._lib.vk_eval_batch_device_async(self._ctx, C.byref(opts), d_rows, int(n), d_lnl, d_chi2,
                                                         d_theory_ws))

    def last_kernel(self):
        return self._lib.vk_last_kernel(self._ctx).decode()

    def last_fused(self):
        return bool(self._lib.vk_last_fused(self._ctx))

    def last_polled(self):
        return bool(self._lib.vk_last_polled(self._ctx))

    def sync(self):
        self._check(self._lib.vk_sync(self._ctx))

    def timing(self, on):
        self._check(self._lib.vk_timing_enable(self._ctx, 1 if on else 0))

    def read_timing(self, reset=True):
        a, b, k = C.c_double(), C.c_double(), C.c_int64()
        self._check(self._lib.vk_timing_read(self._ctx, C.byref(a), C.byref(b), C.byref(k), 1 if reset else 0))
        return a.value, b.value, k.value

    # -- RCCL ----------------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(N.VK_COMM_ID_BYTES)
        rc = self._lib.vk_comm_unique_id(buf)
        if rc != 0:
            raise N.NativeError("vk_comm_unique_id failed (is librccl available?)")
        return buf.raw

    def bus_id(self):
        buf = C.create_string_buffer(64)
        self._check(self._lib.vk_device_bus_id(self._ctx, buf, len(buf)))
        return buf.value.decode()

    def comm_init(self, uid, rank, nranks, timeout=120.0):
        """ncclCommInitRank for this context.  The call blocks until every rank has joined; it runs on a helper thread so that a
        rendezvous that never completes (a rank that died, a fabric problem) surfaces as :class:`CommInitTimeout` after
        ``timeout`` seconds instead of a silent hang.  That error is FATAL for the process: the helper thread is still inside
        RCCL with this context (its stream, a half-built communicator), so the engine refuses every later call and the caller
        must exit - not fall back to the host gather on the same context (bench.py, examples/run_walkers.py exit non-zero)."""
        import threading
        box = {}

        def work():
            box["rc"] = self._lib.vk_comm_init(self._ctx, uid, int(rank), int(nranks))

        t = threading.Thread(target=work, daemon=True)
        t.start()
        t.join(timeout)
        if t.is_alive():
            self._dead = (f"ncclCommInitRank did not complete within {timeout:.0f} s; this context is still inside RCCL and "
                          "is unusable - restart the job")
            raise N.CommInitTimeout(self._dead)
        self._check(box["rc"])

    def comm_allgather_async(self, d_send, d_recv, count):
        self._check(self._lib.vk_comm_allgather_async(self._ctx, d_send, d_recv, int(count)))

    def comm_allgather_host_begin(self, local):
        local = N.f64(local)
        self._check(self._lib.vk_comm_allgather_host_begin(self._ctx, N.as_dp(local), local.size))

    def comm_allgather_host_finish(self, count, nranks):
        out = np.empty(int(count) * int(nranks))
        self._check(self._lib.vk_comm_allgather_host_finish(self._ctx, N.as_dp(out)))
        return out

    def comm_destroy(self):
        self._check(self._lib.vk_comm_destroy(self._ctx))

    def comm_rank_info(self):
        """What the live communicator of this context says about itself - {"count": ncclCommCount, "rank": ncclCommUserRank,
        "device": ncclCommCuDevice}, a value ``None`` where the loaded RCCL does not export the call - or ``None`` without a
        communicator.  A multi-GPU record carries it per rank next to the PCI bus id (bench.py: ``config.rccl.ranks``)."""
        c, r, d = C.c_int32(), C.c_int32(), C.c_int32()
        if self._lib.vk_comm_rank_info(self._ctx, C.byref(c), C.byref(r), C.byref(d)) != 0:
            return None
        return {k: (v.value if v.value >= 0 else None) for k, v in (("count", c), ("rank", r), ("device", d))}

    # one process driving several GPUs: the engines of a group, context i = rank i (vk_comm_init_all)
    @staticmethod
    def comm_init_all(engines):
        lead = engines[0]
        ctxs = (C.c_void_p * len(engines))(*[e._ctx for e in engines])
        lead._check(lead._lib.vk_comm_init_all(ctxs, len(engines)))

    @staticmethod
    def comm_allgather_group_async(engines, d_send, d_recv, count):
        """One grouped all-gather: ``d_send[i]`` (count doubles) and ``d_recv[i]`` (len(engines) * count doubles) live on
        engine i's device; enqueued on every engine's stream."""
        lead = engines[0]
        n = len(engines)
        ctxs = (C.c_void_p * n)(*[e._ctx for e in engines])
        snd = (C.c_void_p * n)(*d_send)
        rcv = (C.c_void_p * n)(*d_recv)
        lead._check(lead._lib.vk_comm_allgather_group_async(ctxs, n, snd, rcv, int(count)))
