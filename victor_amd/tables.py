"""Host-side table compiler.

The reference builds SciPy spline objects on every likelihood call
(``ccf_model.py:615-636,654``; ``ccf_fit.py:193``).  The HIP kernels evaluate
explicit piecewise-cubic tables instead, so everything the reference expresses
as a spline object is turned here - once, at construction - into coefficient
arrays:

* not-a-knot cubic splines  == ``InterpolatedUnivariateSpline(k=3)`` (``ccf_model.py:17``;
  FITPACK drops the 2nd and 2nd-last data points as knots, which *is* the not-a-knot
  condition, so one cubic per data interval reproduces it exactly);
* PCHIP pieces in the reconstruction parameter beta == ``PchipInterpolator``
  (``ccf_model.py:326``, ``ccf_fit.py:193``);
* Legendre projection weights == cubic ``interp2d`` + ``utils.multipoles_from_fn``
  (``ccf_model.py:824-825``, ``utils.py:45-56``), a fixed linear map of the mu nodes;
* composite Simpson weights == ``scipy.integrate.simpson`` (``ccf_model.py:690``),
  including the SciPy >= 1.11 treatment of an even number of points.

All routines are written from the textbook formulas (no SciPy spline objects are used
here); ``tests/test_tables.py`` checks each against the SciPy primitive it replaces.
"""

import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz


# --------------------------------------------------------------------------- #
# piecewise-cubic containers
# --------------------------------------------------------------------------- #
class PiecewiseCubic:
    """f(u) = sum_p coef[i, p, ...] (u - knots[i])**p on [knots[i], knots[i+1]].

    ``clamp=True`` evaluates with the argument clamped to the knot range, which is the
    behaviour of FITPACK ``ext=3`` (``ccf_model.py:17``); ``clamp=False`` extrapolates
    with the first / last piece (``PchipInterpolator`` default).
    """

    def __init__(self, knots, coef, clamp=True):
        self.knots = np.ascontiguousarray(knots, dtype=np.float64)
        self.coef = np.ascontiguousarray(coef, dtype=np.float64)  # (n_int, 4, ...)
        self.clamp = clamp

    def __call__(self, u):
        u = np.asarray(u, dtype=np.float64)
        k = self.knots
        if self.clamp:
            u = np.clip(u, k[0], k[-1])
        i = np.clip(np.searchsorted(k, u, side="right") - 1, 0, len(k) - 2)
        dx = u - k[i]
        c = self.coef
        extra = (1,) * (c.ndim - 2)
        dxe = dx.reshape(dx.shape + extra)
        return ((c[i, 3] * dxe + c[i, 2]) * dxe + c[i, 1]) * dxe + c[i, 0]

    def scalar_function(self):
        """A fast float -> float evaluator of a scalar-valued table (for adaptive quadrature loops)."""
        from bisect import bisect_right
        if self.coef.ndim != 2:
            raise ValueError("scalar_function needs a scalar-valued table")
        k = self.knots.tolist()
        c = self.coef.tolist()
        lo, hi, last, clamp = k[0], k[-1], len(k) - 2, self.clamp

        def f(u):
            if clamp:
                u = lo if u < lo else (hi if u > hi else u)
            i = bisect_right(k, u) - 1
            i = 0 if i < 0 else (last if i > last else i)
            dx = u - k[i]
            ci = c[i]
            return ((ci[3] * dx + ci[2]) * dx + ci[1]) * dx + ci[0]

        return f


def _hermite_to_power(y, d, h):
    """Cubic Hermite data (values y, slopes d on intervals of width h) -> power-basis pieces."""
    extra = (1,) * (y.ndim - 1)
    hh = h.reshape(h.shape + extra)
    m = (y[1:] - y[:-1]) / hh
    c0 = y[:-1]
    c1 = d[:-1]
    c2 = (3 * m - 2 * d[:-1] - d[1:]) / hh
    c3 = (d[:-1] + d[1:] - 2 * m) / (hh * hh)
    return np.stack([c0, c1, c2, c3], axis=1)  # (n-1, 4, ...)


def notaknot_coefficients(x, y):
    """Interpolating cubic spline with not-a-knot end conditions.

    ``x`` (n,) strictly increasing, n >= 4; ``y`` (n, ...).  Returns (n-1, 4, ...)
    power-basis coefficients about the left end of each data interval.
    """
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = len(x)
    if n < 4:
        raise ValueError("cubic not-a-knot spline needs at least 4 points")
    if not np.all(np.diff(x) > 0):
        raise ValueError("spline abscissae must be strictly increasing")
    h = np.diff(x)
    extra = (1,) * (y.ndim - 1)
    hh = h.reshape(h.shape + extra)
    m = (y[1:] - y[:-1]) / hh
    A = np.zeros((n, n))
    b = np.zeros(y.shape)
    for i in range(1, n - 1):
        A[i, i - 1] = h[i]
        A[i, i] = 2 * (h[i - 1] + h[i])
        A[i, i + 1] = h[i - 1]
        b[i] = 3 * (h[i] * m[i - 1] + h[i - 1] * m[i])
    # third derivative continuous across x[1] and x[n-2]
    d0 = x[2] - x[0]
    A[0, 0] = h[1]
    A[0, 1] = d0
    b[0] = ((h[0] + 2 * d0) * h[1] * m[0] + h[0] ** 2 * m[1]) / d0
    d1 = x[-1] - x[-3]
    A[-1, -1] = h[-2]
    A[-1, -2] = d1
    b[-1] = (h[-1] ** 2 * m[-2] + (2 * d1 + h[-1]) * h[-2] * m[-1]) / d1
    slopes = np.linalg.solve(A, b.reshape(n, -1)).reshape(y.shape)
    return _hermite_to_power(y, slopes, h)


def notaknot(x, y, clamp=True):
    return PiecewiseCubic(x, notaknot_coefficients(x, y), clamp=clamp)


def pchip_coefficients(x, y):
    """Fritsch-Carlson monotone cubic Hermite pieces along axis 0 of ``y`` (n, ...)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = len(x)
    h = np.diff(x)
    extra = (1,) * (y.ndim - 1)
    hh = h.reshape(h.shape + extra)
    m = (y[1:] - y[:-1]) / hh
    if n == 2:
        d = np.stack([m[0], m[0]])
        return _hermite_to_power(y, d, h)
    d = np.zeros_like(y)
    sm = np.sign(m)
    flat = (sm[1:] != sm[:-1]) | (m[1:] == 0) | (m[:-1] == 0)
    w1 = 2 * hh[1:] + hh[:-1]
    w2 = hh[1:] + 2 * hh[:-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        whmean = (w1 / m[:-1] + w2 / m[1:]) / (w1 + w2)
        inner = 1.0 / whmean
    inner[flat] = 0.0
    d[1:-1] = inner

    def edge(h0, h1, m0, m1):
        e = ((2 * h0 + h1) * m0 - h0 * m1) / (h0 + h1)
        e = np.where(np.sign(e) != np.sign(m0), 0.0, e)
        big = (np.sign(m0) != np.sign(m1)) & (np.abs(e) > 3 * np.abs(m0))
        return np.where(big, 3 * m0, e)

    d[0] = edge(h[0], h[1], m[0], m[1])
    d[-1] = edge(h[-1], h[-2], m[-1], m[-2])
    return _hermite_to_power(y, d, h)


def pchip(x, y):
    return PiecewiseCubic(x, pchip_coefficients(x, y), clamp=False)


# --------------------------------------------------------------------------- #
# quadrature / projection weights
# --------------------------------------------------------------------------- #
def legendre_values(ell, mu):
    """P_ell(mu) for ell = 0..4 (closed forms)."""
    mu = np.asarray(mu, dtype=np.float64)
    if ell == 0:
        return np.ones_like(mu)
    if ell == 1:
        return mu
    if ell == 2:
        return 1.5 * mu ** 2 - 0.5
    if ell == 3:
        return 2.5 * mu ** 3 - 1.5 * mu
    if ell == 4:
        return (35 * mu ** 4 - 30 * mu ** 2 + 3) / 8
    raise ValueError("multipoles above ell=4 are not supported")


#: accepted spellings of the even-N Simpson convention -> canonical name (see :func:`simpson_weights`)
SIMPSON_EVEN = {"simpson": "simpson", "scipy>=1.11": "simpson", "avg": "avg", "scipy<1.11": "avg",
                "first": "first", "last": "last"}
SIMPSON_EVEN_DEFAULT = "simpson"


def simpson_even_rule(name=None):
    """Canonical name of an even-N Simpson convention (``None`` -> the default); ``ValueError`` if unknown."""
    key = SIMPSON_EVEN_DEFAULT if name is None else str(name).strip().lower().replace(" ", "")
    if key not in SIMPSON_EVEN:
        raise ValueError(f"unknown simpson_even rule '{name}': choose from {sorted(SIMPSON_EVEN)}")
    return SIMPSON_EVEN[key]


def _composite_simpson(m):
    """Unit-spacing composite Simpson weights on an odd number ``m`` of points: (1, 4, 2, ..., 4, 1) / 3."""
    w = np.full(m, 2.0 / 3.0)
    w[1::2] = 4.0 / 3.0
    w[0] = w[-1] = 1.0 / 3.0
    return w


def simpson_weights(n, even=None):
    """Unit-spacing weights w with ``simps(y, dx=1) == w @ y`` on the velocity nodes of ``ccf_model.py:570,690``.

    Odd n: composite Simpson.  Even n (the reference has n = 50): the reference calls ``scipy.integrate.simps`` with its
    default ``even=`` argument and pins only ``scipy>=1.6.3`` (``setup.py:31``), so the rule depends on the SciPy it runs on:

    ``'simpson'`` (alias ``'scipy>=1.11'``; the default here)
        composite Simpson on the first n-1 points plus the three-point correction of the last interval
        (5/12, 2/3, -1/12) - SciPy >= 1.11;
    ``'avg'`` (alias ``'scipy<1.11'``)
        the mean of ``'first'`` and ``'last'`` - SciPy < 1.11, the rule behind the numbers printed in the reference's
        notebook (``notebooks/victor_usage_demo.ipynb:491-499``);
    ``'first'`` / ``'last'``
        composite Simpson on the first / last n-1 points and a trapezoid on the remaining end interval (the other two
        values of ``even=`` in SciPy < 1.11; not reachable from the reference, which never passes ``even=``).
    """
    if n < 3:
        raise ValueError("need at least 3 points")
    rule = simpson_even_rule(even)
    if n % 2 == 1:
        return _composite_simpson(n)
    w = np.zeros(n)
    if rule == "simpson":
        w[: n - 1] = _composite_simpson(n - 1)
        w[n - 1] += 5.0 / 12.0
        w[n - 2] += 2.0 / 3.0
        w[n - 3] -= 1.0 / 12.0
        return w
    first = np.zeros(n)
    first[: n - 1] = _composite_simpson(n - 1)
    first[n - 2:] += 0.5
    if rule == "first":
        return first
    last = first[::-1].copy()
    if rule == "last":
        return last
    return 0.5 * (first + last)


def projection_weights(mu_nodes, poles, npts=200):
    """W[l, i] with  xi_l(s_j) = sum_i W[l, i] xi(mu_i, s_j).

    The reference fits a bicubic interpolant to xi(s, mu) on the (s, mu_nodes) grid
    (``ccf_model.py:824``) and integrates it against (2l+1) P_l(mu) with a 200-point
    trapezoid rule at the same s nodes (``utils.py:45-56``).  At a node s_j a tensor-product
    interpolating spline reduces to the 1-D not-a-knot spline in mu through that column,
    so the whole operation is a fixed linear map of the column.
    """
    mu_nodes = np.asarray(mu_nodes, dtype=np.float64)
    poles = np.atleast_1d(poles)
    even = not np.any(poles % 2)
    if even:
        mu = np.linspace(0.0, 1.0, npts)
        factors = [2 * l + 1 for l in poles]
    else:
        mu = np.linspace(-1.0, 1.0, npts)
        factors = [(2 * l + 1) / 2 for l in poles]
    basis = notaknot(mu_nodes, np.eye(len(mu_nodes)))(mu)  # (npts, n_nodes)
    tw = np.zeros(npts)
    dmu = np.diff(mu)
    tw[:-1] += 0.5 * dmu
    tw[1:] += 0.5 * dmu
    W = np.empty((len(poles), len(mu_nodes)))
    for a, l in enumerate(poles):
        W[a] = factors[a] * ((tw * legendre_values(int(l), mu)) @ basis)
    return W


def mu_nodes_for(poles, n=100):
    """ccf_model.py:816-822."""
    poles = np.atleast_1d(poles)
    if np.any(poles % 2):
        return np.linspace(-1, 1, n)
    return np.linspace(0, 1, n)


# --------------------------------------------------------------------------- #
# helpers for the device table layout
# --------------------------------------------------------------------------- #
def uniform_spacing(knots, lead=0, rtol=1e-9):
    """Return the spacing of knots[lead:] if they are uniform, else None."""
    k = np.asarray(knots, dtype=np.float64)[lead:]
    if len(k) < 2:
        return None
    d = np.diff(k)
    h = (k[-1] - k[0]) / (len(k) - 1)
    if np.all(np.abs(d - h) <= rtol * abs(h)):
        return h
    return None


def spline_table_from_beta_poly(r, nodal_poly):
    """Spline coefficients of nodal values given as polynomials in beta.

    ``nodal_poly`` (n_beta-1, 4[p], n_r): nodal value at r_n on beta interval k is sum_p nodal_poly[k, p, n]
    (beta - beta_k)^p.  Returns (n_beta-1, n_r-1, 4[q], 4[p]): coefficient of (u - r_i)^q (beta - beta_k)^p.
    Exact, because the not-a-knot spline is linear in its nodal values.
    """
    lin = notaknot_coefficients(r, np.eye(len(r)))              # (n_r-1, 4[q], n_r)
    return np.einsum("iqn,kpn->kiqp", lin, np.asarray(nodal_poly, dtype=np.float64))


def near_uniform_spacing(knots, lead=0, tol=0.4):
    """Mean spacing of knots[lead:] if every knot lies within ``tol`` spacings of its uniform position, else None."""
    k = np.asarray(knots, dtype=np.float64)[lead:]
    if len(k) < 2:
        return None
    h = (k[-1] - k[0]) / (len(k) - 1)
    if np.all(np.abs(k - (k[0] + h * np.arange(len(k)))) <= tol * h):
        return h
    return None


def beta_dependent_spline_table(r, beta, values):
    """Coefficients of the r-spline of PCHIP-in-beta nodal values, as polynomials in beta.

    ``values`` (n_beta, n_r).  Composes ``ccf_model.py:323-326`` (PCHIP over beta) with ``:619-621`` (spline in r).
    """
    return spline_table_from_beta_poly(r, pchip_coefficients(beta, np.asarray(values, dtype=np.float64)))


def trapezoid_weights(x):
    """w with ``np.trapz(y, x) == w @ y``."""
    x = np.asarray(x, dtype=np.float64)
    w = np.zeros_like(x)
    d = np.diff(x)
    w[:-1] += 0.5 * d
    w[1:] += 0.5 * d
    return w


def bicubic_patches(x, y, z):
    """Power-basis patches of the tensor-product not-a-knot spline through ``z`` (len(x), len(y)).

    Returns (len(x)-1, len(y)-1, 4, 4): coefficient of (u - x_i)^p (v - y_j)^q.  This is what
    ``RectBivariateSpline(x, y, z)`` (kx = ky = 3, s = 0) interpolates with: FITPACK places its knots on the data
    points minus the 2nd and 2nd-last in each direction, i.e. the not-a-knot spline along each axis.
    """
    lx = notaknot_coefficients(x, np.eye(len(x)))      # (nx-1, 4, nx)
    ly = notaknot_coefficients(y, np.eye(len(y)))      # (ny-1, 4, ny)
    return np.einsum("ipn,jqm,nm->ijpq", lx, ly, np.asarray(z, dtype=np.float64))


# --------------------------------------------------------------------------- #
# one refined grid for several piecewise-cubic tables (single interval index on the device)
# --------------------------------------------------------------------------- #
def common_refinement(knots_a, knots_b, max_intervals=512, max_den=64, tol=1e-9):
    """A uniform grid (u0, h, n) that contains every knot of two uniform knot sets, or None.

    Both sets must be uniform and commensurate (spacings and offset in small integer ratios, e.g. r = 2, 6, 10, ...
    with r_sv = 3, 9, 15, ... -> h = 1).  Every interval of the refined grid then lies inside exactly one interval
    of each table (or outside its range), so one index serves both tables.
    """
    from fractions import Fraction
    ka, kb = np.asarray(knots_a, dtype=np.float64), np.asarray(knots_b, dtype=np.float64)
    ha, hb = uniform_spacing(ka), uniform_spacing(kb)
    if ha is None or hb is None:
        return None
    fb = Fraction(hb / ha).limit_denominator(max_den)
    fo = Fraction((kb[0] - ka[0]) / ha).limit_denominator(max_den)
    if abs(float(fb) - hb / ha) > tol or abs(float(fo) - (kb[0] - ka[0]) / ha) > tol:
        return None
    den = np.lcm(fb.denominator, fo.denominator)
    num = np.gcd(np.gcd(den, int(fb * den)), int(abs(fo) * den)) if fo != 0 else np.gcd(den, int(fb * den))
    h = ha * float(num) / float(den)
    u0 = min(ka[0], kb[0])
    u1 = max(ka[-1], kb[-1])
    n = int(round((u1 - u0) / h))
    if n < 1 or n > max_intervals:
        return None
    for k in (ka, kb):                                  # every knot must sit on the refined grid
        pos = (k - u0) / h
        if np.max(np.abs(pos - np.round(pos))) > 1e-7:
            return None
    return float(u0), float(h), n


def refine_pp(knots, coef, u0, h, n):
    """Re-express a clamped piecewise cubic on the refined grid, in interval units.

    ``coef`` (n_int, 4, ...) about the left knots of ``knots``.  Returns (n, 4, ...): on refined interval q the
    function is sum_p out[q, p] tau^p with tau = (u - u0 - q h)/h in [0, 1).  Refined intervals outside the table's
    range hold the (constant) boundary value, which reproduces the clamped evaluation exactly.
    """
    left = u0 + h * np.arange(n)
    return refine_pp_on(knots, coef, left, np.full(n, float(h)))


def refine_pp_on(knots, coef, left, width):
    """``refine_pp`` for arbitrary target intervals [left_q, left_q + width_q], each inside one interval of ``knots``
    (or outside its range): coefficients of tau^p, tau = (u - left_q)/width_q."""
    knots = np.asarray(knots, dtype=np.float64)
    coef = np.asarray(coef, dtype=np.float64)
    left = np.asarray(left, dtype=np.float64)
    width = np.asarray(width, dtype=np.float64)
    mid = left + 0.5 * width
    i = np.clip(np.searchsorted(knots, mid, side="right") - 1, 0, len(knots) - 2)
    d = left - knots[i]
    extra = (1,) * (coef.ndim - 2)
    dd = d.reshape(d.shape + extra)
    w = width.reshape(width.shape + extra)
    c0, c1, c2, c3 = coef[i, 0], coef[i, 1], coef[i, 2], coef[i, 3]
    out = np.stack([((c3 * dd + c2) * dd + c1) * dd + c0,
                    ((3 * c3 * dd + 2 * c2) * dd + c1) * w,
                    (3 * c3 * dd + c2) * w ** 2,
                    c3 * w ** 3], axis=1)
    below = mid < knots[0]
    above = mid > knots[-1]
    if np.any(below):
        out[below] = 0.0
        out[below, 0] = coef[0, 0]
    if np.any(above):
        hl = knots[-1] - knots[-2]
        end = ((coef[-1, 3] * hl + coef[-1, 2]) * hl + coef[-1, 1]) * hl + coef[-1, 0]
        out[above] = 0.0
        out[above, 0] = end
    return out


def union_grid(*knot_sets, max_cells=4096):
    """Union of several knot sets plus a uniform look-up table that locates its intervals with two comparisons.

    Returns ``(U, lut, inv_g)`` or None.  ``U`` (n+1,) are the sorted distinct knots.  The table has G cells of width
    1/inv_g over [0, U[-1]) plus one guard cell; every cell holds at most TWO interior knots of U (two families of
    knots may interleave arbitrarily closely; a cell narrower than the spacing within each family never sees three),
    and ``lut[c]`` (uint16) is the index of the interval that contains the cell's left edge, so the interval of u in
    [U[0], U[-1]) is ``q = lut[int(u * inv_g)];  q += (u >= U[q + 1]) + (u >= U[q + 2])`` (U padded with U[-1]).
    """
    U = np.unique(np.concatenate([np.asarray(k, dtype=np.float64) for k in knot_sets]))
    n = len(U) - 1
    if n < 1 or n > 4000 or U[0] < 0:
        return None
    inner = U[1:-1]
    span = U[-1]
    G = 64
    while True:
        edges = span * np.arange(G + 1) / G
        # knots sitting exactly on a cell's left edge belong to the index, not to the cell
        count = np.searchsorted(inner, edges[1:], side="left") - np.searchsorted(inner, edges[:-1], side="right")
        if np.all(count <= 2):
            break
        G *= 2
        if G > max_cells:
            return None
    # one extra cell: u is clamped just below U[-1], but the rounded product u * inv_g can still reach G
    lut = np.searchsorted(inner, edges, side="right").astype(np.uint16)
    return U, lut, G / span
