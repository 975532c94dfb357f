"""Host table compiler vs the SciPy primitives the reference calls (CPU only)."""
import numpy as np
import pytest
import scipy.interpolate as si
from scipy.integrate import simpson
from scipy.special import legendre

from victor_amd import tables as T


def _grid(rng, n, uniform):
    if uniform:
        return 2.0 + 4.0 * np.arange(n)
    return np.cumsum(rng.uniform(0.5, 3.0, n))


@pytest.mark.parametrize("n", [4, 5, 25, 30, 31, 55])
@pytest.mark.parametrize("uniform", [True, False])
def test_notaknot_matches_fitpack_ius(n, uniform):
    rng = np.random.default_rng(n)
    x = _grid(rng, n, uniform)
    y = rng.normal(size=n)
    ours = T.notaknot(x, y)
    ref = si.InterpolatedUnivariateSpline(x, y, ext=3)
    u = np.concatenate([np.linspace(x[0] - 5, x[-1] + 5, 2001), x])
    assert np.max(np.abs(ours(u) - ref(u))) <= 5e-13 * max(1.0, np.max(np.abs(y)))


def test_notaknot_extra_leading_node_and_vector_valued():
    x = np.append([0.01], 2.0 + 4.0 * np.arange(30))
    rng = np.random.default_rng(3)
    Y = rng.normal(size=(31, 3))
    ours = T.notaknot(x, Y)
    u = np.linspace(0, 130, 777)
    for k in range(3):
        ref = si.InterpolatedUnivariateSpline(x, Y[:, k], ext=3)
        assert np.max(np.abs(ours(u)[:, k] - ref(u))) < 1e-12


def test_spline_abscissa_rescaling_identity():
    # ccf_model.py:613-621: spline(c*r, y)(x) == spline(r, y)(x/c)
    r = 2.0 + 4.0 * np.arange(30)
    y = np.sin(r / 9.0)
    c = 1.0375
    u = np.linspace(0, 140, 500)
    a = si.InterpolatedUnivariateSpline(c * r, y, ext=3)(u)
    b = T.notaknot(r, y)(u / c)
    assert np.max(np.abs(a - b)) < 1e-13


@pytest.mark.parametrize("seed", range(6))
def test_pchip_matches_scipy(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 32))
    x = np.cumsum(rng.uniform(0.01, 0.03, n)) + 0.15
    y = rng.normal(size=(n, 7))
    y[:, 0] = np.sort(y[:, 0])          # monotone column
    y[:, 1] = 0.3                       # flat column
    if n > 4:
        y[2:4, 2] = y[2, 2]             # zero slope segment
    ours = T.pchip(x, y)
    ref = si.PchipInterpolator(x, y, axis=0)
    u = np.concatenate([np.linspace(x[0] - 0.05, x[-1] + 0.05, 501), x])
    assert np.max(np.abs(ours(u) - ref(u))) < 1e-12


@pytest.mark.parametrize("n", [3, 5, 49, 50, 51, 8])
def test_simpson_weights_match_scipy(n):
    rng = np.random.default_rng(n)
    y = rng.normal(size=n)
    x = np.linspace(-6, 6, n) * 380.0
    h = x[1] - x[0]
    assert abs(T.simpson_weights(n) @ y * h - simpson(y, x=x)) < 1e-10 * h


@pytest.mark.parametrize("n", [4, 8, 50, 51])
def test_simpson_weights_legacy_even_rules(n):
    """SciPy < 1.11 ``simps(even=...)``: 'first' / 'last' integrate quadratics exactly on n-2 intervals and use a
    trapezoid on the remaining end interval; 'avg' is their mean (the rule the reference's notebook numbers come from)."""
    x = np.linspace(-6, 6, n)
    h = x[1] - x[0]
    first, last, avg = (T.simpson_weights(n, r) for r in ("first", "last", "avg"))
    assert np.array_equal(T.simpson_weights(n, "scipy<1.11"), avg)
    assert np.array_equal(T.simpson_weights(n, "scipy>=1.11"), T.simpson_weights(n))
    if n % 2:
        assert np.array_equal(first, T.simpson_weights(n)) and np.array_equal(avg, first)
        return
    assert np.allclose(avg, 0.5 * (first + last), rtol=0, atol=1e-16) and np.allclose(last, first[::-1])
    for w in (first, last, avg):
        assert abs(w.sum() - (n - 1)) < 1e-13                      # constants exactly
        assert abs(w @ x * h) < 1e-12                              # straight lines exactly
    q = 3 * x ** 2 - x + 1
    exact = lambda a, b: (b ** 3 - b ** 2 / 2 + b) - (a ** 3 - a ** 2 / 2 + a)   # noqa: E731
    trap = lambda i: 0.5 * h * (q[i] + q[i + 1])                                  # noqa: E731
    assert abs(first @ q * h - (exact(x[0], x[-2]) + trap(n - 2))) < 1e-10
    assert abs(last @ q * h - (trap(0) + exact(x[1], x[-1]))) < 1e-10
    with pytest.raises(ValueError):
        T.simpson_weights(n, "middle")


@pytest.mark.parametrize("poles", [[0, 2], [0, 2, 4], [0], [1, 3], [0, 1, 2]])
def test_projection_weights_match_reference_construction(poles):
    """Same construction as ccf_model.py:822-825 with FITPACK bicubic + utils.py:45-56."""
    rng = np.random.default_rng(1)
    s = 2.0 + 4.0 * np.arange(12)
    mu = T.mu_nodes_for(poles)
    xi = rng.normal(size=(len(mu), len(s))) * 0.1 + np.cos(3 * mu)[:, None] * np.exp(-s / 50)[None, :]
    W = T.projection_weights(mu, poles)
    got = W @ xi
    spl = si.RectBivariateSpline(s, mu, xi.T, kx=3, ky=3, s=0)
    even = not np.any(np.asarray(poles) % 2)
    m200 = np.linspace(0, 1, 200) if even else np.linspace(-1, 1, 200)
    for a, l in enumerate(poles):
        f = (2 * l + 1) if even else (2 * l + 1) / 2
        for j, sj in enumerate(s):
            want = f * np.trapz(spl(sj, m200)[0] * legendre(l)(m200), m200)
            assert abs(got[a, j] - want) < 2e-13


def test_projection_weight_sums_are_not_exact():
    # the "-1" of ccf_model.py:690 projects to -sum(W_l), not to zero
    W = T.projection_weights(np.linspace(0, 1, 100), [0, 2, 4])
    assert abs(W[0].sum() - 1) < 1e-13
    assert 1e-5 < W[1].sum() < 1e-4
    assert 1e-4 < W[2].sum() < 1e-3


def test_beta_dependent_table_equals_pchip_then_spline():
    rng = np.random.default_rng(5)
    r = 2.0 + 4.0 * np.arange(30)
    beta = np.linspace(0.16, 0.65, 31)
    vals = rng.normal(size=(31, 30)).cumsum(axis=0) * 0.01
    tab = T.beta_dependent_spline_table(r, beta, vals)
    for b in (0.37, 0.16, 0.2004, 0.649, 0.1, 0.7):
        k = int(np.clip(np.searchsorted(beta, b, side="right") - 1, 0, len(beta) - 2))
        db = b - beta[k]
        coef = ((tab[k, :, :, 3] * db + tab[k, :, :, 2]) * db + tab[k, :, :, 1]) * db + tab[k, :, :, 0]
        nodal = si.PchipInterpolator(beta, vals, axis=0)(b)
        ref = si.InterpolatedUnivariateSpline(r, nodal, ext=3)
        u = np.linspace(0, 125, 400)
        ours = T.PiecewiseCubic(r, coef)(u)
        assert np.max(np.abs(ours - ref(u))) < 1e-12


def test_legendre_closed_forms():
    mu = np.linspace(-1, 1, 33)
    for l in range(5):
        assert np.max(np.abs(T.legendre_values(l, mu) - legendre(l)(mu))) < 1e-14


def test_uniform_spacing_detection():
    assert T.uniform_spacing(2.0 + 4.0 * np.arange(30)) == pytest.approx(4.0)
    assert T.uniform_spacing(np.append([0.01], 2.0 + 4.0 * np.arange(30))) is None
    assert T.uniform_spacing(np.append([0.01], 2.0 + 4.0 * np.arange(30)), lead=1) == pytest.approx(4.0)


def test_near_uniform_spacing_detection():
    k = 0.05 + 0.12 * np.arange(25) + 0.01 * np.sin(np.arange(25))
    assert T.uniform_spacing(k) is None
    assert T.near_uniform_spacing(k) == pytest.approx((k[-1] - k[0]) / 24)
    assert T.near_uniform_spacing(np.array([0.0, 1.0, 2.0, 3.9, 4.0])) is None          # too irregular
    kk = np.append([0.001], k)
    assert T.near_uniform_spacing(kk, lead=1) == pytest.approx((k[-1] - k[0]) / 24)


def test_common_refinement_and_refined_tables():
    r = 2.0 + 4.0 * np.arange(30)            # BOSS-like xi grid
    rsv = 3.0 + 6.0 * np.arange(25)          # and dispersion grid
    u0, h, n = T.common_refinement(r, rsv)
    assert (u0, h, n) == (2.0, 1.0, 145)
    u0s, hs, ns = T.common_refinement(1.5 + 3.0 * np.arange(40), 3.0 + 6.0 * np.arange(25))
    assert (u0s, hs, ns) == (1.5, 1.5, 97)
    assert T.common_refinement(r, 3.3 + 6.1 * np.arange(25)) is None                 # incommensurate
    assert T.common_refinement(r + 0.01 * np.sin(np.arange(30)), rsv) is None        # not uniform
    rng = np.random.default_rng(0)
    for knots in (r, rsv, np.append([0.01], r)):
        y = rng.normal(size=(len(knots), 2))
        pc = T.notaknot(knots, y)
        ref = T.refine_pp(knots, pc.coef, u0, h, n)
        assert ref.shape == (n, 4, 2)
        u = rng.uniform(u0, u0 + n * h, 4000)
        q = np.minimum(((u - u0) / h).astype(int), n - 1)
        tau = (u - u0) / h - q
        val = ((ref[q, 3] * tau[:, None] + ref[q, 2]) * tau[:, None] + ref[q, 1]) * tau[:, None] + ref[q, 0]
        assert np.max(np.abs(val - pc(u))) < 1e-12 * max(1.0, np.max(np.abs(y)))     # incl. the clamped ranges
    # beta-polynomial coefficient arrays refine the same way (the shift is linear in the coefficients)
    beta = np.linspace(0.16, 0.65, 7)
    vals = rng.normal(size=(7, 30))
    tab = T.beta_dependent_spline_table(r, beta, vals)                   # (6, 29, 4, 4)
    ref = T.refine_pp(r, np.moveaxis(tab, 0, -1), u0, h, n)              # (n, 4, 4[p_beta], 6)
    k, db = 2, 0.03
    coef_r = ((tab[k, :, :, 3] * db + tab[k, :, :, 2]) * db + tab[k, :, :, 1]) * db + tab[k, :, :, 0]
    want = T.refine_pp(r, coef_r, u0, h, n)
    got = ((ref[:, :, 3, k] * db + ref[:, :, 2, k]) * db + ref[:, :, 1, k]) * db + ref[:, :, 0, k]
    assert np.max(np.abs(got - want)) < 1e-13


def test_union_grid_lookup_and_refinement():
    """Union-grid form of the unified tables: the u16 look-up table plus one comparison finds the interval of every
    u (knots and the clamped upper end included), and the re-expanded records reproduce the clamped splines."""
    from victor_amd import tables as T
    rng = np.random.default_rng(4)
    for trial in range(5):
        r = np.sort(np.append([0.01], 1.5 + 3 * np.arange(40) + rng.uniform(-0.9, 0.9, 40)))
        sv = 3 + 6 * np.arange(25) + rng.uniform(-2, 2, 25)
        U, lut, inv_g = T.union_grid(r, sv)
        assert len(U) == 66 and lut.dtype == np.uint16 and len(lut) <= 257
        top = U[-1] * (1 - 2.0 ** -52)
        u = np.concatenate([rng.uniform(U[0], top, 50000), U[:-1], [top], np.nextafter(U[1:-1], 0)])
        cell = (u * inv_g).astype(int)
        assert cell.max() < len(lut)
        q = lut[cell].astype(int)
        Up = np.append(U, U[-1])
        q = q + (u >= Up[q + 1]) + (u >= Up[q + 2])
        assert np.array_equal(q, np.searchsorted(U, u, side="right") - 1)
        y = rng.normal(size=len(sv))
        rec = T.refine_pp_on(sv, T.notaknot_coefficients(sv, y), U[:-1], np.diff(U))
        tau = (u - U[q]) / (U[q + 1] - U[q])
        got = ((rec[q, 3] * tau + rec[q, 2]) * tau + rec[q, 1]) * tau + rec[q, 0]
        want = T.notaknot(sv, y)(np.clip(u, sv[0], sv[-1]))
        assert np.max(np.abs(got - want)) < 1e-13 * np.max(np.abs(want))
    # commensurate uniform grids need few cells; coincident knots are merged
    U, lut, inv_g = T.union_grid(np.append([0.01], 2 + 4.0 * np.arange(30)), 3 + 6.0 * np.arange(25), [2.0, 6.0])
    assert len(U) == 56 and len(lut) <= 257
    assert T.union_grid([0.01, 1.0, 1.0 + 1e-7, 1.0 + 2e-7, 200.0], [5.0]) is None     # would need > 4096 cells
