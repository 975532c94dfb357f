"""Matter / velocity profile tables for the kernels, built from plain model attributes.

Everything here is duck-typed on the attributes both this package's ``CCFModel`` and the reference's own
``victor.CCFModel`` carry (``r``, ``beta``, ``real_multipoles``, ``fixed_real_input``, ``delta``,
``integrated_delta``, ``radial_velocity``, ``get_interpolated_real_multipoles``; reference: ccf_model.py:99-326), so
that ``victor_amd.engine.build_tables`` can compile the device tables from either object (route B).
"""

import numpy as np

from . import tables as T
from .utils import InputError


def linear_bias_maps(model, r_nodes):
    """Matrices (Bd, Td) with  b*delta(r_nodes) = Bd @ y  and  b*Delta(r_nodes) = Td @ y  for nodal values y of
    the real-space monopole on ``model.r``: the reference's spline of xi^r_0 and its 100-point trapezoid integral
    3/(b r^3) int_0^r xi x^2 dx (ccf_model.py:358-370) are both linear in y."""
    r_nodes = np.asarray(r_nodes, dtype=float)
    r = np.asarray(model.r, dtype=float)
    basis = T.notaknot(r, np.eye(len(r)))
    Bd = basis(r_nodes)
    Td = np.empty_like(Bd)
    for n, rn in enumerate(r_nodes):
        x = np.linspace(0, rn, 100)
        Td[n] = 3.0 / rn ** 3 * ((T.trapezoid_weights(x) * x ** 2) @ basis(x))
    return Bd, Td


def matter_nodal(model, matter_model, r_nodes, beta=None):
    """(delta, Delta) at ``r_nodes`` with the 1/bias factors left out (they are per-point amplitudes)."""
    if matter_model == "template":
        return np.asarray(model.delta(r_nodes)), np.asarray(model.integrated_delta(r_nodes))
    if matter_model == "linear_bias":
        y = model.get_interpolated_real_multipoles(beta)[0]
        Bd, Td = linear_bias_maps(model, r_nodes)
        return Bd @ y, Td @ y
    if matter_model == "excursion_set":
        raise InputError("matter_model 'excursion_set' is not implemented in victor_amd")
    raise InputError(f"Invalid choice of matter_model {matter_model}")


def empirical_gradient_tables(model, r_ext, delta_ext, int_delta_ext):
    """Nodal values (at r_ext) of the two numerical-derivative pieces of the empirical_corr branch
    (ccf_model.py:455-459): d/dr of r*Delta and of r*Delta*delta on linspace(0.1, r_max, 100)."""
    rg = np.linspace(0.1, np.max(model.r), 100)
    Ds = T.notaknot(r_ext, int_delta_ext)(rg)
    ds = T.notaknot(r_ext, delta_ext)(rg)
    g1 = T.notaknot(rg, np.gradient(rg * Ds, rg))(r_ext)
    g2 = T.notaknot(rg, np.gradient(rg * Ds * ds, rg))(r_ext)
    return g1, g2


def _poly_mul(a, b):
    """Product of polynomials stored along axis 1: (k, pa, n) x (k, pb, n) -> (k, pa + pb - 1, n)."""
    out = np.zeros((a.shape[0], a.shape[1] + b.shape[1] - 1, a.shape[2]))
    for i in range(a.shape[1]):
        for j in range(b.shape[1]):
            out[:, i + j] += a[:, i] * b[:, j]
    return out


def empirical_beta_tables(model):
    """beta-dependent counterpart of V2, Ge1, Ge2 for linear_bias on a reconstructed real-space ccf.

    delta and Delta at the nodes are PCHIP cubics in (beta - beta_k); every later step of ccf_model.py:451-459
    (splines onto the gradient grid, np.gradient, splines back) is linear, so V2 = r Delta delta and Ge2 are
    polynomials of degree 6 and Ge1 one of degree 3 (stored with zero high coefficients).
    Returns (3, n_beta-1, n_r, 4, 7): spline coefficient of (u - r_i)^q times (beta - beta_k)^p.
    """
    r_ext = np.append([0.01], np.asarray(model.r, dtype=float))
    Bd, Td = linear_bias_maps(model, r_ext)
    ypoly = T.pchip_coefficients(model.beta, model.real_multipoles["0"])       # (n_beta-1, 4, n_r)
    d_poly = np.einsum("nm,kpm->kpn", Bd, ypoly)
    D_poly = np.einsum("nm,kpm->kpn", Td, ypoly)
    rg = np.linspace(0.1, np.max(model.r), 100)
    to_rg = T.notaknot(r_ext, np.eye(len(r_ext)))(rg)                          # (100, n_ext)
    back = T.notaknot(rg, np.gradient(np.eye(len(rg)), rg, axis=0))(r_ext)     # (n_ext, 100): gradient, then spline
    Ds = np.einsum("gn,kpn->kpg", to_rg, D_poly)
    ds = np.einsum("gn,kpn->kpg", to_rg, d_poly)
    g1 = np.einsum("ng,kpg->kpn", back, rg * Ds)                               # degree 3
    g2 = np.einsum("ng,kpg->kpn", back, rg * _poly_mul(Ds, ds))                # degree 6
    v2 = r_ext * _poly_mul(D_poly, d_poly)
    g1 = np.concatenate([g1, np.zeros((g1.shape[0], 3, g1.shape[2]))], axis=1)
    return np.stack([T.spline_table_from_beta_poly(r_ext, v2),
                     T.spline_table_from_beta_poly(r_ext, g1 / 3),
                     T.spline_table_from_beta_poly(r_ext, g2 / 3)])


def velocity_tables(model, matter_model):
    """Coefficient arrays of the velocity tables on r_ext = [0.01, r...] for one matter model.

    Returns ``(coef, beta_dependent)``: fixed -> (5, n_r, 4) for V1 = r*Delta, Da = delta - 2 Delta/3,
    V2 = r*Delta*delta, Ge1, Ge2; beta-dependent -> (2, n_beta-1, n_r, 4, 4) for V1, Da (the other three:
    ``empirical_beta_tables``).
    """
    r_ext = np.append([0.01], np.asarray(model.r, dtype=float))
    if matter_model == "velocity_template":
        # v_r = growth_t * V_t(r/c); its derivative by the reference's numerical gradient (ccf_model.py:486-490)
        rg = np.linspace(0.1, np.max(model.r), 100)
        gt = T.notaknot(rg, np.gradient(model.radial_velocity(rg), rg))(r_ext)
        zero = np.zeros_like(r_ext)
        nodal = np.stack([np.asarray(model.radial_velocity(r_ext)), gt / 3, zero, zero, zero], axis=1)
        return np.moveaxis(T.notaknot_coefficients(r_ext, nodal), 2, 0), False
    if matter_model == "linear_bias" and not model.fixed_real_input:
        Bd, Td = linear_bias_maps(model, r_ext)
        ypoly = T.pchip_coefficients(model.beta, model.real_multipoles["0"])       # (n_beta-1, 4, n_r)
        d_poly = np.einsum("nm,kpm->kpn", Bd, ypoly)
        D_poly = np.einsum("nm,kpm->kpn", Td, ypoly)
        V1 = T.spline_table_from_beta_poly(r_ext, r_ext * D_poly)
        Da = T.spline_table_from_beta_poly(r_ext, d_poly - 2 * D_poly / 3)
        return np.stack([V1, Da]), True
    d, D = matter_nodal(model, matter_model, r_ext)
    g1, g2 = empirical_gradient_tables(model, r_ext, d, D)
    nodal = np.stack([r_ext * D, d - 2 * D / 3, r_ext * D * d, g1 / 3, g2 / 3], axis=1)   # (n_ext, 5)
    return np.moveaxis(T.notaknot_coefficients(r_ext, nodal), 2, 0), False              # (5, n_r, 4)


def dispersion_is_isotropic(model):
    """True when sigma_v(r, mu) carries no mu dependence (2-key template: every mu row identical)."""
    flag = getattr(model, "_sv_isotropic", None)
    if flag is not None:
        return bool(flag)
    sv = np.asarray(model.sv_rmu)
    return bool(np.all(sv == sv[:1]))
