"""Helpers mirroring ``victor/utils.py`` for the parts the likelihood path touches."""

import numpy as np

from . import tables as T

_trapz = getattr(np, "trapezoid", None) or np.trapz


class InputError(Exception):
    """Error raised when something is wrong with the input data (reference: utils.py:5)."""


def multipoles_from_fn(frmu, r, ell=[0, 2, 4], even=True, npts=200):
    """Legendre multipoles of a callable f(r, mu) at the radii ``r`` (reference: utils.py:9-58).

    ``frmu(r_j, mu)`` must return the function on the 1-D array ``mu``.  On the likelihood path this
    projection runs inside the HIP kernel as fixed weights (:func:`victor_amd.tables.projection_weights`);
    this host version serves the set-up steps that call it once (dispersion template normalisation,
    ``format: rmu`` real-space input).
    """
    try:
        len(ell)
    except TypeError:
        ell = np.array([ell])
    out = {f"{l}": np.zeros(len(r)) for l in ell}
    if even:
        mu = np.linspace(0.0, 1.0, npts)
        factors = [2 * l + 1 for l in ell]
    else:
        mu = np.linspace(-1, 1, npts)
        factors = [(2 * l + 1) / 2 for l in ell]
    for i, l in enumerate(ell):
        lmu = T.legendre_values(int(l), mu)
        for j in range(len(r)):
            y = np.asarray(frmu(r[j], mu)).reshape(-1)
            out[f"{l}"][j] = factors[i] * _trapz(y * lmu, mu)
    return out


def bilinear_on_grid(x, y, z):
    """f(x_j, y*) for z tabulated on (y, x): linear in both directions (``interp2d`` default kind).

    Returns a callable ``f(xq, yq)`` -> array over ``yq`` at scalar ``xq``; arguments outside the grid
    are clamped to it, as FITPACK does for regular-grid splines.
    """
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    z = np.asarray(z, dtype=float)  # (len(y), len(x))

    def f(xq, yq):
        xq = float(np.clip(xq, x[0], x[-1]))
        col = np.array([np.interp(xq, x, z[k]) for k in range(len(y))])
        return np.interp(np.clip(yq, y[0], y[-1]), y, col)

    return f


class GridInterpolant:
    """Bilinear interpolant of ``z`` tabulated on ``(y, x)`` with the call convention of
    ``scipy.interpolate.interp2d(x, y, z)`` (default ``kind='linear'``; removed from SciPy 1.14): ``f(xq, yq)`` with
    scalars or sorted 1-D arrays returns shape ``(len(yq), len(xq))`` (squeezed for scalars); queries outside the
    grid take the nearest edge value.  Used for the 2-D model grids of ccf_model.py:862-934."""

    def __init__(self, x, y, z):
        self.x = np.asarray(x, dtype=float)
        self.y = np.asarray(y, dtype=float)
        self.z = np.asarray(z, dtype=float)
        if self.z.shape != (len(self.y), len(self.x)):
            raise InputError("GridInterpolant: z must have shape (len(y), len(x))")

    @staticmethod
    def _weights(grid, q):
        q = np.clip(q, grid[0], grid[-1])
        i = np.clip(np.searchsorted(grid, q, side="right") - 1, 0, len(grid) - 2)
        t = (q - grid[i]) / (grid[i + 1] - grid[i])
        return i, t

    def __call__(self, xq, yq):
        xs, ys = np.atleast_1d(np.asarray(xq, dtype=float)), np.atleast_1d(np.asarray(yq, dtype=float))
        i, tx = self._weights(self.x, xs)
        j, ty = self._weights(self.y, ys)
        z = self.z
        lo = z[j][:, i] * (1 - tx) + z[j][:, i + 1] * tx
        hi = z[j + 1][:, i] * (1 - tx) + z[j + 1][:, i + 1] * tx
        out = lo * (1 - ty)[:, None] + hi * ty[:, None]
        if np.ndim(xq) == 0 and np.ndim(yq) == 0:
            return out[0]                       # interp2d returns a length-1 array for scalar queries
        if np.ndim(xq) == 0:
            return out[:, 0]
        if np.ndim(yq) == 0:
            return out[0]
        return out


def fn_from_multipoles(r, poles, multipoles, npts=200):
    """f(r, mu) = sum_l multipoles[l](r) P_l(mu) tabulated on ``npts`` values of mu in [-1, 1], as a bilinear
    interpolant with the ``interp2d`` call convention (reference: utils.py:60-94)."""
    from .tables import legendre_values
    poles = [poles] if isinstance(poles, int) else poles
    multipoles = np.asarray(multipoles, dtype=float)
    if multipoles.shape != (len(poles), len(r)):
        raise ValueError(f"Wrong shape of multipoles: expected ({len(poles)}, {len(r)}), "
                         f"but received {multipoles.shape}")
    mu = np.linspace(-1, 1, npts)
    grid = np.zeros((len(mu), len(r)))
    for i, ell in enumerate(poles):
        grid += legendre_values(int(ell), mu)[:, None] * multipoles[i]
    return GridInterpolant(r, mu, grid)


def read_input_file(path, extensions):
    """ccf_model.py:57-68: choose the reader from the file extension (npy dict or HDF5)."""
    fmt = None
    for file_format, exts in extensions.items():
        if any(path.endswith(ext) for ext in exts):
            fmt = file_format
            break
    if fmt == "npy":
        return np.load(path, allow_pickle=True).item()
    if fmt == "hdf5":
        from . import h5lite
        try:
            return h5lite.read_all(path)
        except h5lite.H5LiteError as exc:
            raise InputError(f"Cannot read {path}: {exc}")
    raise InputError(f"Unrecognised format of input file {path}")
