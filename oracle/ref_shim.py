"""DEV-ONLY harness: import the unmodified reference from /root/reference.

TEST INFRASTRUCTURE, NOT PRODUCT.  Nothing under ``victor_amd/`` imports this.
It exists so that (1) ``oracle/victor_oracle.py`` (our CPU restatement) can be
pinned against the real reference in the development container and (2)
``oracle/make_golden.py`` can dump golden vectors into ``tests/golden/``.
On the GPU box ``/root/reference`` does not exist and :func:`available`
returns False; nothing in the ``-m gpu`` tests, ``smoke()`` or ``bench.py``
calls into this module.

The reference (victor 0.1.4) cannot be imported as-is in this image for
ordinary reasons: h5py and astropy are not installed,
``scipy.integrate.simps`` and ``scipy.interpolate.interp2d`` are gone from
SciPy 1.15.  Four stand-in modules/functions are registered before the import
and the reference source itself is left untouched:

1. ``h5py.File``  -> dict of arrays read with ``h5dump -b`` (bit-exact binary
   dump; falls back to our h5lite reader when h5dump is absent);
2. ``astropy.cosmology.LambdaCDM`` -> closed-form flat/curved LCDM ``H(z)``
   with ``Tcmb0 = 0`` (astropy's default), all that ``cosmology.py:41-45`` uses;
3. ``scipy.integrate.simps`` (removed in SciPy 1.14).  The reference calls it with the default
   ``even=`` on 50 nodes (ccf_model.py:690), and that default CHANGED in SciPy 1.11, so there are two
   legitimate stand-ins under the reference's pin (``scipy>=1.6.3``, setup.py:31) and the result differs
   between them by up to 2e-4 relative in chi2:
   ``'simpson'`` = this image's ``scipy.integrate.simpson`` (what ``simps`` does in SciPy 1.11-1.13), and
   ``'avg'``     = the SciPy < 1.11 default (``victor_oracle.simps_legacy``, a restatement of the documented
   rule: mean of {Simpson + end trapezoid} taken from either end), the convention behind the numbers printed in
   the reference's notebook.  :func:`set_simpson_rule` switches the stand-in (also after the import);
4. ``scipy.interpolate.interp2d`` -> ``RectBivariateSpline(x, y, z.T, kx, ky, s=0)``,
   SciPy's own documented replacement for regular grids (identical FITPACK fit).
"""

import os
import subprocess
import sys
import tempfile
import types
import warnings

import numpy as np

REFERENCE_ROOT = os.environ.get("VICTOR_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "victor", "ccf_model.py"))


# --------------------------------------------------------------------------- #
def _h5_read(fn):
    h5ls = "/opt/conda/bin/h5ls"
    h5dump = "/opt/conda/bin/h5dump"
    if os.path.isfile(h5ls) and os.path.isfile(h5dump):
        out = {}
        listing = subprocess.check_output([h5ls, fn], text=True)
        for line in listing.splitlines():
            name = line.split()[0]
            dims = line[line.index("{") + 1:line.index("}")]
            shape = tuple(int(x) for x in dims.replace(" ", "").split(",") if x and x != "SCALAR")
            with tempfile.NamedTemporaryFile(suffix=".bin") as tmp:
                subprocess.check_call([h5dump, "-d", "/" + name, "-b", "LE", "-o", tmp.name, fn],
                                      stdout=subprocess.DEVNULL)
                out[name] = np.fromfile(tmp.name, dtype="<f8").reshape(shape)
        return out
    import importlib.util
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location(
        "_h5lite_for_shim", os.path.join(here, "..", "victor_amd", "h5lite.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.read_all(fn)


class _FakeDataset:
    def __init__(self, arr):
        self._a = arr

    def __getitem__(self, idx):
        return self._a[idx]

    @property
    def shape(self):
        return self._a.shape


class _FakeH5File:
    def __init__(self, fn, mode="r"):
        self._d = _h5_read(fn)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def keys(self):
        return self._d.keys()

    def __getitem__(self, key):
        return _FakeDataset(self._d[key])

    def __contains__(self, key):
        return key in self._d


class _Quantity:
    def __init__(self, v):
        self.value = v


class _FakeLambdaCDM:
    def __init__(self, H0, Om0, Ode0, **kw):
        self.H0, self.Om0, self.Ode0 = H0, Om0, Ode0
        self.Ok0 = 1.0 - Om0 - Ode0

    def _E(self, z):
        z = np.asarray(z, dtype=float)
        return np.sqrt(self.Om0 * (1 + z) ** 3 + self.Ok0 * (1 + z) ** 2 + self.Ode0)

    def H(self, z):
        return _Quantity(self.H0 * self._E(z))

    def Om(self, z):
        z = np.asarray(z, dtype=float)
        return self.Om0 * (1 + z) ** 3 / self._E(z) ** 2


def _make_interp2d():
    import scipy.interpolate as si

    class interp2d:  # noqa: N801 - mirrors the removed SciPy name
        def __init__(self, x, y, z, kind="linear", **kw):
            k = {"linear": 1, "cubic": 3, "quintic": 5}[kind]
            x = np.asarray(x, dtype=float).ravel()
            y = np.asarray(y, dtype=float).ravel()
            z = np.asarray(z, dtype=float)
            if z.shape != (len(y), len(x)):
                z = z.reshape(len(y), len(x))
            self._spl = si.RectBivariateSpline(x, y, z.T, kx=k, ky=k, s=0)

        def __call__(self, x, y):
            x = np.sort(np.atleast_1d(np.asarray(x, dtype=float)))
            y = np.sort(np.atleast_1d(np.asarray(y, dtype=float)))
            z = self._spl(x, y).T
            if z.shape[0] == 1:
                z = z[0]
            return z

    return interp2d


_victor = None
_simpson_rule = "simpson"


def _make_simps(rule):
    import scipy.integrate
    if rule == "simpson":
        return scipy.integrate.simpson
    if rule != "avg":
        raise ValueError("simps stand-in: rule must be 'simpson' (SciPy >= 1.11) or 'avg' (SciPy < 1.11)")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    try:
        from victor_oracle import simps_legacy
    finally:
        sys.path.pop(0)

    def simps(y, x=None, dx=1.0, axis=-1, even="avg"):
        y = np.asarray(y, dtype=float)
        if x is None:
            x = dx * np.arange(y.shape[axis])
        return simps_legacy(y, x, axis=axis, even=even)

    return simps


def set_simpson_rule(rule):
    """Choose which SciPy's ``simps`` the reference sees: 'simpson' (SciPy >= 1.11, default) or 'avg' (SciPy < 1.11).
    Works before or after :func:`load` (the reference binds the name at import, ccf_model.py:8, so the stand-in is
    swapped in the imported module's namespace as well; the reference source is not touched)."""
    global _simpson_rule
    fn = _make_simps(rule)
    _simpson_rule = rule
    import scipy.integrate
    scipy.integrate.simps = fn
    if _victor is not None:
        for name in ("victor.ccf_model", "victor.ccf"):
            mod = sys.modules.get(name)
            if mod is not None and hasattr(mod, "simps"):
                mod.simps = fn


def simpson_rule():
    return _simpson_rule


def load():
    """Return the reference ``victor`` package (imported once, with stand-ins)."""
    global _victor
    if _victor is not None:
        return _victor
    if not available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    warnings.filterwarnings("ignore")
    import matplotlib
    matplotlib.use("Agg")

    h5 = types.ModuleType("h5py")
    h5.File = _FakeH5File
    sys.modules["h5py"] = h5

    astropy = types.ModuleType("astropy")
    cosmology = types.ModuleType("astropy.cosmology")
    cosmology.LambdaCDM = _FakeLambdaCDM
    astropy.cosmology = cosmology
    sys.modules["astropy"] = astropy
    sys.modules["astropy.cosmology"] = cosmology

    import scipy.integrate
    import scipy.interpolate
    scipy.integrate.simps = _make_simps(_simpson_rule)
    scipy.interpolate.interp2d = _make_interp2d()

    # our own repo also ships a drop-in package called ``victor`` (a thin alias
    # of victor_amd); make sure the *reference* wins for this process
    for name in [m for m in sys.modules if m == "victor" or m.startswith("victor.")]:
        del sys.modules[name]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import victor
    finally:
        sys.path.remove(REFERENCE_ROOT)
    assert os.path.abspath(victor.__file__).startswith(os.path.abspath(REFERENCE_ROOT))
    _victor = victor
    return victor


def load_cobaya_plugin():
    """Import the reference's cobaya plug-in against a stub ``cobaya`` base class."""
    load()
    cobaya = types.ModuleType("cobaya")
    likelihood = types.ModuleType("cobaya.likelihood")

    class Likelihood:
        model = None
        data = None
        config_file = None

        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)
            self.initialize()

    likelihood.Likelihood = Likelihood
    cobaya.likelihood = likelihood
    sys.modules.setdefault("cobaya", cobaya)
    sys.modules.setdefault("cobaya.likelihood", likelihood)
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_ref_CCFLikelihood", os.path.join(REFERENCE_ROOT, "victor", "likelihoods", "CCFLikelihood.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.CCFLikelihood


def boss_config():
    import yaml
    with open(os.path.join(REFERENCE_ROOT, "config", "boss_config.yaml")) as fh:
        info = yaml.full_load(fh)
    info["model"]["dir"] = REFERENCE_ROOT
    info["data"]["dir"] = REFERENCE_ROOT
    return info


if __name__ == "__main__":
    v = load()
    info = boss_config()
    fit = v.CCFFit(info["model"], info["data"])
    p = {"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0}
    lnl, chi2 = fit.log_likelihood(p)
    print("iaH = %.18g" % fit.iaH)
    print("chi2 = %.12f lnL = %.12f" % (chi2, lnl))
