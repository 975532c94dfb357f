"""Background expansion used by the likelihood path: E(z) and what follows from it directly.

The reference's ``BackgroundCosmology`` (``victor/cosmology.py:6-45``) wraps astropy's ``LambdaCDM`` with its default
``Tcmb0 = 0`` (no radiation), i.e. ``E(z)^2 = Om (1+z)^3 + Ok (1+z)^2 + 1 - Om - Ok``.  Only ``Ez`` is on the hot path
(``ccf_model.py:43-45``); ``H`` and ``Om`` come for free.  The distance and growth-rate approximations of the reference
class are not part of the path and are not provided.
"""

import numpy as np


class BackgroundCosmology:
    """Same constructor and attributes as the reference class (``cosmology.py:16-33``)."""

    def __init__(self, cosmology=None):
        cosmology = cosmology or {}
        self.c = 299792.458                                     # km/s
        self.OmegaM = cosmology.get("Omega_m", 0.31)
        self.OmegaK = cosmology.get("Omega_K", 0)
        self.OmegaL = 1 - self.OmegaM - self.OmegaK
        self.H0 = cosmology.get("H0", 100 * cosmology.get("h", 0.675))
        self.rd = cosmology.get("sound_horizon", 148.1)
        self.sigma8 = cosmology.get("sigma8", 0.81)

    def Ez(self, z):
        """H(z)/H0 (``cosmology.py:41-45``)."""
        z = np.asarray(z, dtype=float)
        e = np.sqrt(self.OmegaM * (1 + z) ** 3 + self.OmegaK * (1 + z) ** 2 + self.OmegaL)
        return float(e) if e.ndim == 0 else e

    def H(self, z):
        """Hubble parameter in km/s/Mpc (``cosmology.py:35-39``)."""
        return self.H0 * self.Ez(z)

    def Om(self, z):
        """Matter density parameter at redshift z (``cosmology.py:47-51``)."""
        z = np.asarray(z, dtype=float)
        return self.OmegaM * (1 + z) ** 3 / self.Ez(z) ** 2
