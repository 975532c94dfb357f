"""victor_amd - MI355X-native likelihood engine with victor's CCFModel / CCFFit interface.

The public names match the reference package (``victor/__init__.py:3-9``) for everything on the
likelihood path; evaluation runs in hand-written HIP kernels behind ``libvictor_hip.so``.
"""

from . import utils
from .ccf_fit import CCFFit
from .ccf_model import CCFModel
from .cosmology import BackgroundCosmology
from .utils import InputError

__version__ = "0.1.0"
__all__ = ["CCFModel", "CCFFit", "BackgroundCosmology", "InputError", "utils", "__version__"]
