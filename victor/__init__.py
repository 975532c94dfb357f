"""Drop-in alias: ``import victor`` resolves to the MI355X implementation in :mod:`victor_amd`.

Existing scripts, notebooks and cobaya YAML files written for seshnadathur/victor
(``from victor import CCFFit``; ``python_path: ./victor/likelihoods/``) keep working unchanged.
"""

from victor_amd import BackgroundCosmology, CCFFit, CCFModel, InputError, __version__, utils  # noqa: F401
