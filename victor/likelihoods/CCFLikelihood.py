"""cobaya plug-in (reference: ``victor/likelihoods/CCFLikelihood.py:6-42``).

Same class name, attributes (``model``, ``data``, ``config_file``), hooks and ``state`` keys as the
reference, so ``config/boss_cobaya_config.yaml`` runs unchanged; the likelihood itself is evaluated by the
HIP kernels through :class:`victor_amd.CCFFit`.  cobaya is imported lazily: without it the class still
works as a plain object (used by the tests and by the batched samplers in :mod:`victor_amd.sampler`).
"""

import os

import yaml

try:  # pragma: no cover - cobaya is optional
    from cobaya.likelihood import Likelihood
except ImportError:  # minimal stand-in with cobaya's attribute-injection behaviour
    class Likelihood:
        model = None
        data = None
        config_file = "config/boss_config.yaml"

        def __init__(self, info=None, **kwargs):
            for key, value in dict(info or {}, **kwargs).items():
                setattr(self, key, value)
            self.initialize()

from victor import CCFFit


class CCFLikelihood(Likelihood):

    def initialize(self):
        """Build the fitter from the ``model``/``data`` blocks, or from ``config_file`` when they are absent."""
        if self.model is None or self.data is None:
            print(f"'model' or 'data' blocks not provided, attempting to read them from config file "
                  f"{self.config_file}")
            if not os.path.isfile(self.config_file):
                raise KeyError(f"config file {self.config_file} not found")
            with open(self.config_file) as fh:
                info = yaml.full_load(fh)
            self.model = info["model"]
            self.data = info["data"]
        from victor_amd.sharding import default_device
        self.ccf = CCFFit(self.model, self.data, device=default_device())   # one chain per GPU under mpirun / torchrun

    def get_can_provide_params(self):
        return ["fsigma8"]

    def calculate(self, state, want_derived=True, **params_values):
        """Set ``state['logp']`` and the derived chi-square for one sample."""
        lnlike, chisq = self.ccf.log_likelihood(params_values)
        state["logp"] = lnlike
        state["derived"] = {"chi2_ccf_correct": chisq}
        if self.model["matter_ccf"]["model"] == "use_excursion_model":
            state["derived"]["fsigma8"] = params_values["f"] * self.ccf.s8z

    def calculate_batch(self, params_values):
        """Extension: evaluate many samples at once; returns (logp[n], chi2[n])."""
        return self.ccf.log_likelihood_batch(params_values)
