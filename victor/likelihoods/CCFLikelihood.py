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
except ImportError:
    class Likelihood:
        """Stand-in for ``cobaya.likelihood.Likelihood`` when cobaya is not installed, shaped like the real one: the same
        constructor signature, class defaults read from ``<ClassName>.yaml`` next to the class's module and overridden by
        ``info`` (cobaya's defaults mechanism), every top-level option set as an attribute, the ``params`` block kept as
        ``params`` / ``input_params`` / ``output_params``, then ``initialize()``."""

        def __init__(self, info=None, name=None, timing=None, packages_path=None, initialize=True, standalone=True,
                     **kwargs):
            import inspect
            merged = {}
            path = os.path.splitext(inspect.getfile(type(self)))[0] + ".yaml"
            if os.path.isfile(path):
                with open(path) as fh:
                    merged.update(yaml.full_load(fh) or {})
            merged.update(info or {})
            merged.update(kwargs)
            self._name = name or type(self).__name__
            self.packages_path = packages_path
            self.params = dict(merged.pop("params", None) or {})
            derived = {k for k, v in self.params.items() if isinstance(v, dict) and v.get("derived")}
            self.input_params = [k for k in self.params if k not in derived]
            self.output_params = sorted(derived)
            for key, value in merged.items():
                setattr(self, key, value)
            if initialize:
                self.initialize()

        def get_name(self):
            return self._name

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
        # One chain per GPU under mpirun / torchrun - or, with VICTOR_HIP_BROKER set in the job's environment, every chain's
        # calculate() goes to the mailbox of one GPU owner process that batches them (victor_amd/broker.py): this file, the
        # YAML and the mpirun line stay as they are.
        if os.environ.get("VICTOR_HIP_BROKER"):
            from victor_amd.broker import broker_device
            device = broker_device()            # without asking the HIP runtime: a brokered chain never initialises the GPU
        else:
            from victor_amd.sharding import default_device
            device = default_device()
        self.ccf = CCFFit(self.model, self.data, device=device)

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
