"""Batched Metropolis sampler on CPU: config parsing, determinism, prior handling, a known Gaussian target."""

import numpy as np
import pytest

from tests import cases
from victor_amd.sampler import EnsembleMetropolis, ParamSpec, gelman_rubin, parse_cobaya_params


def test_parse_reference_style_params_block():
    info = cases.cobaya_info()
    specs, fixed = parse_cobaya_params(info["params"])
    assert [s.name for s in specs] == ["fsigma8", "beta", "sigma_v", "epsilon"]
    f = specs[0]
    assert (f.lo, f.hi, f.ref_loc, f.ref_scale, f.proposal) == (0.05, 1.5, 0.47, 0.05, 0.02)   # boss_cobaya_config.yaml:51-61
    s = specs[2]
    assert (s.lo, s.hi, s.ref_loc, s.ref_scale, s.proposal) == (100, 500, 380, 20, 10)
    assert fixed == {}
    specs, fixed = parse_cobaya_params({"a": {"prior": {"min": 0, "max": 1}}, "b": 2.5, "c": {"derived": True},
                                        "d": {"value": "lambda a: a"}, "e": None})
    assert [s.name for s in specs] == ["a"] and fixed == {"b": 2.5}


def gaussian_target(mu, sig):
    def evaluate(batch):
        x = np.stack([batch["a"], batch["b"]], axis=1)
        return -0.5 * np.sum(((x - mu) / sig) ** 2, axis=1)
    return evaluate


def test_recovers_gaussian_and_respects_prior():
    mu, sig = np.array([0.3, -1.0]), np.array([0.1, 0.5])
    specs = [ParamSpec("a", -1, 1, 0.2, 0.05, 0.15), ParamSpec("b", -4, 0.0, -1.2, 0.2, 0.7)]
    ens = EnsembleMetropolis(gaussian_target(mu, sig), specs, n_walkers=64, seed=3).initialise()
    ens.run(300)                                      # burn-in
    chain, lnl = ens.run(1500)
    flat = chain.reshape(-1, 2)
    assert np.all(flat[:, 1] <= 0.0) and np.all(flat[:, 0] >= -1)
    assert abs(flat[:, 0].mean() - 0.3) < 0.01 and abs(flat[:, 0].std() - 0.1) < 0.01
    # b is truncated at 0 (2 sigma above the mean): compare with the truncated normal moments
    from scipy.stats import truncnorm
    tn = truncnorm((-4 + 1.0) / 0.5, (0.0 + 1.0) / 0.5, loc=-1.0, scale=0.5)
    assert abs(flat[:, 1].mean() - tn.mean()) < 0.03 and abs(flat[:, 1].std() - tn.std()) < 0.03
    assert 0.15 < ens.acceptance < 0.8
    assert np.all(gelman_rubin(chain) < 0.05)
    assert np.allclose(lnl[-1], gaussian_target(mu, sig)(ens._batch(chain[-1])))


def test_deterministic_for_fixed_seed_and_batched_calls():
    calls = []

    def evaluate(batch):
        calls.append(len(batch["a"]))
        return -0.5 * (batch["a"] ** 2 + (batch["b"] - batch["c"]) ** 2)

    specs = [ParamSpec("a", -5, 5, 0, 1, 0.5), ParamSpec("b", -5, 5, 0, 1, 0.5)]
    runs = []
    for _ in range(2):
        ens = EnsembleMetropolis(evaluate, specs, n_walkers=8, seed=11, fixed={"c": 0.25}).initialise()
        runs.append(ens.run(20))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    assert max(calls) <= 8 and len(calls) == 2 * 21          # one batched likelihood call per step (+ initialisation)
    other = EnsembleMetropolis(evaluate, specs, n_walkers=8, seed=12, fixed={"c": 0.25}).initialise().run(20)
    assert not np.array_equal(other[0], runs[0][0])


def test_out_of_prior_proposals_are_never_evaluated():
    seen = []

    def evaluate(batch):
        seen.append(batch["a"].copy())
        return np.zeros(len(batch["a"]))

    ens = EnsembleMetropolis(evaluate, [ParamSpec("a", 0, 1, 0.5, 0.1, 5.0)], n_walkers=32, seed=0).initialise()
    ens.run(10)
    allv = np.concatenate(seen)
    assert np.all((allv >= 0) & (allv <= 1))
    assert ens.n_evals == len(allv) < 32 * 11


def test_stretch_move_recovers_a_correlated_gaussian():
    """EnsembleStretch on a strongly correlated 3-d Gaussian inside a wide prior box: mean and covariance come back
    without any proposal tuning (the affine-invariant move does not care about the correlation)."""
    from victor_amd.sampler import EnsembleStretch, ParamSpec
    from victor_amd.utils import InputError
    mean = np.array([0.5, -1.0, 2.0])
    A = np.array([[1.0, 0.0, 0.0], [0.95, 0.3, 0.0], [-0.5, 0.2, 0.1]])
    cov = A @ A.T
    icov = np.linalg.inv(cov)
    calls = []

    def evaluate(batch):
        x = np.stack([batch["a"], batch["b"], batch["c"]], axis=1) - mean
        calls.append(len(x))
        return -0.5 * np.einsum("ni,ij,nj->n", x, icov, x)

    specs = [ParamSpec(n, -20, 20, m, 0.5, 0.1) for n, m in zip("abc", mean)]
    ens = EnsembleStretch(evaluate, specs, 64, seed=3)
    chain, lnl = ens.run(1500)
    assert chain.shape == (1500, 64, 3) and 0.2 < ens.acceptance < 0.8
    assert set(calls[1:]) == {32}                                   # two half-ensemble batches per step
    flat = chain[300:].reshape(-1, 3)
    assert np.max(np.abs(flat.mean(axis=0) - mean)) < 0.08
    assert np.max(np.abs(np.cov(flat.T) - cov)) < 0.12 * np.max(cov)
    with pytest.raises(InputError):
        EnsembleStretch(evaluate, specs, 7)
    with pytest.raises(InputError):
        EnsembleStretch(evaluate, specs, 6)


class _RowsEngine:
    """Stand-in for victor_amd.engine.Engine on the CPU: the two halves of a small host-buffer batch (vk_eval_batch_begin /
    _finish), evaluating a simple function of the parameter ROWS - so the direct route of the sampler (rows written in place,
    half-ensembles on two contexts, pipelined run) is exercised end to end without a GPU."""
    device, simpson_even = 0, "simpson"

    def __init__(self, log):
        import ctypes as C
        self.C, self.log, self._ctx, self._pending = C, log, None, None
        self._lib = self

    @staticmethod
    def lnl_of_rows(rows):          # fsigma8, sigma_v, aperp, apar, epsilon, beta: every column the sampler writes matters
        return -0.5 * (((rows[:, 0] - 0.47) / 0.05) ** 2 + ((rows[:, 1] - 380.0) / 30.0) ** 2 + ((rows[:, 5] - 0.4) / 0.08) ** 2
                       + ((rows[:, 2] - 1.0) / 0.05) ** 2 + ((rows[:, 3] - 1.0) / 0.07) ** 2 + (rows[:, 4] - 1.0) ** 2)

    def vk_eval_batch_begin(self, ctx, opts, p_rows, n):
        assert self._pending is None, "one batch per context at a time"
        rows = np.ctypeslib.as_array(p_rows, shape=(n, 12)).copy()
        self._pending = self.lnl_of_rows(rows)
        self.log.append(n)
        return 0

    def vk_eval_batch_finish(self, ctx, p_lnl, p_chi):
        out = np.ctypeslib.as_array(p_lnl, shape=(len(self._pending),))
        out[:] = self._pending
        self._pending = None
        return 0

    def _check(self, rc):
        assert rc == 0


class _RowsFit:
    """The parts of CCFFit the sampler's direct route uses, with CCFModel's own row building."""

    def __init__(self, log):
        from victor_amd.ccf_model import CCFModel
        self.model = {"bias": 1.9}
        self._engine = _RowsEngine(log)
        self._rows = CCFModel._param_rows.__get__(self)
        self._scalar_row = CCFModel._scalar_row.__get__(self)
        self.log = log

    def _single_point_plan(self):
        return (self._engine, None, True, True, None)

    def _fit_rows(self, params, model):
        return self._rows(params, True, True)

    def _merged(self, kw):
        return dict(self.model)

    def _engine_key(self, model):
        return "template"

    def log_likelihood_batch(self, params):
        rows = self._rows(params, True, True)
        return _RowsEngine.lnl_of_rows(rows), None


@pytest.mark.parametrize("walkers", [8, 7, 2, 1])
def test_direct_route_is_the_dictionary_route_chain(monkeypatch, walkers):
    """Rows written in place, two half-ensembles, run() pipelined over the steps: same random numbers, same rows - the chain of
    the dictionary route, value for value (the stand-in engine is a function of the row alone), through proposals outside the
    prior, across a block boundary of the pre-drawn random numbers, for odd ensembles and a single walker; step() and run() agree."""
    import victor_amd.engine as E
    info = cases.cobaya_info()
    specs, fixed = parse_cobaya_params(info["params"])
    # narrow the prior of sigma_v so that proposals leave the box often
    specs = [ParamSpec(s.name, s.lo, s.hi, s.ref_loc, s.ref_scale, s.proposal) if s.name != "sigma_v"
             else ParamSpec("sigma_v", 360.0, 400.0, 380.0, 5.0, 25.0) for s in specs]
    log = []
    monkeypatch.setattr(E, "Engine", lambda *a, **k: _RowsEngine(log))
    fit = _RowsFit(log)

    def evaluate(batch):
        return fit.log_likelihood_batch(batch)[0]

    d = EnsembleMetropolis(None, specs, walkers, seed=9, fixed=fixed, fit=fit, native=False).initialise()      # (the stand-in engine has no context for vk_walk_run: tests/test_gpu_workloads.py covers that loop)
    g = EnsembleMetropolis(evaluate, specs, walkers, seed=9, fixed=fixed).initialise()
    assert d._direct is not None and len(d._direct["engines"]) == (2 if walkers >= 2 else 1)
    cd, ld = d.run(150)
    cg, lg = g.run(150)
    assert np.array_equal(cd, cg) and np.array_equal(ld, lg)
    assert d.n_accept == g.n_accept and d.n_evals == g.n_evals and d.n_steps == g.n_steps == 150
    assert 0 < d.n_accept < 150 * walkers and d.n_evals < 151 * walkers            # some proposals were outside the prior
    if walkers >= 2:
        assert set(log) <= {walkers // 2, walkers - walkers // 2}                   # only half-ensemble batches were launched
    s1 = EnsembleMetropolis(None, specs, walkers, seed=9, fixed=fixed, fit=fit, native=False).initialise()
    for t in range(70):
        s1.step()
        assert np.array_equal(s1.x, cd[t]) and np.array_equal(s1.lnl, ld[t]), t
    # an unknown sampled parameter keeps the dictionary route
    odd = [ParamSpec("alpha", 0.9, 1.1, 1.0, 0.01, 0.01)] + specs
    e = EnsembleMetropolis(evaluate, odd, walkers, seed=1, fixed=fixed, fit=fit, native=False).initialise()
    assert e._direct is None
