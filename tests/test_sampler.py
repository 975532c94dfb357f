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
