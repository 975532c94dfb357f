"""Batched Metropolis walkers: the caller that turns batch throughput into sampling wall-clock.

The reference is sampled by cobaya, which asks for one likelihood per call
(``victor/likelihoods/CCFLikelihood.py:32``); under MPI it runs independent chains, one process each
(``README.md:30``).  Here W walkers advance in lock-step: each step proposes W points, evaluates them as ONE
batch on the GPU and accepts/rejects per walker.  Priors, reference (starting) distributions and proposal
widths are read from the same ``params:`` block cobaya reads (``config/boss_cobaya_config.yaml:50-97`` in the
reference).  With several GPUs every rank drives its own walkers and the step's log-likelihoods are
all-gathered (RCCL) so that each rank sees the whole ensemble (BASELINE config "8 walkers x 8 GPUs").

This is a plain random-walk Metropolis with diagonal Gaussian proposals - deliberately simple and
deterministic for a given seed, so that a chain driven by the HIP path can be compared step for step with
the same chain driven by the CPU oracle (tests/test_gpu_workloads.py).
"""

import numpy as np

from .utils import InputError


class ParamSpec:
    def __init__(self, name, lo, hi, ref_loc, ref_scale, proposal):
        self.name, self.lo, self.hi = name, float(lo), float(hi)
        self.ref_loc, self.ref_scale, self.proposal = float(ref_loc), float(ref_scale), float(proposal)
        if not self.hi > self.lo:
            raise InputError(f"prior of {name}: max must exceed min")


def parse_cobaya_params(params_block):
    """Split a cobaya ``params`` block into sampled :class:`ParamSpec` s and fixed values.

    Sampled parameters have a ``prior`` with ``min``/``max`` (uniform); ``ref`` (norm loc/scale) gives the
    starting distribution and ``proposal`` the step size.  Scalars are fixed values; entries with a ``value``
    lambda or ``derived: True`` are cobaya bookkeeping and are skipped (aperp/apar are derived from alpha and
    epsilon inside the likelihood, ccf_model.py:589-592).
    """
    sampled, fixed = [], {}
    for name, spec in (params_block or {}).items():
        if spec is None:
            continue
        if isinstance(spec, (int, float)):
            fixed[name] = float(spec)
            continue
        if not isinstance(spec, dict) or "prior" not in spec:
            continue
        prior = spec["prior"]
        if prior.get("dist", "uniform") != "uniform":
            raise InputError(f"only uniform priors are supported (parameter {name})")
        ref = spec.get("ref", {})
        if isinstance(ref, (int, float)):
            ref = {"loc": ref, "scale": 0.0}
        lo, hi = prior["min"], prior["max"]
        width = hi - lo
        sampled.append(ParamSpec(name, lo, hi, ref.get("loc", 0.5 * (lo + hi)), ref.get("scale", 0.1 * width),
                                 spec.get("proposal", 0.05 * width)))
    if not sampled:
        raise InputError("no sampled parameters found")
    return sampled, fixed


class EnsembleMetropolis:
    """W independent random-walk Metropolis chains advanced together, one likelihood batch per step.

    ``evaluate(batch) -> lnL`` receives a dict ``name -> array(W')`` (sampled + fixed parameters) for the W'
    proposals that lie inside the prior box and returns their log-likelihoods.
    """

    def __init__(self, evaluate, specs, n_walkers, seed=0, fixed=None):
        self.evaluate = evaluate
        self.specs = list(specs)
        self.fixed = dict(fixed or {})
        self.n_walkers = int(n_walkers)
        self.rng = np.random.default_rng(seed)
        self.lo = np.array([s.lo for s in self.specs])
        self.hi = np.array([s.hi for s in self.specs])
        self.width = np.array([s.proposal for s in self.specs])
        self.x = None
        self.lnl = None
        self.n_accept = 0
        self.n_steps = 0
        self.n_evals = 0

    @property
    def names(self):
        return [s.name for s in self.specs]

    def _batch(self, x):
        batch = {s.name: np.ascontiguousarray(x[:, j]) for j, s in enumerate(self.specs)}
        for k, v in self.fixed.items():
            batch[k] = v
        return batch

    def _lnl(self, x):
        inside = np.all((x >= self.lo) & (x <= self.hi), axis=1)
        out = np.full(len(x), -np.inf)
        if np.any(inside):
            out[inside] = np.asarray(self.evaluate(self._batch(x[inside])), dtype=float)
            self.n_evals += int(inside.sum())
        return out

    def initialise(self):
        """Draw each walker from the reference distribution, redrawing until it lies inside the prior."""
        loc = np.array([s.ref_loc for s in self.specs])
        scale = np.array([s.ref_scale for s in self.specs])
        x = np.empty((self.n_walkers, len(self.specs)))
        for w in range(self.n_walkers):
            for _ in range(1000):
                cand = loc + scale * self.rng.standard_normal(len(self.specs))
                if np.all((cand >= self.lo) & (cand <= self.hi)):
                    break
            else:
                raise InputError("reference distribution lies outside the prior")
            x[w] = cand
        self.x = x
        self.lnl = self._lnl(x)
        return self

    def step(self):
        if self.x is None:
            self.initialise()
        prop = self.x + self.width * self.rng.standard_normal(self.x.shape)
        logu = np.log(self.rng.random(self.n_walkers))
        lnl_prop = self._lnl(prop)
        accept = logu < lnl_prop - self.lnl
        self.x = np.where(accept[:, None], prop, self.x)
        self.lnl = np.where(accept, lnl_prop, self.lnl)
        self.n_accept += int(accept.sum())
        self.n_steps += 1
        return accept

    def run(self, n_steps, on_step=None):
        """Advance ``n_steps``; returns ``(chain[n_steps, W, P], lnl[n_steps, W])``."""
        if self.x is None:
            self.initialise()
        chain = np.empty((n_steps, self.n_walkers, len(self.specs)))
        lnl = np.empty((n_steps, self.n_walkers))
        for t in range(n_steps):
            self.step()
            chain[t] = self.x
            lnl[t] = self.lnl
            if on_step is not None:
                on_step(t, self)
        return chain, lnl

    @property
    def acceptance(self):
        return self.n_accept / max(1, self.n_steps * self.n_walkers)


class EnsembleStretch(EnsembleMetropolis):
    """Affine-invariant ensemble sampler (Goodman & Weare 2010 stretch move, parallel form of Foreman-Mackey et al.
    2013): the ensemble is split in two halves, each half moves along lines through walkers drawn from the other half
    and is evaluated as ONE likelihood batch - two batches of W/2 per step, no proposal widths to tune.  Same
    ``evaluate`` / ``specs`` / prior-box conventions as :class:`EnsembleMetropolis`; W must be even and >= 2 (P + 1).
    """

    def __init__(self, evaluate, specs, n_walkers, seed=0, fixed=None, a=2.0):
        super().__init__(evaluate, specs, n_walkers, seed=seed, fixed=fixed)
        if self.n_walkers % 2 or self.n_walkers < 2 * (len(self.specs) + 1):
            raise InputError("the stretch move needs an even number of walkers, at least 2 (n_params + 1)")
        self.a = float(a)

    def step(self):
        if self.x is None:
            self.initialise()
        half = self.n_walkers // 2
        ndim = len(self.specs)
        accepted = np.zeros(self.n_walkers, dtype=bool)
        for first in (True, False):
            move = slice(0, half) if first else slice(half, None)
            other = self.x[half:] if first else self.x[:half]
            # z ~ g(z) proportional to 1/sqrt(z) on [1/a, a]
            z = ((self.a - 1.0) * self.rng.random(half) + 1.0) ** 2 / self.a
            partner = other[self.rng.integers(0, half, size=half)]
            prop = partner + z[:, None] * (self.x[move] - partner)
            logu = np.log(self.rng.random(half))
            lnl_prop = self._lnl(prop)
            accept = logu < (ndim - 1) * np.log(z) + lnl_prop - self.lnl[move]
            self.x[move] = np.where(accept[:, None], prop, self.x[move])
            self.lnl[move] = np.where(accept, lnl_prop, self.lnl[move])
            accepted[move] = accept
        self.n_accept += int(accepted.sum())
        self.n_steps += 1
        return accepted


def gelman_rubin(chain):
    """R-1 per parameter from ``chain[steps, walkers, params]`` (between- over within-walker variance)."""
    n = chain.shape[0]
    means = chain.mean(axis=0)
    within = chain.var(axis=0, ddof=1).mean(axis=0)
    between = n * means.var(axis=0, ddof=1)
    var_hat = (n - 1) / n * within + between / n
    return np.sqrt(var_hat / within) - 1.0


class DistributedEnsemble:
    """One :class:`EnsembleMetropolis` per rank plus an all-gather of every step's log-likelihoods.

    ``gather(local_lnl) -> all_lnl`` is ``Dist.allgather_host`` (the ranks' socket group) or an RCCL gather through the engine;
    the walkers themselves never interact, so the collective is monitoring traffic only (W doubles per rank).
    """

    def __init__(self, evaluate, specs, walkers_per_rank, dist, seed=0, fixed=None, gather=None,
                 sampler=None):
        self.dist = dist
        self.local = (sampler or EnsembleMetropolis)(evaluate, specs, walkers_per_rank, seed=seed + 7919 * dist.rank,
                                                     fixed=fixed)
        self.gather = gather or (lambda v: dist.allgather_host(v, len(v)))
        self.all_lnl = []

    def run(self, n_steps):
        def on_step(t, ens):
            self.all_lnl.append(self.gather(np.ascontiguousarray(ens.lnl)))
        chain, lnl = self.local.run(n_steps, on_step=on_step)
        return chain, lnl, np.array(self.all_lnl)
