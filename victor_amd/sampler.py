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

import ctypes as C

import numpy as np

from . import _native as N
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

    ``evaluate(batch) -> lnL`` receives a dict ``name -> array(W')`` (sampled + fixed parameters) for the W' proposals that lie
    inside the prior box and returns their log-likelihoods.

    ``fit`` (a :class:`victor_amd.CCFFit`): the step's likelihood batch goes straight to the engine instead - a preallocated
    ``(W, VK_NPAR)`` row array whose sampled columns are overwritten in place, preallocated result buffers, no dictionary and
    no option merging per step; ``evaluate`` is then unused.  A step of 8 walkers is one ~20 us launch: every NumPy call on the
    way counts (BENCH ``walker_ensembles``).  Both routes draw their random numbers in the same blocks and form the rows with
    the same NumPy expressions, so a chain is the same chain whichever route evaluates it (tests/test_gpu_workloads.py).

    ``native`` (default, with ``fit``): :meth:`run` hands each block of pre-drawn random numbers to ``vk_walk_run``
    (include/victor_hip.h) - the same two half-ensembles, the same pipelining, the same rows (one routine forms the
    Alcock-Paczynski factors on every route: ``vk_epsilon_to_ap``) as the Python loop below, which remains the definition
    (``native=False``) and the route of :meth:`step`; what goes are the ~20 NumPy calls per step that made the host the limit of
    a small ensemble (8 walkers: 26-28 -> 17-18 us per step).

    ``speculate`` (default up to ``SPECULATE_MAX_WALKERS`` walkers): the library loop takes TWO steps per launch.  **Parity with
    the Python loop is then to rounding, not bit for bit**: a launch of three rows per walker takes another work split than one
    of one row, so the log-likelihoods agree to ~1e-13 relative (the tests allow 1e-9) and an acceptance
    ``logu < lnL' - lnL`` decided within that margin could fall the other way - positions and decisions were identical in
    every comparison made (tests/test_gpu_workloads.py: thousands of steps against the Python loop, the dictionary route and an
    oracle-driven chain), the stored log-likelihoods differ in their last bits.  Pass ``speculate=False`` where the chain of
    ``native=False`` is wanted bit for bit (one step per launch: the same launches).  For a given setting the chain does not
    depend on how :meth:`run` is cut into pieces: a left-over single step travels in a launch of the same shape.
    """

    BLOCK = 64          # steps whose proposal increments and acceptance levels are drawn together

    SPECULATE_MAX_WALKERS = 96       # two steps per launch (vk_walk_create: speculate) up to this ensemble size (measured: 1.94 x the
                                     # Python loop at 8 walkers, 1.7 x at 64, level with one step per launch at 128, behind it beyond)

    def __init__(self, evaluate, specs, n_walkers, seed=0, fixed=None, fit=None, native=True, speculate=None, native_contexts=None):
        self.native = bool(native)       # with `fit`: run() hands whole blocks of steps to the library (vk_walk_run)
        # the library loop takes two steps per launch (three evaluations per walker for two steps: the same chain, half the
        # round trips) while the GPU has room for it; None = by ensemble size
        self.speculate = (int(n_walkers) <= self.SPECULATE_MAX_WALKERS) if speculate is None else bool(speculate)
        self.native_contexts = native_contexts       # contexts the library loop spreads the ensemble over (1 or 2; None = choose)
        self._walk = None
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
        self._fit = fit
        self._direct = None
        self._dz = None
        self._logu = None
        self._at = self.BLOCK

    @property
    def names(self):
        return [s.name for s in self.specs]

    def _batch(self, x):
        batch = {s.name: np.ascontiguousarray(x[:, j]) for j, s in enumerate(self.specs)}
        for k, v in self.fixed.items():
            batch[k] = v
        return batch

    # ---- the direct route: rows written in place, two half-ensembles on two contexts, results into preallocated buffers ----
    _COLUMNS = {"fsigma8": 0, "sigma_v": 1, "beta": 5, "astar": 6, "M": 7, "Q": 8, "bias": 9, "Av": 10}

    def _bind(self, x):
        """Set up the direct route for the fit given at construction; False when the sampled parameters are not ones whose
        row columns are known here (the dictionary route then serves)."""
        fit = self._fit
        plan = fit._single_point_plan() if fit is not None else None
        if plan is None or plan[0] is None:
            return False
        names = self.names
        pairs, eps = [], None
        for j, name in enumerate(names):
            if name in self._COLUMNS:
                pairs.append((j, self._COLUMNS[name]))
            elif name == "epsilon" and not ({"alpha", "aperp", "apar"} & set(names)):
                eps = j
            else:
                return False
        if eps is None and "epsilon" in self.fixed and {"aperp", "apar"} & set(names):
            return False
        from .engine import Engine
        rows = np.array(fit._fit_rows(self._batch(x), fit.model), dtype=np.float64, order="C")     # fixed values and defaults
        W = self.n_walkers
        if rows.shape != (W, N.VK_NPAR):
            return False
        # The walkers are independent, so the ensemble advances as two halves on two contexts (streams): while one half is on
        # the GPU the host accepts / rejects the other half and forms its next proposals - a step costs one launch latency
        # instead of a launch latency plus the host's work (run()); step() overlaps the two launches only.
        h = W // 2 if W >= 2 else W
        bounds = [(0, h)] + ([(h, W)] if h < W else [])
        first = plan[0]
        engines = [first]
        if len(bounds) == 2:
            key = fit._engine_key(fit._merged({}))
            engines.append(Engine(fit, fit, device=first.device, matter_model=key, simpson_even=first.simpson_even))
        out = np.empty((2, W))
        alpha = self.fixed.get("alpha", 1)
        stride = N.VK_NPAR * 8
        self._direct = {
            "rows": rows, "pairs": pairs, "eps": eps, "alpha": None if alpha == 1 else alpha, "out": out, "bounds": bounds,
            "engines": engines, "opts": plan[1], "mask": [None] * len(bounds),
            "p_rows": [C.cast(rows.ctypes.data + lo * stride, N._dp) for lo, _ in bounds],
            "p_lnl": [C.cast(out[0].ctypes.data + lo * 8, N._dp) for lo, _ in bounds],
            "p_chi": [C.cast(out[1].ctypes.data + lo * 8, N._dp) for lo, _ in bounds],
            "t1": np.empty_like(x), "t2": np.empty_like(x), "e": [np.empty(hi - lo) for lo, hi in bounds],
            "a": [np.empty(hi - lo) for lo, hi in bounds], "t_aperp": [np.empty(hi - lo) for lo, hi in bounds]}
        # views of the halves, made once: a slice is a new array object every time it is written down
        d = self._direct
        d["t1h"] = [d["t1"][lo:hi] for lo, hi in bounds]
        d["t2h"] = [d["t2"][lo:hi] for lo, hi in bounds]
        d["colv"] = [[(j, rows[lo:hi, c]) for j, c in pairs] for lo, hi in bounds]
        d["derived"] = [(rows[lo:hi, 2], rows[lo:hi, 3], rows[lo:hi, 4]) for lo, hi in bounds]
        d["lnlh"] = [out[0][lo:hi] for lo, hi in bounds]
        d["small"] = [(hi - lo) * x.shape[1] <= 64 for lo, hi in bounds]
        d["box"] = [list(zip(np.tile(self.lo, hi - lo).tolist(), np.tile(self.hi, hi - lo).tolist())) for lo, hi in bounds]
        if self.native and type(self).step is EnsembleMetropolis.step and max(hi - lo for lo, hi in bounds) <= 4096:
            cols = np.full(len(names), N.VK_WALK_EPSILON, dtype=np.int32)
            for j, c in pairs:
                cols[j] = c
            n_ctx = min(len(engines), int(self.native_contexts or len(engines)))
            ctxs = (C.c_void_p * n_ctx)(*[e._ctx for e in engines[:n_ctx]])
            err = C.create_string_buffer(512)
            lib = first._lib
            self._walk = lib.vk_walk_create(ctxs, n_ctx, C.byref(plan[4]), W, len(names),
                                            cols.ctypes.data_as(C.POINTER(C.c_int32)), N.as_dp(N.f64(self.lo)), N.as_dp(N.f64(self.hi)),
                                            N.as_dp(rows), float(alpha), 1 if self.speculate else 0, err, len(err))
            if not self._walk:
                raise InputError("vk_walk_create failed: " + err.value.decode())
            self._walk_lib = lib
        return True

    def __del__(self):
        walk, lib = getattr(self, "_walk", None), getattr(self, "_walk_lib", None)
        if walk and lib is not None:
            try:
                lib.vk_walk_destroy(walk)
            except Exception:
                pass
            self._walk = None

    def _run_native(self, n_steps, on_step):
        """``run`` through ``vk_walk_run``: the blocks of random numbers are drawn exactly as ``_next_randoms`` draws them."""
        W, P = self.x.shape
        chain = np.empty((n_steps, W, P))
        lnl = np.empty((n_steps, W))
        lib = self._walk_lib
        self.x = np.ascontiguousarray(self.x, dtype=np.float64)
        self.lnl = np.ascontiguousarray(self.lnl, dtype=np.float64)
        acc, ev = C.c_int64(0), C.c_int64(0)
        done = 0
        while done < n_steps:
            if self._at >= self.BLOCK:
                self._dz = self.width * self.rng.standard_normal((self.BLOCK,) + self.x.shape)
                self._logu = np.log(self.rng.random((self.BLOCK, self.n_walkers)))
                self._at = 0
            k = min(self.BLOCK - self._at, n_steps - done)
            dz = np.ascontiguousarray(self._dz[self._at:self._at + k])
            logu = np.ascontiguousarray(self._logu[self._at:self._at + k])
            rc = lib.vk_walk_run(self._walk, k, N.as_dp(self.x), N.as_dp(self.lnl), N.as_dp(dz), N.as_dp(logu),
                                 N.as_dp(chain[done:done + k]), N.as_dp(lnl[done:done + k]), C.byref(acc), C.byref(ev))
            if rc != 0:
                msg = (lib.vk_walk_last_error(self._walk) or b"").decode() or f"vk_walk_run failed ({rc})"
                raise (InputError if rc == -1 else N.NativeError)(msg)
            self._at += k
            self.n_steps += k
            if on_step is not None:                      # the state after every step of the block, one step at a time
                x_end, l_end = self.x.copy(), self.lnl.copy()
                for t in range(done, done + k):
                    self.x[:], self.lnl[:] = chain[t], lnl[t]
                    on_step(t, self)
                self.x[:], self.lnl[:] = x_end, l_end
            done += k
        self.n_accept += acc.value
        self.n_evals += ev.value
        return chain, lnl

    def _half_begin(self, k, prop):
        """Enqueue the likelihood of rows lo:hi of the proposals ``prop`` (a (W, P) array, or the half itself)."""
        d = self._direct
        lo, hi = d["bounds"][k]
        xs = prop[lo:hi] if len(prop) == self.n_walkers else prop
        t1, t2 = d["t1h"][k], d["t2h"][k]
        if d["small"][k]:                             # a handful of numbers: a Python loop over a list beats four NumPy calls
            outside = False
            for v, (a, b) in zip(xs.ravel().tolist(), d["box"][k]):
                if not (a <= v <= b):
                    outside = True
                    break
        else:                                         # every walker inside the box <=> min(x - lo, hi - x) >= 0
            np.subtract(xs, self.lo, out=t1)
            np.subtract(self.hi, xs, out=t2)
            outside = min(t1.min(), t2.min()) < 0.0
        mask = None
        if outside:                                   # a proposal outside the prior: its row keeps the walker's position (a valid
            np.subtract(xs, self.lo, out=t1)          # point; the result is discarded) and its log-likelihood reads -inf
            np.subtract(self.hi, xs, out=t2)
            np.minimum(t1, t2, out=t1)
            mask = t1.min(axis=1) >= 0.0
            xs = np.where(mask[:, None], xs, self.x[lo:hi])
        d["mask"][k] = mask
        for j, col in d["colv"][k]:
            col[:] = xs[:, j]
        if d["eps"] is not None:                      # the expressions of CCFModel._param_rows, on the same (contiguous) arrays
            eps, apar = d["e"][k], d["a"][k]
            c_aperp, c_apar, c_eps = d["derived"][k]
            np.copyto(eps, xs[:, d["eps"]])           # as _batch() hands it over, through the same routine (vk_epsilon_to_ap)
            N.epsilon_to_ap(eps, 1.0 if d["alpha"] is None else d["alpha"], aperp=d["t_aperp"][k], apar=apar)
            c_aperp[:] = d["t_aperp"][k]
            c_apar[:] = apar
            c_eps[:] = eps
        eng = d["engines"][k]
        rc = eng._lib.vk_eval_batch_begin(eng._ctx, d["opts"], d["p_rows"][k], hi - lo)
        if rc != 0:
            eng._check(rc)

    def _half_finish(self, k):
        """log-likelihoods of the half begun last (a view of a reused buffer)."""
        d = self._direct
        lo, hi = d["bounds"][k]
        eng = d["engines"][k]
        rc = eng._lib.vk_eval_batch_finish(eng._ctx, d["p_lnl"][k], d["p_chi"][k])
        if rc != 0:
            eng._check(rc)
        lnl = d["lnlh"][k]
        mask = d["mask"][k]
        if mask is None:
            self.n_evals += hi - lo
        else:
            lnl[~mask] = -np.inf
            self.n_evals += int(np.count_nonzero(mask))
        return lnl

    def _lnl(self, x):
        """log-likelihoods of the rows of ``x`` (-inf outside the prior box).  The result may be a view of a reused buffer: callers
        that keep it copy it."""
        d = self._direct
        if d is not None and len(x) == self.n_walkers:
            for k in range(len(d["bounds"])):
                self._half_begin(k, x)
            for k in range(len(d["bounds"])):
                self._half_finish(k)
            return d["out"][0]
        inside = ((x >= self.lo) & (x <= self.hi)).all(axis=1)
        n_in = int(np.count_nonzero(inside))
        out = np.full(len(x), -np.inf)
        if n_in:
            if d is not None:
                out[inside] = self._fit.log_likelihood_batch(self._batch(x[inside]))[0]
            else:
                out[inside] = np.asarray(self.evaluate(self._batch(x[inside])), dtype=float)
            self.n_evals += n_in
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
        if self._fit is not None and self._direct is None:
            self._bind(x)
        self.lnl = np.array(self._lnl(x), dtype=float)
        return self

    def _next_randoms(self):
        """(proposal increments (W, P), log acceptance levels (W,)) of the next step, drawn BLOCK steps at a time."""
        if self._at >= self.BLOCK:
            self._dz = self.width * self.rng.standard_normal((self.BLOCK,) + self.x.shape)
            self._logu = np.log(self.rng.random((self.BLOCK, self.n_walkers)))
            self._at = 0
        t = self._at
        self._at = t + 1
        return self._dz[t], self._logu[t]

    def _accept(self, lo, hi, prop, lnl_prop, logu, x_h=None, lnl_h=None):
        if x_h is None:
            x_h, lnl_h = self.x[lo:hi], self.lnl[lo:hi]
        accept = logu < lnl_prop - lnl_h
        np.copyto(x_h, prop, where=accept[:, None])
        np.copyto(lnl_h, lnl_prop, where=accept)
        self.n_accept += int(accept.sum()) if len(accept) > 64 else accept.tolist().count(True)
        return accept

    def step(self):
        if self.x is None:
            self.initialise()
        dz, logu = self._next_randoms()
        prop = self.x + dz
        lnl_prop = self._lnl(prop)
        accept = self._accept(0, self.n_walkers, prop, lnl_prop, logu)
        self.n_steps += 1
        return accept

    def run(self, n_steps, on_step=None):
        """Advance ``n_steps``; returns ``(chain[n_steps, W, P], lnl[n_steps, W])``."""
        if self.x is None:
            self.initialise()
        if self._walk and n_steps >= 1:
            return self._run_native(n_steps, on_step)
        chain = np.empty((n_steps, self.n_walkers, len(self.specs)))
        lnl = np.empty((n_steps, self.n_walkers))
        d = self._direct
        if d is None or len(d["bounds"]) != 2 or type(self).step is not EnsembleMetropolis.step or n_steps < 1:
            for t in range(n_steps):
                self.step()
                chain[t] = self.x
                lnl[t] = self.lnl
                if on_step is not None:
                    on_step(t, self)
            return chain, lnl
        # Two halves, pipelined over the steps: the same half-batches as step() evaluates - the same chain - but half A of
        # step t + 1 is on the GPU while the host finishes half B of step t.
        (a0, a1), (b0, b1) = d["bounds"]
        xa, xb, la, lb = self.x[a0:a1], self.x[b0:b1], self.lnl[a0:a1], self.lnl[b0:b1]       # views, made once
        dz, logu = self._next_randoms()
        prop_a = xa + dz[a0:a1]
        self._half_begin(0, prop_a)
        for t in range(n_steps):
            prop_b = xb + dz[b0:b1]
            self._half_begin(1, prop_b)
            self._accept(a0, a1, prop_a, self._half_finish(0), logu[a0:a1], xa, la)
            if t + 1 < n_steps:
                dz_next, logu_next = self._next_randoms()
                prop_a = xa + dz_next[a0:a1]
                self._half_begin(0, prop_a)
            self._accept(b0, b1, prop_b, self._half_finish(1), logu[b0:b1], xb, lb)
            self.n_steps += 1
            chain[t] = self.x
            lnl[t] = self.lnl
            if on_step is not None:
                on_step(t, self)
            if t + 1 < n_steps:
                dz, logu = dz_next, logu_next
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

    def __init__(self, evaluate, specs, n_walkers, seed=0, fixed=None, a=2.0, fit=None):
        super().__init__(evaluate, specs, n_walkers, seed=seed, fixed=fixed)        # half-ensemble batches: the dictionary route
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
    """One :class:`EnsembleMetropolis` per rank plus an all-gather of the steps' log-likelihoods.

    ``gather(local) -> all`` takes this rank's doubles and returns every rank's, rank-major (``Dist.allgather_host`` over the
    ranks' socket group, or :class:`victor_amd.sharding.RcclGather` on the engine's stream).  The walkers of different ranks
    never interact - the reference's scale-out is N independent chains under ``mpirun`` (README.md:30) - so the collective is
    monitoring traffic only (W doubles per rank and step) and must not sit in the step: the local log-likelihoods of
    ``gather_block`` steps (default: the 64-step block the random numbers are drawn in) are kept and exchanged as ONE
    ``[K, W]`` array, so a step of 8 walkers (26 us) pays a 64th of a collective instead of a blocking one (round 4: upload,
    all-gather, stream synchronisation and download after EVERY step).  ``all_lnl`` is the same ``[steps, world * W]`` array
    whatever the block length (tests/test_sharding.py, tests/test_gpu_rccl_double.py); ``n_collectives`` counts the gathers.
    """

    def __init__(self, evaluate, specs, walkers_per_rank, dist, seed=0, fixed=None, gather=None,
                 sampler=None, fit=None, gather_block=None, overlap=True):
        self.dist = dist
        self.local = (sampler or EnsembleMetropolis)(evaluate, specs, walkers_per_rank, seed=seed + 7919 * dist.rank,
                                                     fixed=fixed, fit=fit)
        self.gather = gather or (lambda v: dist.allgather_host(v, len(v)))
        self.gather_block = int(gather_block if gather_block is not None else EnsembleMetropolis.BLOCK)
        if self.gather_block < 1:
            raise InputError("gather_block must be at least 1")
        self.all_lnl = []
        self.n_collectives = 0
        # A gather that comes in two halves (victor_amd.sharding.RcclGather: begin / finish over vk_comm_allgather_host_begin /
        # _finish, on a GPU context of its own) is enqueued when a block is complete and collected ONE BLOCK LATER: upload,
        # ncclAllGather and download run on the gather's stream while the walkers take the next block, nothing on the host
        # waits for the other ranks.  (A helper thread doing the blocking gather instead was measured and is slower than the
        # blocking gather itself: 143-164 against 101-105 us per collective, the two threads contending for the interpreter
        # and the HIP runtime.)  Plain callables (the ranks' socket group) are called as they are.
        self.overlap = bool(overlap) and hasattr(self.gather, "begin") and hasattr(self.gather, "finish")
        self._in_flight = None               # (k, W) of the block whose exchange is enqueued

    def _collect(self):
        """Bring home the block enqueued last, if any."""
        if self._in_flight is not None:
            k, W = self._in_flight
            self._in_flight = None
            self._append(np.asarray(self.gather.finish()), k, W)

    def _append(self, got, k, W):
        K, world = self.gather_block, self.dist.world
        got = got.reshape(world, K, W)
        self.all_lnl.extend(np.ascontiguousarray(got[:, t, :]).reshape(world * W) for t in range(k))

    def _padded(self, block):
        k, W = block.shape
        if k == self.gather_block:
            return np.ascontiguousarray(block)
        buf = np.full((self.gather_block, W), np.nan)        # a short last block travels at the full count (RCCL: equal counts, fixed buffers)
        buf[:k] = block
        return buf

    def _exchange(self, block):
        """One collective for the steps' log-likelihoods in ``block`` ``[k, W]``, k <= gather_block."""
        k, W = block.shape
        self.n_collectives += 1
        if self.overlap:
            self._collect()
            self.gather.begin(self._padded(block).reshape(self.gather_block * W))
            self._in_flight = (k, W)
        else:
            self._append(np.asarray(self.gather(self._padded(block).reshape(self.gather_block * W))), k, W)

    def run(self, n_steps):
        """Advance ``n_steps`` - the local sampler in pieces of its own block of 64 steps, whatever ``gather_block`` is, so that
        the chain (and every bit of its log-likelihoods) does not depend on how the history is exchanged - with one collective
        behind every ``gather_block`` steps."""
        chains, lnls = [], []
        pending = []                         # rows of log-likelihoods not yet exchanged
        held = done = 0
        piece = EnsembleMetropolis.BLOCK
        while done < n_steps:
            k = min(piece, n_steps - done)
            c, l = self.local.run(k)
            chains.append(c)
            lnls.append(l)
            done += k
            if not held and k == self.gather_block:      # the common case: a piece is a block
                self._exchange(l)
                continue
            pending.append(l)
            held += k
            if held >= self.gather_block:
                rows = np.concatenate(pending)
                full = (held // self.gather_block) * self.gather_block
                for a in range(0, full, self.gather_block):
                    self._exchange(rows[a:a + self.gather_block])
                pending = [rows[full:]] if full < held else []
                held -= full
        if held:
            self._exchange(np.concatenate(pending))
        self._collect()                      # every block of this run is in all_lnl when run() returns
        W, P = self.local.n_walkers, len(self.local.specs)
        chain = np.concatenate(chains) if chains else np.empty((0, W, P))
        lnl = np.concatenate(lnls) if lnls else np.empty((0, W))
        return chain, lnl, np.array(self.all_lnl)
