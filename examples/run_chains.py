#!/usr/bin/env python3
"""The reference's own way of sampling - independent chains, one process each, one likelihood per call (cobaya under
``mpirun``, reference README.md:30, CCFLikelihood.py:32-39) - on ONE GPU.

    python examples/run_chains.py --chains 8 --steps 2000

starts CHAINS processes; each builds the cobaya plug-in (victor/likelihoods/CCFLikelihood.py) from
config/boss_cobaya_config.yaml, draws a start from the `ref` distributions of its `params` block and runs a random-walk
Metropolis chain by calling ``calculate(state, **one_point)`` per step - what ``mpirun -n CHAINS cobaya-run`` does, minus
cobaya (which is not installed here).  The processes run with ``VICTOR_HIP_BROKER=auto`` in their environment: the first one
starts a GPU owner process (victor_amd/broker.py), all of them attach to its mailboxes, none of them opens the GPU; every
likelihood value is bit-identical to what the chain would compute on a GPU context of its own.  ``--direct`` gives each chain
its own context instead (a GPU box admits only a handful of such processes).

Under a real launcher nothing but the environment variable is needed:

    mpirun -n 16 -x VICTOR_HIP_BROKER=auto cobaya-run config/boss_cobaya_config.yaml

Prints one JSON line: aggregate evaluations per second, acceptance, R-1 over the chains, posterior means.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def chain(idx, config, steps, seed, broker, barrier, queue):
    """One chain (a child process)."""
    try:
        import numpy as np
        import yaml
        os.chdir(ROOT)                               # the data paths in the config are relative to the repository root
        if broker:
            os.environ["VICTOR_HIP_BROKER"] = broker
        else:
            os.environ.pop("VICTOR_HIP_BROKER", None)
        sys.path.insert(0, os.path.join(ROOT, "victor", "likelihoods"))
        from CCFLikelihood import CCFLikelihood
        from victor_amd.sampler import parse_cobaya_params
        with open(config) as fh:
            info = yaml.full_load(fh)
        lk_info = info["likelihood"]["CCFLikelihood"]
        lk = CCFLikelihood({"model": lk_info["model"], "data": lk_info["data"]})
        specs, fixed = parse_cobaya_params(info["params"])
        rng = np.random.default_rng(seed + 7919 * idx)
        lo = np.array([s.lo for s in specs])
        hi = np.array([s.hi for s in specs])
        width = np.array([s.proposal for s in specs])
        names = [s.name for s in specs]
        while True:
            x = np.array([s.ref_loc for s in specs]) + np.array([s.ref_scale for s in specs]) * rng.standard_normal(len(specs))
            if np.all((x >= lo) & (x <= hi)):
                break
        state = {}

        def logp(v):
            lk.calculate(state, want_derived=True, **dict(zip(names, v.tolist())), **fixed)
            return state["logp"]

        cur = logp(x)
        barrier.wait(timeout=600)
        samples = np.empty((steps, len(specs)))
        accepted = evals = 0
        t0 = time.perf_counter()
        for t in range(steps):
            prop = x + width * rng.standard_normal(len(specs))
            if np.all((prop >= lo) & (prop <= hi)):
                new = logp(prop)
                evals += 1
                if np.log(rng.random()) < new - cur:
                    x, cur = prop, new
                    accepted += 1
            samples[t] = x
        dt = time.perf_counter() - t0
        from victor_amd import _native
        queue.put((idx, samples, accepted, evals, dt, cur, _native._lib is None, None))
    except Exception as exc:       # noqa: BLE001 - reported to the parent
        try:
            barrier.abort()
        except Exception:
            pass
        queue.put((idx, None, 0, 0, 0.0, 0.0, False, repr(exc)))


def main():
    import multiprocessing as mp
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(ROOT, "config", "boss_cobaya_config.yaml"))
    ap.add_argument("--chains", type=int, default=8)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--direct", action="store_true", help="a GPU context per chain instead of the shared owner process")
    ap.add_argument("--broker", default="auto", help="value of VICTOR_HIP_BROKER for the chains (auto, or the name of a running owner)")
    args = ap.parse_args()

    import numpy as np
    from victor_amd.sampler import gelman_rubin
    ctx = mp.get_context("spawn")                    # fresh interpreters: this process never touches the GPU
    barrier, queue = ctx.Barrier(args.chains), ctx.Queue()
    procs = [ctx.Process(target=chain, args=(i, args.config, args.steps, args.seed, None if args.direct else args.broker, barrier, queue))
             for i in range(args.chains)]
    for p in procs:
        p.start()
    res = sorted((queue.get(timeout=3600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    errors = [r[7] for r in res if r[7]]
    if errors:
        print(json.dumps({"error": errors[0]}))
        sys.exit(1)
    chains = np.stack([r[1] for r in res], axis=1)                      # (steps, chains, params)
    burn = args.steps // 4
    evals, wall = sum(r[3] for r in res), max(r[4] for r in res)
    with open(args.config) as fh:
        import yaml
        names = [k for k, v in yaml.full_load(fh)["params"].items() if isinstance(v, dict) and "prior" in v]
    out = {"chains": args.chains, "steps": args.steps, "route": "a GPU context per chain" if args.direct else f"VICTOR_HIP_BROKER={args.broker}",
           "likelihood_evaluations": evals, "wall_s": wall, "evals_per_s": evals / wall, "us_per_call_per_chain": 1e6 * wall * args.chains / max(evals, 1),
           "acceptance": sum(r[2] for r in res) / max(evals, 1), "chains_never_loaded_the_gpu_library": all(r[6] for r in res),
           "mean": dict(zip(names, chains[burn:].mean(axis=(0, 1)).round(4).tolist())),
           "max_Rminus1": float(np.max(gelman_rubin(chains[burn:]))) if args.chains > 1 else None,
           "last_logp": [r[5] for r in res]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
