#!/usr/bin/env python3
"""BASELINE config [3]: Metropolis walkers on the BOSS cobaya configuration, W walkers per GPU.

Single GPU:      python examples/run_walkers.py --steps 200
Several GPUs:    python examples/run_walkers.py --gpus 8 --steps 200          (one process, one context per GPU)
             or  mpirun -n 8 python examples/run_walkers.py --steps 200      (one process per GPU; also srun, torchrun)

One process per GPU: every rank drives its own walkers on its own GPU (likelihood batches through libvictor_hip.so); the
log-likelihoods of all walkers are all-gathered over RCCL - one collective per block of 64 steps, the walkers never read it -
so that every rank can monitor the whole ensemble; the ranks find each other through a standard-library socket group
(victor_amd/rendezvous.py: MASTER_ADDR / MASTER_PORT or VICTOR_RDZV), no torch and no MPI binding.  One process for all GPUs: a single ensemble of gpus x walkers walkers whose
proposals are sharded over the devices, the log-likelihoods all-gathered on the GPUs by a grouped RCCL call.
Priors, starting distributions and proposal widths come from config/boss_cobaya_config.yaml, the file cobaya itself
would read.  Prints one JSON line with the acceptance rate, R-1 and posterior means.
"""

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(ROOT, "config", "boss_cobaya_config.yaml"))
    ap.add_argument("--walkers", type=int, default=8, help="walkers per GPU")
    ap.add_argument("--gpus", type=int, default=1, help="GPUs driven by THIS process (ignored under a launcher: one per rank)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--gather-block", type=int, default=64,
                    help="steps whose log-likelihoods are exchanged in one collective (1 = a blocking gather after every step)")
    ap.add_argument("--sampler", choices=["metropolis", "stretch"], default="metropolis",
                    help="random-walk Metropolis walkers, or the affine-invariant stretch-move ensemble")
    args = ap.parse_args()

    import numpy as np
    import yaml
    import victor_amd
    from victor_amd.sampler import (DistributedEnsemble, EnsembleMetropolis, EnsembleStretch, gelman_rubin,
                                    parse_cobaya_params)
    from victor_amd.sharding import Dist, RcclGather

    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")      # keep RCCL's log lines off stdout
    dist = Dist()
    if dist.launched:
        dist.connect()
    os.chdir(ROOT)                                   # the data paths in the config are relative to the repo root
    with open(args.config) as fh:
        info = yaml.full_load(fh)
    lk = info["likelihood"]["CCFLikelihood"]
    specs, fixed = parse_cobaya_params(info["params"])
    sampler = EnsembleStretch if args.sampler == "stretch" else EnsembleMetropolis
    gather = None
    multi = None
    if not dist.launched and args.gpus > 1:
        # one process, several GPUs: one ensemble, every batch of proposals sharded over the devices and its log-likelihoods
        # all-gathered on the GPUs (grouped RCCL call); through the host when no communicators can be built
        from victor_amd import _native
        from victor_amd.sharding import MultiGPUFit
        n_dev = max(_native.load().vk_device_count(), 1)
        multi = MultiGPUFit(lk["model"], lk["data"], devices=[i % n_dev for i in range(args.gpus)])
        if not multi.enable_rccl():
            print(f"RCCL gather unavailable ({multi._rccl_error}); gathering through the host", file=sys.stderr)
        ens = DistributedEnsemble(multi.log_likelihood_gathered, specs, args.walkers * args.gpus, dist, seed=args.seed,
                                  fixed=fixed, sampler=sampler, gather_block=args.gather_block)
        gather_name = "rccl (grouped, one process)" if multi._rccl else "host"
    else:
        from victor_amd import _native
        n_dev = max(_native.load().vk_device_count(), 1)     # more ranks than GPUs (a rehearsal): ranks share devices
        fit = victor_amd.CCFFit(lk["model"], lk["data"], device=(dist.local_rank % n_dev) if dist.launched else 0)
        engine = fit._get_engine()                           # tables on the GPU now: a failure here is not an RCCL problem
        # log-likelihoods of all ranks per step: RCCL all-gather on the engine's stream when there are several ranks and a
        # communicator can be built; the ranks' socket group otherwise (RCCL missing, ranks sharing a device)
        if dist.world > 1:
            ok = 1.0
            try:
                gather = RcclGather.own_context(fit, dist, args.walkers * args.gather_block)      # a context (stream) of its own
            except _native.CommInitTimeout as exc:                     # a thread is stuck inside RCCL on this context: fatal
                print(f"rank {dist.rank}: {exc}", file=sys.stderr)
                sys.stderr.flush()
                os._exit(4)
            except Exception as exc:                                   # every rank must take the same branch
                print(f"rank {dist.rank}: RCCL gather unavailable ({exc}); using the socket group", file=sys.stderr)
                ok = 0.0
            if dist.min_float(ok) == 0.0:
                if gather is not None:
                    gather.close()
                gather = None
        ens = DistributedEnsemble(lambda batch: fit.log_likelihood_batch(batch)[0], specs, args.walkers, dist,
                                  seed=args.seed, fixed=fixed, gather=gather, sampler=sampler, fit=fit,
                                  gather_block=args.gather_block)
        gather_name = "rccl" if gather is not None else "host"
    # the first evaluation of a process pays for the HIP runtime, the code object and the device tables (~0.25 s): timed apart
    t0 = time.perf_counter()
    ens.local.initialise()
    first = time.perf_counter() - t0
    evals0 = ens.local.n_evals
    t0 = time.perf_counter()
    chain, lnl, all_lnl = ens.run(args.steps)
    wall = time.perf_counter() - t0
    if gather is not None:
        gather.close()
    if multi is not None:
        multi.close()
    if dist.rank == 0:
        burn = args.steps // 4
        print(json.dumps({
            "walkers_total": args.walkers * dist.world * (args.gpus if multi is not None else 1), "steps": args.steps, "wall_s": wall,
            "first_evaluation_s": first,
            "likelihood_evaluations": (ens.local.n_evals - evals0) * dist.world,
            "acceptance": ens.local.acceptance,
            "max_Rminus1": float(np.max(gelman_rubin(chain[burn:]))),
            "mean": dict(zip(ens.local.names, chain[burn:].mean(axis=(0, 1)).round(4).tolist())),
            "best_lnl_over_all_ranks": float(all_lnl.max()),
            "gathered_shape": list(all_lnl.shape), "gathered_sum": float(all_lnl.sum()),
            "gathered_sha256": hashlib.sha256(np.ascontiguousarray(all_lnl).tobytes()).hexdigest(),
            "gather": gather_name, "gather_block": ens.gather_block, "collectives": ens.n_collectives}))
    dist.barrier()
    dist.close()


if __name__ == "__main__":
    main()
