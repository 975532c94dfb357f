#!/usr/bin/env python3
"""Throughput of the likelihood hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole path (theory kernel + likelihood kernel [+ RCCL all-gather of lnL when
N > 1]) over one batch of synthetic parameter points per GPU.  Workload = BASELINE config 3, the grid the
metric is quoted on: 40 s bins x 100 mu x 50 velocity nodes, data multipoles l = 0,2,4 (N = 120),
anisotropic real-space xi (l = 0,2,4), AP-dependent template rescaling, sigma_v(r) template, batch of 65536
Halton points PER GPU (weak scaling: rank g evaluates points [g*B, (g+1)*B) of one global sequence).
Inputs are resident in HBM before the timed region; outputs stay in HBM.

For N > 1 the driver launches one process per GPU with torch.distributed.run; the ranks themselves never import torch:
RANK / WORLD_SIZE come from the environment, the rendezvous (RCCL unique id, barriers, max-over-ranks of the time) is a
standard-library socket group (victor_amd/rendezvous.py), the data path is libvictor_hip.so + RCCL.  Started on its own,
`python bench.py --gpus N` drives N contexts from ONE process (ncclCommInitAll, one grouped ncclAllGather per step).
After the weak-scaling line's timed loop a multi-GPU run also times a fixed global batch (65536 points in total,
`strong_scaling`) and every GPU's copy of the gathered lnL vector is checked against local recomputation of rows from every
other shard.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_legs import (  # noqa: E402,F401 - re-exported: tests and tools reach these through `bench`
    BATCH_PER_GPU, CONFIG, PEAK_FP64_VALU_TFLOPS, PEAK_HBM_GBS, PEAK_CLOCK_GHZ, F_PT_ANISO,
    F_PT_ISO, flops_per_eval, model_flops, cpu_worker, chain_worker, chains_sharing_one_gpu,
    stdout_to_stderr, host_cores, cpu_baseline, cpu_single_thread, warm_up, profiled_traffic,
    boss_measurement, at_sustained_clock, clock_fields, batch_sweep, api_latency, walker_rates,
    MODEL_OPTIONS, FROM_DATA_LABEL, from_data_options, option_rates, dsplit_measurement, Slot,
    make_gather, timed_steps, dsplit_sharded, walkers_distributed)


# Exit status: 0 = every measurement of the line is good; 1 = the headline's own checks failed (non-finite outputs, gathered vector
# differs from local recomputation) or a leg's `gather_matches_local` is false; 2 = a leg of the N > 1 run raised; 3 = a leg did not
# finish within LEG_TIMEOUT (a hung collective).  In cases 2 and 3 rank 0 has printed its line before the process ends: the headline
# stands, the leg carries {"error": ...}.
EXIT_LEG_FAILED = 2
EXIT_LEG_HUNG = 3
LEG_TIMEOUT = 300.0              # seconds the N > 1 legs (configs 4 and 5) may take together before rank 0 prints its line without them
REAL_STDOUT = os.dup(1)          # the stdout this program was started with


def clock_pass(args):
    """Child mode (`--clock-pass`, started by bench_pmc.live_clocks under `rocprofv3 --pmc GRBM_GUI_ACTIVE`): launch every
    workload the line quotes a roofline fraction for - the headline batch, BOSS CMASS, the model options - a fixed number of
    times, resident, and print what was launched in order ({label, warm, timed, event_ms}); the parent matches the profiler's
    dispatch records against it.  One theory kernel per call (batches <= 65536 are one launch)."""
    import victor_amd
    import workloads as cases
    from victor_amd.build import build_native
    build_native()
    done = []

    def run(label, fit, batch, with_beta, warm_s=0.15, timed=6, **kw):
        model = fit._merged(kw)
        eng = fit._get_engine(fit._engine_key(model))
        o = eng.make_opts(model, fit.fit_options)
        rows = fit._fit_rows(cases.halton_params(batch, with_beta=with_beta), model)
        bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
        eng.upload(bufs[0], rows)
        warm, t_end = 0, time.perf_counter() + warm_s       # as long a warm-up as the timed legs get; its launches are counted
        while warm < 2 or time.perf_counter() < t_end:
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
            eng.sync()
            warm += 1
        eng.timing(True)
        eng.read_timing(reset=True)
        for _ in range(timed):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        k1, _, launches = eng.read_timing(reset=True)
        eng.timing(False)
        for b in bufs:
            eng.free(b)
        done.append({"label": label, "warm": warm, "timed": timed, "event_ms": k1 / max(launches, 1), "kernel": eng.last_kernel()})

    model, data = cases.synth_options(CONFIG)
    model["numerics"] = {"simpson_even": args.simpson_even}
    run("config3", victor_amd.CCFFit(model, data), min(args.batch, 65536), False)
    boss = victor_amd.CCFFit(*cases.boss_options("config"))
    run("boss_cmass", boss, 16384, True)
    for label, kw in MODEL_OPTIONS:
        run(label, boss, 16384, True, timed=4, **kw)
    run(FROM_DATA_LABEL, victor_amd.CCFFit(*from_data_options()), 16384, True, timed=4)
    os.write(REAL_STDOUT, (json.dumps({"clock_pass": done}) + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="points per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boss", action="store_true", help="skip the secondary BOSS CMASS measurement")
    ap.add_argument("--no-chains", action="store_true", help="skip the chains-sharing-one-GPU measurement (child processes)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs that measure the theory kernel's HBM traffic (roofline.traffic)")
    ap.add_argument("--clock-pass", action="store_true",
                    help="child mode of the GRBM_GUI_ACTIVE counter pass: launch every measured workload a fixed number of times and "
                         "print the sequence (bench_pmc.live_clocks)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="oracle evaluations (default: about 10 per core)")
    ap.add_argument("--simpson-even", default="simpson",
                    help="even-N Simpson convention of the velocity integral: 'simpson' (SciPy >= 1.11, default) or "
                         "'avg' (SciPy < 1.11); same cost, recorded in config.simpson_even")
    args = ap.parse_args()
    if args.clock_pass:
        return clock_pass(args)

    # Two ways to several GPUs, no torch and no MPI binding in either (victor_amd/rendezvous.py, vk_comm_*):
    #   launched   one process per GPU under a launcher (the driver: torch.distributed.run; mpirun / srun work the same) -
    #              RANK / WORLD_SIZE from the environment, a standard-library socket group for the rendezvous and the barriers,
    #              ncclCommInitRank + ncclAllGather on the context's stream;
    #   standalone `python bench.py --gpus N`: ONE process, N contexts, ncclCommInitAll + a grouped ncclAllGather per step.
    from victor_amd.sharding import Dist
    dist = Dist()
    launched = dist.launched
    if launched:
        if dist.rank == 0:
            from victor_amd.build import build_native as _build     # (a no-op when the library is current) before the others
            _build()                                                # are kept waiting in a collective
        dist.connect(timeout=600.0)
    rank, world = dist.rank, dist.world
    if launched and world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    n_local = 1 if launched else max(args.gpus, 1)       # contexts this process drives
    total = world * n_local                              # GPUs of the run

    import numpy as np
    import victor_amd  # noqa: F401
    from victor_amd.build import build_native
    from victor_amd.engine import Engine
    import workloads as cases
    if rank == 0:
        build_native()
    dist.barrier()

    model, data = cases.synth_options(CONFIG)
    model["numerics"] = {"simpson_even": args.simpson_even}
    B = args.batch
    hp_all = cases.halton_params(B * total)

    # CPU baseline first: it spawns worker processes, which must happen before this process touches the GPU
    base = None
    if rank == 0 and total == 1 and not args.no_cpu_baseline:
        cores = host_cores()
        ns = min(args.cpu_sample or max(16, 40 * cores), B)      # ~40 oracle evaluations per core (~50 ms each): ~30 core-seconds of CPU work
        sel = np.linspace(0, B - 1, ns).astype(int)
        sample = [cases.point(hp_all, int(i)) for i in sel]
        base, vals, theory_o = cpu_baseline(sample, args.simpson_even)
        base["single_thread"] = cpu_single_thread(sample[:: max(1, len(sample) // 64)][:64], args.simpson_even)
    # HBM traffic of the dominant kernel, measured live: two child runs under rocprofv3 --pmc, again before the GPU is touched
    # ... and the shader clock every measured kernel sustains, from a third child run (GRBM_GUI_ACTIVE): the clock behind
    # `frac_at_sustained_clock` is measured in THIS run on THIS box, or the field is null
    traffic_live = clocks = None
    if rank == 0 and total == 1 and not args.no_live_traffic and not args.no_boss:
        import bench_pmc
        gdir = os.path.join(ROOT, "gpurun_out")
        gdir = gdir if os.path.isdir(gdir) else None
        traffic_live = bench_pmc.live_traffic(B, args.simpson_even, gdir)
        clocks = bench_pmc.live_clocks(B, args.simpson_even, gdir)
    # the reference's own calling convention under load (P chains, one point per call): child processes again, so before the GPU
    chains = None
    if rank == 0 and total == 1 and not args.no_boss and not args.no_chains:
        try:
            chains = chains_sharing_one_gpu()
        except Exception as exc:       # noqa: BLE001 - a secondary leg must not take the headline measurement down with it
            chains = {"error": repr(exc)}

    # one context per GPU; on a box with fewer GPUs than contexts (rehearsals) contexts share devices and no RCCL
    # communicator can be built, which exercises the host-gather fallback below
    from victor_amd import _native
    n_dev = max(_native.load().vk_device_count(), 1)
    slots = [Slot(rank * n_local + i, ((dist.local_rank if launched else i) % n_dev), model, data, hp_all, B, total,
                  total > 1 or launched) for i in range(n_local)]
    engines = [s.eng for s in slots]
    lead = slots[0]
    fit, eng = lead.fit, lead.eng
    N = eng.n_data

    # the gather (victor_amd/sharding.py: DeviceGather): "rank" = one communicator per process, "group" = one process, grouped
    # calls, "host" = degraded mode (RCCL unavailable or refused), "none" = a single GPU started without a launcher
    gatherer = make_gather(dist, engines, launched, [s.d_lnl for s in slots], [s.d_all for s in slots], B)
    gather = gatherer.mode

    def gather_step(n):
        gatherer([s.d_lnl for s in slots], [s.d_all for s in slots], n)

    def step(n=B):
        for s in slots:
            s.launch(n)
        gather_step(n)

    def sync_all():
        for e in engines:
            e.sync()

    def timed(n):
        """K steps bracketed by sync + barrier on both sides; returns (max-over-ranks seconds, theory ms per launch per slot)."""
        sync_all()
        dist.barrier()
        for e in engines:
            e.timing(True)
            e.read_timing(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(n)
        sync_all()
        dist.barrier()
        el = time.perf_counter() - t0
        per = []
        for e in engines:
            th, lk, launches = e.read_timing(reset=True)
            e.timing(False)
            per.append((th / max(launches, 1), lk / max(launches, 1)))
        return dist.max_float(el), per

    warm_up(eng, step)                 # untimed pre-warm (runtime's one-off post-allocation stall), then the W steps
    for _ in range(args.warmup):
        step()
    elapsed, per_slot = timed(B)
    theory_ms, like_ms = per_slot[0]

    kernel_name = eng.last_kernel()
    lnl = eng.download(lead.d_lnl, B)
    chi2 = eng.download(lead.d_chi, B)
    ok = True
    gathered_ok = None
    # Every slot checks the WHOLE gathered vector: its own part bit for bit, every other slot's part finite and - for a few
    # rows spread over that slot's shard - equal to this slot's own evaluation of the same global Halton points (another
    # batch size, hence possibly another kernel mapping: agreement to rounding).
    good = True
    for s in slots:
        mine_l = s.eng.download(s.d_lnl, B)
        ok = ok and bool(np.all(np.isfinite(mine_l)) and np.all(s.eng.download(s.d_chi, B) > 0))
        if gather == "none":
            continue
        gathered = s.eng.download(s.d_all, B * total)
        good = good and bool(np.array_equal(gathered[s.g * B:(s.g + 1) * B], mine_l)) and bool(np.all(np.isfinite(gathered)))
        probe = np.unique(np.linspace(0, B - 1, 6).astype(int))
        for other in range(total):
            if other == s.g:
                continue
            theirs = {k: v[other * B + probe] for k, v in hp_all.items()}
            own_l, _ = s.fit.log_likelihood_batch(theirs)
            good = good and bool(np.max(np.abs(gathered[other * B + probe] - own_l)) <= 1e-9 * np.max(np.abs(own_l)))
    if gather != "none":
        gathered_ok = bool(dist.min_float(1.0 if good else 0.0))
    k_local = np.array([p[0] for p in per_slot])
    kernel_ms_ranks = (dist.allgather_host(k_local, len(k_local)) if launched else k_local).tolist() if total > 1 or launched else None

    # fixed global batch next to the weak-scaling line: BATCH_PER_GPU points in total, B / total per GPU
    strong = None
    if total > 1:
        Bs = max(1, args.batch // total)
        for s in slots:
            mine_s = {k: v[s.g * Bs:(s.g + 1) * Bs] for k, v in hp_all.items()}
            s.eng.upload(s.d_rows, s.fit._fit_rows(mine_s, s.fit.model))
        for _ in range(max(args.warmup, 1)):
            step(Bs)
        el_s, per_s = timed(Bs)
        ks_local = np.array([p[0] for p in per_s])
        ks = (dist.allgather_host(ks_local, len(ks_local)) if launched else ks_local).tolist()
        strong = {"global_batch": Bs * total, "batch_per_gpu": Bs, "value": Bs * total * args.steps / el_s, "unit": "evals/s",
                  "ms_per_step": 1e3 * el_s / args.steps, "scaling": "strong", "kernel": eng.last_kernel(),
                  "theory_kernel_ms_per_rank": ks}

    comm_info = _native.comm_info() if (rank == 0 and (total > 1 or launched)) else None
    # what RCCL itself saw: ncclCommCount / ncclCommUserRank / ncclCommCuDevice of every rank's LIVE communicator beside its PCI bus id
    rank_records = gatherer.rank_records() if (total > 1 or launched) else None
    if comm_info is not None:
        comm_info["ranks"] = rank_records
    out = None
    if rank == 0:
        value = B * total * args.steps / elapsed
        k1_ms, k2_ms = theory_ms, like_ms
        aniso = not fit.model["assume_isotropic"]
        F = flops_per_eval(len(fit.s), 100, 50, len(fit.poles_s), aniso)
        achieved_tf = F * B / (k1_ms * 1e-3) / 1e12 if k1_ms > 0 else None
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (MI355X_MICROARCH.md): made by this run
        # itself as child processes (live_traffic); the last profiled figure is reported beside it with its source
        traffic_profiled = profiled_traffic("config3", B)
        alg_bytes = (8 * 12 + 16) * B     # a parameter row of VK_NPAR = 12 doubles in, lnL + chi2 out, per evaluation
        out = {
            "metric": "likelihood evals/sec (40 s-bins, 100 mu, l=0,2,4)",
            "value": value, "unit": "evals/s", "n_gpus": total, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: synthetic 40 s x 100 mu x 50 v grid, xi_r l=0,2,4, data l=0,2,4 "
                                   "(N=120), AP-dependent rescale, sigma_v(r) template, gaussian likelihood",
                       "simpson_even": eng.simpson_even + (" (SciPy >= 1.11 simps rule; 'avg' = SciPy < 1.11)"
                                                           if eng.simpson_even == "simpson" else " (SciPy < 1.11 simps rule)"),
                       "batch_per_gpu": B, "global_batch": B * total, "parallelism": f"batch-sharded x{total}",
                       "processes": world, "contexts_per_process": n_local,
                       "rendezvous": "standard-library socket group (victor_amd/rendezvous.py)" if launched else "none (one process)",
                       "gather": {"rank": "rccl allgather of lnL (ncclCommInitRank, one process per GPU)",
                                  "group": "rccl allgather of lnL (ncclCommInitAll, grouped calls, one process)",
                                  "host": "host allgather of lnL (RCCL unavailable or refused)",
                                  "none": "none (single GPU)"}[gather]},
            "roofline": {"bound": "fp64-valu", "kernel": kernel_name + "<3,3>",
                         "achieved": achieved_tf, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / PEAK_FP64_VALU_TFLOPS if achieved_tf else None,
                         # against the peak at the clock this kernel sustained (peak x clock / 2.4 GHz): the clock is measured by a
                         # child pass of THIS run (bench_pmc.live_clocks) - null when that pass could not be made
                         **clock_fields(F * B, clocks, "config3"),
                         "clock_method": __import__("bench_pmc").CLOCK_METHOD if clocks else None,
                         # HBM bytes per launch of this kernel from PMC counters: measured in this run (live_traffic) when the
                         # profiler is there, else null with the last profiled figure beside it
                         "traffic": traffic_live["bytes_per_launch"] if traffic_live else None,
                         "traffic_measured": traffic_live,
                         "traffic_ratio_to_algorithmic": traffic_live["bytes_per_launch"] / alg_bytes if traffic_live else None,
                         "traffic_profiled": traffic_profiled, "flops_per_eval": F, "kernel_ms": k1_ms,
                         "note": "path is FP64 vector-ALU bound (no MFMA, ~1e-5 of HBM peak); sqrt/div/exp counted "
                                 "as one flop each per SURVEY.md 8(d)"},
            "roofline_hbm": {"bound": "hbm", "achieved": alg_bytes / ((k1_ms + k2_ms) * 1e-3) / 1e9 if k1_ms else None,
                             "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": alg_bytes / ((k1_ms + k2_ms) * 1e-3) / 1e9 / PEAK_HBM_GBS if k1_ms else None,
                             "bytes_per_eval": 112},
            "kernels_ms": {"theory": k1_ms, "likelihood": k2_ms},
            "outputs_finite": ok,
        }
        if gathered_ok is not None:
            out["gather_matches_local"] = gathered_ok       # every rank, every slot (see above)
        if kernel_ms_ranks is not None:
            out["config"]["rccl"] = comm_info
            out["theory_kernel_ms_per_rank"] = kernel_ms_ranks
        if strong is not None:
            out["strong_scaling"] = strong
        if total == 1 and not args.no_boss:
            out["boss_cmass"] = boss_measurement(args, clocks=clocks)
            out["batch_sweep"] = batch_sweep()
            out["dsplit5"] = dsplit_measurement()
            out["host_api"] = api_latency()
            if chains is not None:
                out["chains_sharing_one_gpu"] = chains
            out["walker_ensembles"] = walker_rates()
            out["model_options"] = option_rates(clocks=clocks)
        if base is not None:
            chi_o = np.array([v[1] for v in vals])
            lnl_o = np.array([v[0] for v in vals])
            out["cpu_baseline"] = {"value": base["evals_per_s"], "unit": "evals/s", "cores": base["cores"],
                                   "value_per_core": base["evals_per_s"] / base["cores"], "kind": "port",
                                   "single_thread": {"value": base["single_thread"]["evals_per_s"], "unit": "evals/s",
                                                     "sample": f"{base['single_thread']['n']} of the same points, one process, "
                                                               f"one thread, {base['single_thread']['busy_s']:.1f} s busy"},
                                   "sample": f"{ns} of the {B} batch points through oracle/victor_oracle.py "
                                             f"(NumPy/SciPy restatement, bit-identical to the reference here), "
                                             f"{base['cores']} processes x 1 thread, {base['busy_s']:.1f} s busy"}
            out["max_rel_dchi2_vs_oracle"] = float(np.max(np.abs(chi2[sel] / chi_o - 1)))
            out["max_abs_dchi2_vs_oracle"] = float(np.max(np.abs(chi2[sel] - chi_o)))
            out["max_rel_dlnl_vs_oracle"] = float(np.max(np.abs(lnl[sel] / lnl_o - 1)))
            kk = sorted(theory_o)
            th_g = fit.theory_vector_batch({k: v[sel[kk]] for k, v in lead.mine.items()})
            th_o = np.array([theory_o[k] for k in kk])
            scale = np.max(np.abs(th_o.reshape(len(kk), len(fit.poles_s), -1)), axis=2, keepdims=True)
            dxi = np.abs(th_g - th_o).reshape(len(kk), len(fit.poles_s), -1) / scale
            out["max_rel_dxi_ell_vs_oracle"] = float(dxi.max())      # relative to max|xi_l| of each multipole

    # The ONE line of rank 0 (written to the stdout this program was started with, whatever a native library's banner has been
    # routed to meanwhile), exactly once - from here or from the watchdog of the legs below.
    import threading
    line_lock, line_out = threading.Lock(), [False]
    headline_ok = ok and gathered_ok is not False

    def emit(extra=None):
        with line_lock:
            if line_out[0]:
                return
            line_out[0] = True
            if rank == 0:
                if extra:
                    out.update(extra)
                os.write(REAL_STDOUT, (json.dumps(out) + "\n").encode())

    # BASELINE configs 4 and 5 at N > 1: the density-split joint fit with its global batch of 16384 sharded over the GPUs, and
    # 8 Metropolis walkers per GPU on the BOSS cobaya configuration with the block gather of their log-likelihoods.  The headline
    # above is complete by now; these legs build communicators of their own, so they run under a watchdog: a leg that raises is
    # reported as {"error": ...}, a collective that hangs costs LEG_TIMEOUT seconds - never the headline's record.
    legs = {}
    if total > 1 and not args.no_boss:
        gatherer.close()                      # one communicator at a time: every leg builds its own on its own contexts
        for s_ in slots:
            s_.free()
        slots = []

        def give_up():
            msg = f"did not finish within {LEG_TIMEOUT:.0f} s (a hung collective?); the measurements above are unaffected"
            emit({k: legs.get(k, {"error": msg}) for k in ("dsplit5", "walker_ensembles")})
            os._exit(EXIT_LEG_HUNG)          # the line is out; a hung run is not a success (nothing is restarted from here)

        timer = threading.Timer(LEG_TIMEOUT, give_up)
        timer.daemon = True
        timer.start()
        for name, leg in (("dsplit5", lambda: dsplit_sharded(dist, launched, n_local, n_dev, total, args.steps, max(args.warmup, 1))),
                          ("walker_ensembles", lambda: walkers_distributed(dist, launched, n_local, n_dev, total))):
            try:
                legs[name] = leg()
            except Exception as exc:       # noqa: BLE001 - reported in the line; the other ranks' watchdogs end a collective left half-way
                legs[name] = {"error": repr(exc)}
                print(f"rank {rank}: leg {name} failed: {exc!r}", file=sys.stderr)
        timer.cancel()
    emit(legs)

    gatherer.close()
    for s_ in slots:
        s_.free()
    failed_leg = any("error" in leg for leg in legs.values())
    if failed_leg:                           # ranks may be out of step with each other: no further collective, no destructors
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(EXIT_LEG_FAILED)            # the line (with the leg's {"error": ...}) is out; a failed leg is not a success
    dist.barrier()
    dist.close()
    legs_ok = all(leg.get("gather_matches_local") is not False for leg in legs.values())
    if not headline_ok or not legs_ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
