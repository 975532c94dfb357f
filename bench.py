#!/usr/bin/env python3
"""Throughput of the likelihood hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole path (theory kernel + likelihood kernel [+ RCCL all-gather of lnL when
N > 1]) over one batch of synthetic parameter points per GPU.  Workload = BASELINE config 3, the grid the
metric is quoted on: 40 s bins x 100 mu x 50 velocity nodes, data multipoles l = 0,2,4 (N = 120),
anisotropic real-space xi (l = 0,2,4), AP-dependent template rescaling, sigma_v(r) template, batch of 65536
Halton points PER GPU (weak scaling: rank g evaluates points [g*B, (g+1)*B) of one global sequence).
Inputs are resident in HBM before the timed region; outputs stay in HBM.

For N > 1 the driver launches one process per GPU with torch.distributed.run; torch is used only for the
host-side rendezvous (gloo: barriers, the RCCL unique id, max-over-ranks of the time).  The data path
is libvictor_hip.so + RCCL.  After the weak-scaling line's timed loop a multi-rank run also times a fixed global batch
(65536 points in total, `strong_scaling`) and every rank checks the whole gathered lnL vector against its own
recomputation of rows from every other rank's shard.
"""

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH_PER_GPU = 65536
CONFIG = 3
PEAK_FP64_VALU_TFLOPS = 78.6     # MI355X vector FP64: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# flops per integrand point for l_r = 0,2,4 (sqrt, divide, exp counted as one each)
F_PT_ANISO = 77
F_PT_ISO = 47


def flops_per_eval(n_s, n_mu, n_x, n_ell, aniso):
    """Algorithmic flops of the theory kernel per evaluation (SURVEY.md 8d without the chi-square term)."""
    return n_s * n_mu * n_x * (F_PT_ANISO if aniso else F_PT_ISO) + 2 * n_ell * n_s * n_mu


def cpu_worker(args):
    """Time the oracle on a slice of the sample (runs in a child process, one per host core)."""
    idx, pts, rule = args
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = "1"
    import warnings
    warnings.filterwarnings("ignore")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import victor_oracle as vo
    from tests import cases
    model, data = cases.synth_options(CONFIG)
    model["numerics"] = {"simpson_even": rule}
    fit = vo.OracleFit(model, data)
    out = []
    t0 = time.perf_counter()
    for p in pts:
        out.append(fit.log_likelihood(dict(p)))
    busy = time.perf_counter() - t0
    theory = [fit.theory_multipole_vector(fit.s, dict(p), fit.poles_s) for p in pts[:2]]   # untimed: for the xi_l check
    return idx, busy, out, theory


class stdout_to_stderr:
    """Route file descriptor 1 to stderr while native libraries (gloo, RCCL) print their banners, so that the only thing
    this program ever writes to stdout is rank 0's JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def host_cores():
    """Cores this job may really use: affinity mask, capped by the cgroup CPU quota and by 16 (the GPU box's
    documented share per GPU; the visible 256 hardware threads are not ours - 256 workers ran at 0.6 evals/s each
    against 8-9 on a free core)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    cores = min(cores, max(1, quota // period))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(cores, int(os.environ.get("VICTOR_BENCH_CORES", "16"))))


def cpu_baseline(sample_pts, rule):
    """Oracle ('port' of the reference algorithm) on the host cores, bounded sample."""
    import multiprocessing as mp
    cores = max(1, min(host_cores(), len(sample_pts)))
    chunks = [(i, sample_pts[i::cores], rule) for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(cpu_worker, chunks)
    wall = time.perf_counter() - t0
    busy = max(r[1] for r in res)          # excludes interpreter start-up and table construction
    vals = [None] * len(sample_pts)
    theory = {}
    for idx, _, out, th in res:
        for k, v in enumerate(out):
            vals[idx + k * cores] = v
        for k, t in enumerate(th):
            theory[idx + k * cores] = t
    return {"evals_per_s": len(sample_pts) / busy, "cores": cores, "wall_s": wall, "busy_s": busy}, vals, theory


def cpu_single_thread(sample_pts, rule):
    """The same oracle in ONE process on ONE thread (SURVEY.md 8(d) leg (i)), timed in a child process so that this
    process has not touched the GPU yet and the BLAS thread count is pinned before NumPy is imported."""
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(1) as pool:
        _, busy, out, _ = pool.map(cpu_worker, [(0, sample_pts, rule)])[0]
    return {"evals_per_s": len(sample_pts) / busy, "busy_s": busy, "n": len(sample_pts)}


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def warm_up(eng, launch, seconds=0.3):
    """Run ``launch`` back to back for ``seconds``: the HIP runtime stalls once (~75 ms, kernels unaffected) some tens
    of milliseconds after fresh allocations, which must not land in a short timed window."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            launch()
        eng.sync()


def boss_measurement(args, batch=16384, steps=20):
    """Secondary figure: the BOSS DR12 CMASS configuration the north star's 1e5 evals/s target is quoted on
    (config/boss_config.yaml: 30 s bins x 100 mu x 50 v, l = 0,2, reconstruction-beta dependent tables, data and
    covariance, Sellentin-Heavens likelihood).  Inputs resident in HBM; same timing discipline as the main line."""
    import victor_amd
    from tests import cases
    fit = victor_amd.CCFFit(*cases.boss_options("config"))
    eng = fit._get_engine()
    opts = eng.make_opts(fit.model, fit.fit_options)
    rows = fit._fit_rows(cases.halton_params(batch, with_beta=True), fit.model)
    d_rows, d_lnl, d_chi, d_ws = eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)
    eng.upload(d_rows, rows)
    warm_up(eng, lambda: eng.eval_device_async(opts, d_rows, batch, d_lnl, d_chi, d_ws))
    eng.timing(True)
    eng.read_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.eval_device_async(opts, d_rows, batch, d_lnl, d_chi, d_ws)
    eng.sync()
    dt = time.perf_counter() - t0
    k1, k2, launches = eng.read_timing(reset=True)
    eng.timing(False)
    import numpy as np
    lnl = eng.download(d_lnl, batch)
    for p in (d_rows, d_lnl, d_chi, d_ws):
        eng.free(p)
    F = flops_per_eval(30, 100, 50, 2, False)
    k1 /= max(launches, 1)
    return {"evals_per_s": batch * steps / dt, "batch": batch, "steps": steps, "kernel": eng.last_kernel() + "<1,2>",
            "kernels_ms": {"theory": k1, "likelihood": k2 / max(launches, 1)},
            "fp64_valu_frac": F * batch / (k1 * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS if k1 > 0 else None,
            "flops_per_eval": F, "all_finite": bool(np.all(np.isfinite(lnl)))}


def batch_sweep():
    """evals/s with inputs resident in HBM at the batch sizes SURVEY.md 8(d) asks for: BASELINE config [1]
    (batch 1024, isotropic xi_r, l = 0,2) and the metric grid (config 3) at batch 1, 64 and 1024."""
    import victor_amd
    from tests import cases
    res = {}
    for config, batches in ((2, (1024,)), (3, (1, 64, 1024))):
        fit = victor_amd.CCFFit(*cases.synth_options(config))
        eng = fit._get_engine()
        opts = eng.make_opts(fit.model, fit.fit_options)
        for batch in batches:
            rows = fit._fit_rows(cases.halton_params(batch), fit.model)
            bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
            eng.upload(bufs[0], rows)
            steps = 200 if batch <= 64 else 50
            warm_up(eng, lambda: eng.eval_device_async(opts, bufs[0], batch, bufs[1], bufs[2], bufs[3]))
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.eval_device_async(opts, bufs[0], batch, bufs[1], bufs[2], bufs[3])
            eng.sync()
            dt = (time.perf_counter() - t0) / steps
            res[f"config{config}_batch{batch}"] = {"evals_per_s": batch / dt, "ms_per_batch": dt * 1e3,
                                                   "kernel": eng.last_kernel()}
            for p in bufs:
                eng.free(p)
    return res


def api_latency():
    """Wall-clock of the reference's calling convention - one parameter point per call (CCFLikelihood.py:32-39) - through the
    Python API, host buffers in and out, and the PCIe-inclusive rate of a full host-buffer batch."""
    import victor_amd
    from tests import cases
    res = {}
    for name, opts, beta in (("config3", cases.synth_options(CONFIG), False), ("boss_cmass", cases.boss_options("config"), True)):
        fit = victor_amd.CCFFit(*opts)
        hp = cases.halton_params(BATCH_PER_GPU, with_beta=beta)
        p = cases.point(hp, 3)
        t_end = time.perf_counter() + 0.4      # past the runtime's one-off stall after fresh allocations (see warm_up)
        while time.perf_counter() < t_end:
            fit.log_likelihood(p)
        t0 = time.perf_counter()
        for _ in range(2000):
            fit.log_likelihood(p)
        single = (time.perf_counter() - t0) / 2000
        rows = fit._fit_rows(hp, fit.model)
        for _ in range(3):
            fit.log_likelihood_batch(rows)
        t0 = time.perf_counter()
        for _ in range(5):
            fit.log_likelihood_batch(rows)
        full = (time.perf_counter() - t0) / 5
        res[name] = {"log_likelihood_single_point_us": single * 1e6,
                     "log_likelihood_batch_65536_host_buffers_ms": full * 1e3,
                     "host_buffer_evals_per_s": BATCH_PER_GPU / full}
    return res


def dsplit_measurement(batch=16384, steps=10):
    """BASELINE config 5 on one GPU: five table sets sharing one parameter batch (block-diagonal covariance, N = 5 x 120),
    one upload, the blocks on their own streams, sums on the device (vk_joint_eval_device_async)."""
    import numpy as np
    import victor_amd
    from victor_amd.joint import JointFit
    from tests import cases
    joint = JointFit([victor_amd.CCFFit(*cases.dsplit_options(q)) for q in range(5)])
    engines, opts = joint._plan({})
    rows = joint.fits[0]._fit_rows(cases.halton_params(batch), joint.fits[0].model)
    _, (d_rows, d_out, d_ws) = joint._device_buffers(engines, batch)
    lead = engines[0]
    lead.upload(d_rows, rows)
    d_chi = d_out + 8 * batch
    warm_up(lead, lambda: joint.eval_device_async(engines, opts, d_rows, batch, d_out, d_chi, d_ws))
    t0 = time.perf_counter()
    for _ in range(steps):
        joint.eval_device_async(engines, opts, d_rows, batch, d_out, d_chi, d_ws)
    lead.sync()
    dt = (time.perf_counter() - t0) / steps
    out = lead.download(d_out, 2 * batch)
    kernel = lead.last_kernel()
    g, meta = cases.golden_outputs()
    pts = meta["synth_points"][:6]
    chi6 = joint.log_likelihood_batch({k: np.array([q[k] for q in pts]) for k in pts[0]})[1]
    return {"joint_evals_per_s": batch / dt, "block_evals_per_s": 5 * batch / dt, "ms_per_batch": dt * 1e3, "batch": batch,
            "blocks": 5, "n_data": joint.n_data, "kernel": kernel, "all_finite": bool(np.all(np.isfinite(out))),
            "max_rel_dchi2_vs_reference_golden": float(np.max(np.abs(chi6 / g["dsplit_chi2"] - 1)))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="points per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boss", action="store_true", help="skip the secondary BOSS CMASS measurement")
    ap.add_argument("--cpu-sample", type=int, default=0, help="oracle evaluations (default: about 10 per core)")
    ap.add_argument("--simpson-even", default="simpson",
                    help="even-N Simpson convention of the velocity integral: 'simpson' (SciPy >= 1.11, default) or "
                         "'avg' (SciPy < 1.11); same cost, recorded in config.simpson_even")
    args = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        # start one rank per GPU as child processes (never exec: this process may already hold the GPU)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT") or str(free_port()),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    from victor_amd.sharding import Dist
    dist = Dist()
    if launched:
        with stdout_to_stderr():
            dist.init_process_group("gloo")
    rank, world = dist.rank, dist.world
    if launched and world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    import numpy as np
    import victor_amd
    from victor_amd.build import build_native
    from tests import cases
    if rank == 0:
        build_native()
    dist.barrier()

    model, data = cases.synth_options(CONFIG)
    model["numerics"] = {"simpson_even": args.simpson_even}
    B = args.batch
    hp_all = cases.halton_params(B * world)
    mine = {k: v[rank * B:(rank + 1) * B] for k, v in hp_all.items()}

    # CPU baseline first: it spawns worker processes, which must happen before this process touches the GPU
    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cores = host_cores()
        ns = min(args.cpu_sample or max(16, 40 * cores), B)      # ~40 oracle evaluations per core (~50 ms each): ~30 core-seconds of CPU work
        sel = np.linspace(0, B - 1, ns).astype(int)
        sample = [cases.point(mine, int(i)) for i in sel]
        base, vals, theory_o = cpu_baseline(sample, args.simpson_even)
        base["single_thread"] = cpu_single_thread(sample[:: max(1, len(sample) // 64)][:64], args.simpson_even)

    # one rank per GPU; on a box with fewer GPUs than ranks (rehearsals) ranks share devices and the RCCL communicator
    # cannot be built, which exercises the host-gather fallback below
    from victor_amd import _native
    n_dev = max(_native.load().vk_device_count(), 1)
    fit = victor_amd.CCFFit(model, data, device=(dist.local_rank % n_dev) if launched else 0)
    eng = fit._get_engine()
    opts = eng.make_opts(fit.model, fit.fit_options)
    rows = fit._fit_rows(mine, fit.model)
    N = eng.n_data

    d_rows = eng.alloc(rows.size)
    d_lnl = eng.alloc(B)
    d_chi = eng.alloc(B)
    d_ws = eng.alloc(B * N)
    d_all = eng.alloc(B * world) if world > 1 or launched else None
    eng.upload(d_rows, rows)
    use_comm = launched
    host_gather = False
    if use_comm:
        # RCCL logs (version banner, topology warnings) go to stdout by default; stdout is reserved for the JSON line
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        with stdout_to_stderr():
            rccl_ok = 1.0
            try:
                uid = eng.comm_unique_id() if rank == 0 else None
            except Exception as exc:                        # librccl missing: every rank must learn about it
                print(f"rank {rank}: RCCL unavailable ({exc})", file=sys.stderr)
                uid, rccl_ok = bytes(128), 0.0
            uid = dist.broadcast_bytes(uid, src=0, nbytes=128)
            rccl_ok = dist.min_float(rccl_ok)
            if rccl_ok:
                try:
                    eng.comm_init(uid, rank, world)
                    eng.comm_allgather_async(d_lnl, d_all, B)  # first collective builds the rings (and logs) here
                    eng.sync()
                except Exception as exc:
                    print(f"rank {rank}: RCCL communicator failed ({exc})", file=sys.stderr)
                    rccl_ok = 0.0
                rccl_ok = dist.min_float(rccl_ok)
        if not rccl_ok:
            # degraded mode, reported as such in the JSON line: gather lnL through the host process group
            use_comm = False
            host_gather = True

    def step():
        eng.eval_device_async(opts, d_rows, B, d_lnl, d_chi, d_ws)
        if use_comm:
            eng.comm_allgather_async(d_lnl, d_all, B)
        elif host_gather:
            eng.sync()
            step.gathered = dist.allgather_host(eng.download(d_lnl, B), B)

    warm_up(eng, step)                 # untimed pre-warm (runtime's one-off post-allocation stall), then the W steps
    for _ in range(args.warmup):
        step()
    eng.sync()
    dist.barrier()
    eng.timing(True)
    eng.read_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    theory_ms, like_ms, launches = eng.read_timing(reset=True)
    eng.timing(False)
    elapsed = dist.max_float(elapsed)

    kernel_name = eng.last_kernel()
    lnl = eng.download(d_lnl, B)
    chi2 = eng.download(d_chi, B)
    ok = bool(np.all(np.isfinite(lnl)) and np.all(chi2 > 0))
    # Every rank checks the WHOLE gathered vector: its own slot bit for bit, every other rank's slot finite and - for a few
    # rows spread over that rank's shard - equal to this rank's own evaluation of the same global Halton points (another
    # batch size, hence possibly another kernel mapping: agreement to rounding).
    gathered_ok = None
    gathered = None
    if use_comm:
        gathered = eng.download(d_all, B * world)
    elif host_gather:
        gathered = step.gathered
    if gathered is not None:
        good = bool(np.array_equal(gathered[rank * B:(rank + 1) * B], lnl)) and bool(np.all(np.isfinite(gathered)))
        probe = np.unique(np.linspace(0, B - 1, 6).astype(int))
        for other in range(world):
            if other == rank:
                continue
            theirs = {k: v[other * B + probe] for k, v in hp_all.items()}
            mine_l, _ = fit.log_likelihood_batch(theirs)
            good = good and bool(np.max(np.abs(gathered[other * B + probe] - mine_l)) <= 1e-9 * np.max(np.abs(mine_l)))
        gathered_ok = bool(dist.min_float(1.0 if good else 0.0))
    kernel_ms_ranks = dist.allgather_host(np.array([theory_ms / max(launches, 1)]), 1).tolist() if launched else None

    # fixed global batch next to the weak-scaling line: BATCH_PER_GPU points in total, B / world per rank
    strong = None
    if world > 1:
        Bs = max(1, args.batch // world)
        mine_s = {k: v[rank * Bs:(rank + 1) * Bs] for k, v in hp_all.items()}
        eng.upload(d_rows, fit._fit_rows(mine_s, fit.model))

        def step_s():
            eng.eval_device_async(opts, d_rows, Bs, d_lnl, d_chi, d_ws)
            if use_comm:
                eng.comm_allgather_async(d_lnl, d_all, Bs)
            elif host_gather:
                eng.sync()
                step_s.gathered = dist.allgather_host(eng.download(d_lnl, Bs), Bs)

        for _ in range(max(args.warmup, 1)):
            step_s()
        eng.sync()
        dist.barrier()
        eng.timing(True)
        eng.read_timing(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_s()
        eng.sync()
        dist.barrier()
        el_s = dist.max_float(time.perf_counter() - t0)
        th_s, _, n_s = eng.read_timing(reset=True)
        eng.timing(False)
        ks = dist.allgather_host(np.array([th_s / max(n_s, 1)]), 1).tolist()
        strong = {"global_batch": Bs * world, "batch_per_gpu": Bs, "value": Bs * world * args.steps / el_s, "unit": "evals/s",
                  "ms_per_step": 1e3 * el_s / args.steps, "scaling": "strong", "kernel": eng.last_kernel(),
                  "theory_kernel_ms_per_rank": ks}

    if rank == 0:
        total = B * world * args.steps
        value = total / elapsed
        k1_ms = theory_ms / max(launches, 1)
        k2_ms = like_ms / max(launches, 1)
        aniso = not fit.model["assume_isotropic"]
        F = flops_per_eval(len(fit.s), 100, 50, len(fit.poles_s), aniso)
        achieved_tf = F * B / (k1_ms * 1e-3) / 1e12 if k1_ms > 0 else None
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (MI355X_MICROARCH.md), which this run
        # does not make: `traffic` is null here and the last profiled figure is reported beside it with its source
        traffic_profiled = None
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.isfile(tfile):
            with open(tfile) as fh:
                tj = json.load(fh)
            if tj.get("batch") == B:
                traffic_profiled = {"bytes_per_launch": tj.get("theory_kernel_hbm_bytes_per_launch"),
                                    "source": tj.get("source", "profiles/traffic_latest.json"), "kernel": tj.get("kernel")}
        alg_bytes = (8 * 10 + 16) * B     # 80 B of parameters in, lnL + chi2 out, per evaluation
        out = {
            "metric": "likelihood evals/sec (40 s-bins, 100 mu, l=0,2,4)",
            "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 3: synthetic 40 s x 100 mu x 50 v grid, xi_r l=0,2,4, data l=0,2,4 "
                                   "(N=120), AP-dependent rescale, sigma_v(r) template, gaussian likelihood",
                       "simpson_even": eng.simpson_even + (" (SciPy >= 1.11 simps rule; 'avg' = SciPy < 1.11)"
                                                           if eng.simpson_even == "simpson" else " (SciPy < 1.11 simps rule)"),
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"batch-sharded x{world}",
                       "gather": "rccl allgather of lnL" if use_comm else
                       ("host allgather of lnL (RCCL unavailable)" if host_gather else "none (single process)")},
            "roofline": {"bound": "fp64-valu", "kernel": kernel_name + "<3,3>",
                         "achieved": achieved_tf, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved_tf / PEAK_FP64_VALU_TFLOPS if achieved_tf else None,
                         "traffic": None, "traffic_profiled": traffic_profiled, "flops_per_eval": F, "kernel_ms": k1_ms,
                         "note": "path is FP64 vector-ALU bound (no MFMA, ~1e-5 of HBM peak); sqrt/div/exp counted "
                                 "as one flop each per SURVEY.md 8(d)"},
            "roofline_hbm": {"bound": "hbm", "achieved": alg_bytes / ((k1_ms + k2_ms) * 1e-3) / 1e9 if k1_ms else None,
                             "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": alg_bytes / ((k1_ms + k2_ms) * 1e-3) / 1e9 / PEAK_HBM_GBS if k1_ms else None,
                             "bytes_per_eval": 96},
            "kernels_ms": {"theory": k1_ms, "likelihood": k2_ms},
            "outputs_finite": ok,
        }
        if gathered_ok is not None:
            out["gather_matches_local"] = gathered_ok       # every rank, every slot (see above)
        if launched:
            out["config"]["rccl"] = _native.comm_info()
            out["theory_kernel_ms_per_rank"] = kernel_ms_ranks
        if strong is not None:
            out["strong_scaling"] = strong
        if world == 1 and not args.no_boss:
            out["boss_cmass"] = boss_measurement(args)
            out["batch_sweep"] = batch_sweep()
            out["dsplit5"] = dsplit_measurement()
            out["host_api"] = api_latency()
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
            th_g = fit.theory_vector_batch({k: v[sel[kk]] for k, v in mine.items()})
            th_o = np.array([theory_o[k] for k in kk])
            scale = np.max(np.abs(th_o.reshape(len(kk), len(fit.poles_s), -1)), axis=2, keepdims=True)
            dxi = np.abs(th_g - th_o).reshape(len(kk), len(fit.poles_s), -1) / scale
            out["max_rel_dxi_ell_vs_oracle"] = float(dxi.max())      # relative to max|xi_l| of each multipole
        print(json.dumps(out))

    for p in (d_rows, d_lnl, d_chi, d_ws, d_all):
        if p:
            eng.free(p)
    if use_comm:
        eng.comm_destroy()
    dist.barrier()
    if dist.pg is not None:
        dist.pg.destroy_process_group()
    if not ok or gathered_ok is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
