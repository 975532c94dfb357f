"""The measurements behind bench.py's one JSON line, the headline's timed loop aside: the algorithmic flop counts, the CPU
baseline (oracle workers: child processes), the secondary single-GPU legs (BOSS CMASS, batch sweep, host API latencies, chains
sharing one GPU, walker ensembles, model options, the density-split joint fit) and the legs of a multi-GPU run (sharded joint
fit, distributed walkers).  bench.py imports what it needs; tools/ import the chain worker from here through `bench`."""

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH_PER_GPU = 65536
CONFIG = 3
PEAK_FP64_VALU_TFLOPS = 78.6     # MI355X vector FP64: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# flops per integrand point for l_r = 0,2,4 (sqrt, divide, exp counted as one each)
F_PT_ANISO = 77
F_PT_ISO = 47


PEAK_CLOCK_GHZ = 2.4             # the clock behind PEAK_FP64_VALU_TFLOPS


def flops_per_eval(n_s, n_mu, n_x, n_ell, aniso):
    """Algorithmic flops of the theory kernel per evaluation (SURVEY.md 8d without the chi-square term)."""
    return n_s * n_mu * n_x * (F_PT_ANISO if aniso else F_PT_ISO) + 2 * n_ell * n_s * n_mu


def model_flops(rsd, n_s, n_mu, n_x, n_ell, aniso=False, niter=5, n_data=0, linearised=False, coord_shift=True):
    """Algorithmic flops per evaluation of the other RSD models, counted from the reference's expressions with SURVEY.md 8(d)'s
    conventions (sqrt, divide, exp one flop each; a table look-up 9: index 2, local coordinate 1, cubic Horner 6; an fma 2) -
    the breakdown is DESIGN.md section 5, "Algorithmic flops of the other RSD models".  ``n_data`` > 0 adds the chi-square
    (2 N^2 + 3 N) for launches that take it in the same kernel.

    dispersion (ccf_model.py:658-671), per integrand point: numerator s_par - v/aH 2; first pass 1 (its 1 / (1 + q(s)) is per
        cell: 20); niter passes of [r^2 2, sqrt 1, 1/r 1, u = r/c 1, V look-up 9, q 2, 1 + q 1, divide 1] = 18; final geometry 6
        (r^2 2, sqrt, 1/r, mu_r, u); four look-ups (sigma_v, V, V', xi_0) 36; zero-mean pdf 6; Jacobian 10; accumulate 5
        = 66 + 18 niter = 156 at niter = 5;
    kaiser (:692-741), per (s, mu) cell: s_perp, s_par 4; first pass 18 (s 3, then as a pass without r^2); niter passes 18;
        final geometry 6; three look-ups (V, V', xi_0) 27; J 7; 1 / (1 + J) and (1 + M xi) J^-1 - 1: 5 = 67 + 18 niter = 157
        (linearised: 2 instead of 5; without the coordinate shift the 18 (1 + niter) go);
    euclid_special (:743-784): kaiser's J with other constants, xi = M xi_r - J: 2 = 154.
    Anisotropic real-space multipoles add two look-ups and the Legendre sum: 30 (as 47 -> 77 for streaming)."""
    cells = n_s * n_mu
    extra = 30 if aniso else 0
    tail = 2 * n_data * n_data + 3 * n_data
    if rsd == "streaming":
        return flops_per_eval(n_s, n_mu, n_x, n_ell, aniso) + tail
    if rsd == "dispersion":
        return cells * n_x * (66 + 18 * niter + extra) + cells * 20 + 2 * n_ell * cells + tail
    if rsd in ("kaiser", "euclid_special"):
        shift = 18 * (1 + niter) if coord_shift else 0
        last = 2 if (rsd == "euclid_special" or linearised) else 5
        return cells * (4 + shift + 6 + 27 + 7 + last + extra) + 2 * n_ell * cells + tail
    raise ValueError(rsd)


def cpu_worker(args):
    """Time the oracle on a slice of the sample (runs in a child process, one per host core)."""
    idx, pts, rule = args
    os.environ["OMP_NUM_THREADS"] = os.environ["OPENBLAS_NUM_THREADS"] = "1"
    import warnings
    warnings.filterwarnings("ignore")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import victor_oracle as vo
    import workloads as cases
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


def chain_worker(idx, n_chains, seconds, broker, barrier, queue):
    """One cobaya-style chain (runs in a child process): a CCFLikelihood built from config/boss_cobaya_config.yaml whose
    ``calculate(state, **one_point)`` is called in a loop, as ``mpirun -n P cobaya-run`` does it (reference README.md:30,
    CCFLikelihood.py:32-39).  ``broker``: value of VICTOR_HIP_BROKER for this chain, or None for a GPU context of its own."""
    try:
        os.chdir(ROOT)                      # the data paths in the config are relative to the repository root
        if broker:
            os.environ["VICTOR_HIP_BROKER"] = broker
        else:
            os.environ.pop("VICTOR_HIP_BROKER", None)
        sys.path.insert(0, os.path.join(ROOT, "victor", "likelihoods"))
        from CCFLikelihood import CCFLikelihood
        import workloads as cases
        info = cases.cobaya_info()["likelihood"]["CCFLikelihood"]
        lk = CCFLikelihood({"model": info["model"], "data": info["data"]})
        h = cases.halton(4096 + 64 * idx, bases=(2, 3, 5, 7))[64 * idx:]         # every chain its own points of the prior box
        pts = [{"fsigma8": 0.05 + 1.45 * a, "beta": 0.2 + 0.4 * b, "sigma_v": 100 + 400 * c, "epsilon": 0.8 + 0.4 * d}
               for a, b, c, d in h.tolist()]
        state = {}
        for p in pts[:64]:
            lk.calculate(state, want_derived=True, **p)
        first = []
        for p in pts[:4]:
            lk.calculate(state, want_derived=True, **p)
            first.append(state["logp"])
        barrier.wait(timeout=600)
        n, k = 0, 0
        t0 = time.perf_counter()
        t_end = t0 + seconds
        while time.perf_counter() < t_end:
            for p in pts[k:k + 32]:
                lk.calculate(state, want_derived=True, **p)
            n += 32
            k = (k + 32) % 4096
        dt = time.perf_counter() - t0
        barrier.wait(timeout=600)
        queue.put((idx, n, dt, first, None))
    except Exception as exc:       # noqa: BLE001 - reported to the parent, which must not wait for ever
        try:
            barrier.abort()
        except Exception:
            pass
        queue.put((idx, 0, 0.0, [], repr(exc)))


def chains_sharing_one_gpu(seconds=1.0):
    """P independent chains - processes, one point per ``calculate`` call - sharing ONE GPU: `direct`, each with a context of
    its own (a GPU box admits a handful of GPU processes: P <= 4 here), and `brokered`, all of them attached to one owner
    process that batches whatever is pending (victor_amd/broker.py; the chains never touch the GPU).  Aggregate evaluations
    per second and microseconds per call per P.  Must run before this process initialises the GPU: it starts processes."""
    import multiprocessing as mp
    import subprocess
    ctx = mp.get_context("spawn")
    cores = host_cores()

    def run(P, broker):
        barrier = ctx.Barrier(P)
        queue = ctx.Queue()
        procs = [ctx.Process(target=chain_worker, args=(i, P, seconds, broker, barrier, queue)) for i in range(P)]
        for p in procs:
            p.start()
        res = []
        try:
            for _ in procs:
                res.append(queue.get(timeout=900))
        finally:
            for p in procs:
                p.join(timeout=30)
                if p.is_alive():
                    p.kill()
        errs = [r[4] for r in res if r[4]]
        if errs:
            return {"error": errs[0]}
        calls = sum(r[1] for r in res)
        wall = max(r[2] for r in res)
        first = sorted(res)[0][3]
        return {"chains": P, "evals_per_s": calls / wall, "us_per_call": 1e6 * wall * P / calls, "first_logp": first}

    out = {"config": "config/boss_cobaya_config.yaml (BOSS DR12 CMASS), CCFLikelihood.calculate(state, **one_point) in a loop",
           "seconds_per_run": seconds, "host_cores": cores, "direct": {}, "brokered": {}}
    for P in (1, 2, 4):
        out["direct"][str(P)] = run(P, None)
    name = f"victor_bench_{os.getpid()}"
    log = os.path.join(ROOT, "gpurun_out", "broker_bench.log") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else os.devnull
    env = dict(os.environ, PYTHONPATH=ROOT + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else ""))
    env.pop("VICTOR_HIP_BROKER", None)
    with open(log, "ab") as lf:
        srv = subprocess.Popen([sys.executable, "-m", "victor_amd.broker", "--config", "config/boss_cobaya_config.yaml", "--name", name,
                                "--slots", "32"], cwd=ROOT, env=env, stdin=subprocess.DEVNULL, stdout=lf, stderr=lf)
        try:
            for P in (1, 2, 4, 8, 16):
                r = run(P, name)
                if P >= cores:
                    r["note"] = f"{P} spinning chains + the broker on {cores} cores: oversubscribed"
                out["brokered"][str(P)] = r
                if "error" in r:
                    break
            from victor_amd import broker as B
            try:
                seg = B._Segment(B.shm_path(name))
                st = seg.header.stats
                out["broker_stats"] = {"batches": int(st.batches), "evals": int(st.evals), "max_batch": int(st.max_batch),
                                       "mean_batch": st.evals / max(st.batches, 1), "windows_timed_out": int(st.windows_timed_out),
                                       "busy_seconds": st.busy_seconds, "gather_window_us": seg.header.gather_window_us}
                seg.header.stop = 1
                seg.close()
            except Exception as exc:       # noqa: BLE001
                out["broker_stats"] = {"error": repr(exc)}
        finally:
            try:
                srv.wait(timeout=20)
            except subprocess.TimeoutExpired:
                srv.kill()
    d1 = out["direct"].get("1", {}).get("first_logp")
    b = [v.get("first_logp") for v in out["brokered"].values() if "first_logp" in v]
    out["brokered_logp_identical_to_direct"] = bool(d1) and all(x == d1 for x in b) if b else None
    return out


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


def warm_up(eng, launch, seconds=0.3):
    """Run ``launch`` back to back for ``seconds``: the HIP runtime stalls once (~75 ms, kernels unaffected) some tens
    of milliseconds after fresh allocations, which must not land in a short timed window."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            launch()
        eng.sync()


def profiled_traffic(which, batch):
    """HBM bytes per launch of the dominant kernel from the last rocprofv3 --pmc passes (separate FETCH_SIZE / WRITE_SIZE runs,
    tools/update_traffic.py -> profiles/traffic_latest.json): this run makes no counter passes, so the figure is quoted with
    its source and the commit it was taken at, scaled from the profiled batch to this run's (traffic is per point)."""
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.isfile(tfile):
        return None
    with open(tfile) as fh:
        tj = json.load(fh)
    rec = tj.get(which)
    if not rec:
        return None
    scale = batch / rec["batch"]
    from victor_amd.build import sources_digest
    stored = tj.get("sources_sha256")
    return {"bytes_per_launch": rec["theory_kernel_hbm_bytes_per_launch"] * scale, "batch": batch,
            "profiled_batch": rec["batch"], "algorithmic_bytes_per_launch": rec["algorithmic_bytes_per_launch"] * scale,
            "ratio_to_algorithmic": rec["theory_kernel_hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"],
            "kernel": rec["kernel"], "commit": tj.get("commit"), "source": tj.get("source"),
            # the kernel sources (csrc/*.h, csrc/*.hip, include/victor_hip.h) hash to what the counters were taken at
            "sources_unchanged": (stored == sources_digest()) if stored else None}


def boss_measurement(args, batch=16384, steps=20, clocks=None):
    """Secondary figure: the BOSS DR12 CMASS configuration the north star's 1e5 evals/s target is quoted on
    (config/boss_config.yaml: 30 s bins x 100 mu x 50 v, l = 0,2, reconstruction-beta dependent tables, data and
    covariance, Sellentin-Heavens likelihood).  Inputs resident in HBM; same timing discipline as the main line."""
    import victor_amd
    import workloads as cases
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
    fused = eng.last_fused()
    import numpy as np
    lnl = eng.download(d_lnl, batch)
    for p in (d_rows, d_lnl, d_chi, d_ws):
        eng.free(p)
    F = flops_per_eval(30, 100, 50, 2, False)
    k1 /= max(launches, 1)
    frac = F * batch / (k1 * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS if k1 > 0 else None
    return {"evals_per_s": batch * steps / dt, "batch": batch, "steps": steps, "kernel": eng.last_kernel() + "<1,2>",
            "kernels_ms": {"theory": k1, "likelihood": k2 / max(launches, 1)},
            "fp64_valu_frac": frac, **clock_fields(F * batch, clocks, "boss_cmass"),
            "flops_per_eval": F, "all_finite": bool(np.all(np.isfinite(lnl))), "fused": fused,
            "traffic_profiled": profiled_traffic("boss_cmass", batch)}


def at_sustained_clock(frac, clock_ghz):
    """The same fraction against the peak at the clock the kernel actually sustained (peak x clock / 2.4 GHz)."""
    return frac * PEAK_CLOCK_GHZ / clock_ghz if (frac and clock_ghz) else None


def clock_fields(flops_per_launch, clocks, label):
    """`sustained_clock_ghz`, `frac_at_sustained_clock` and `clock_source` of one workload from THIS run's clock pass
    (bench_pmc.live_clocks: {label: {...}} or None).  The fraction is taken in the cycle domain, on the pass's own dispatches:
    algorithmic flops of a launch / the shader cycles that launch took (GRBM_GUI_ACTIVE / 8) / the peak's flops per cycle
    (78.6 TFLOP/s / 2.4 GHz) - the same as `frac` x 2.4 GHz / clock when both come from the same dispatches, which is the point:
    a kernel time from one run is never paired with a clock from another.  Without a pass, or without this label in it, the
    fields are null."""
    rec = (clocks or {}).get(label)
    if not rec or not flops_per_launch:
        return {"sustained_clock_ghz": None, "frac_at_sustained_clock": None, "clock_source": None}
    per_cycle = PEAK_FP64_VALU_TFLOPS * 1e12 / (PEAK_CLOCK_GHZ * 1e9)
    return {"sustained_clock_ghz": rec["sustained_clock_ghz"],
            "frac_at_sustained_clock": flops_per_launch / rec["cycles_per_dispatch"] / per_cycle,
            "clock_source": "this run", "clock_dispatch_ms": rec["dispatch_ms"]}


def batch_sweep():
    """evals/s with inputs resident in HBM at the batch sizes SURVEY.md 8(d) asks for: BASELINE config [1]
    (batch 1024, isotropic xi_r, l = 0,2) and the metric grid (config 3) at batch 1, 64 and 1024."""
    import victor_amd
    import workloads as cases
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
    import numpy as np
    import victor_amd
    import workloads as cases
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
        # CCFModel.theory_xi (ccf_model.py:538-563), the public face of the integrand: xi^s on the fit's own (s, 100 mu) grid for
        # one point and for 1024 (host buffers in, [n][100][n_s] out), and the kernel that served it
        mu = np.linspace(0, 1, 100)
        for _ in range(20):
            fit.theory_xi(fit.s, mu, p)
        t0 = time.perf_counter()
        for _ in range(200):
            fit.theory_xi(fit.s, mu, p)
        xi_one = (time.perf_counter() - t0) / 200
        sub = {k: v[:1024] for k, v in hp.items()}
        fit.theory_xi_batch(fit.s, mu, sub)
        t0 = time.perf_counter()
        for _ in range(5):
            fit.theory_xi_batch(fit.s, mu, sub)
        xi_1024 = (time.perf_counter() - t0) / 5
        res[name] = {"log_likelihood_single_point_us": single * 1e6,
                     "log_likelihood_batch_65536_host_buffers_ms": full * 1e3,
                     "host_buffer_evals_per_s": BATCH_PER_GPU / full,
                     "theory_xi_single_point_us": xi_one * 1e6, "theory_xi_1024_points_ms": xi_1024 * 1e3,
                     "theory_xi_kernel": fit._get_engine().last_kernel()}
    return res


def walker_rates(steps=320):
    """BASELINE config 4's workload on one GPU: lock-step Metropolis walkers on config/boss_cobaya_config.yaml (priors, start
    distributions and proposal widths from the file cobaya reads), proposals made on the host, one host-buffer likelihood
    batch per step.  Likelihood evaluations per second of wall-clock for 8, 64 and 512 walkers; the first evaluation of the
    process (runtime, code object, tables) is timed apart."""
    import victor_amd
    from victor_amd.sampler import EnsembleMetropolis, parse_cobaya_params
    import workloads as cases
    info = cases.cobaya_info()
    lk = info["likelihood"]["CCFLikelihood"]
    cwd = os.getcwd()
    os.chdir(ROOT)                       # the data paths in the config are relative to the repository root
    try:
        fit = victor_amd.CCFFit(lk["model"], lk["data"])
    finally:
        os.chdir(cwd)
    specs, fixed = parse_cobaya_params(info["params"])
    res = {}
    for walkers in (8, 64, 512):
        entry = {}
        for native in (True, False):         # the step loop inside the library (vk_walk_run: the default), and the Python loop beside it
            ens = EnsembleMetropolis(None, specs, walkers, seed=1, fixed=fixed, fit=fit, native=native)   # rows straight to the engine
            ens.initialise()
            t_end = time.perf_counter() + 0.4    # past the first call's one-off costs and the runtime's stall after fresh allocations (see warm_up)
            while time.perf_counter() < t_end:
                ens.run(10)
            dt = n_ev = None
            for _ in range(3):               # the shortest of three windows (a window is 5-50 ms: one hiccup of the host would own it)
                e_start = ens.n_evals
                t0 = time.perf_counter()
                ens.run(steps)
                t = time.perf_counter() - t0
                if dt is None or t < dt:
                    dt, n_ev = t, ens.n_evals - e_start
            if native:
                entry = {"evals_per_s": n_ev / dt, "us_per_step": 1e6 * dt / steps, "windows": "shortest of 3",
                         "acceptance": ens.n_accept / max(ens.n_steps * walkers, 1),
                         "step_loop": "library (vk_walk_run), " + ("two steps per launch" if ens.speculate else "one step per launch")}
            else:
                entry["python_loop"] = {"evals_per_s": n_ev / dt, "us_per_step": 1e6 * dt / steps}
        res[f"{walkers}_walkers"] = entry
    return res


MODEL_OPTIONS = (("dispersion", {"rsd_model": "dispersion"}), ("kaiser", {"rsd_model": "kaiser"}),
                 ("euclid_special", {"rsd_model": "euclid_special"}), ("empirical_corr", {"empirical_corr": True}),
                 ("linear_bias", {"matter_model": "linear_bias"}),
                 ("linear_bias+empirical_corr+dispersion", {"matter_model": "linear_bias", "empirical_corr": True, "rsd_model": "dispersion"}))
FROM_DATA_LABEL = "from_data (measured model, M+D covariance)"


def from_data_options():
    import workloads as cases
    m, d = cases.boss_options("config")
    m["input_model_data_file"] = "boss/measured_model.npy"
    m["realspace_ccf"]["from_data"] = True
    d["covariance_matrix"]["data_file"] = "boss/cov_md_iso.npy"
    return m, d


def option_rates(batch=16384, steps=4, clocks=None):
    """SURVEY.md 8(f) rows next to the headline: the other RSD models and model options of the reference on the BOSS CMASS
    configuration (and the measured real-space ccf with the model + data covariance), resident, batch 16384 - evals/s and the
    theory kernel that served them."""
    import victor_amd
    import workloads as cases
    res = {}

    def run(fit, label, **kw):
        model = fit._merged(kw)
        eng = fit._get_engine(fit._engine_key(model))
        o = eng.make_opts(model, fit.fit_options)
        rows = fit._fit_rows(cases.halton_params(batch, with_beta=True), model)
        bufs = [eng.alloc(rows.size), eng.alloc(batch), eng.alloc(batch), eng.alloc(batch * eng.n_data)]
        eng.upload(bufs[0], rows)
        warm_up(eng, lambda: eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3]), 0.15)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        dt = (time.perf_counter() - t0) / steps
        # the theory kernel's own duration (HIP events on the context's stream) for the roofline fraction beside the rate
        eng.timing(True)
        eng.read_timing(reset=True)
        for _ in range(steps):
            eng.eval_device_async(o, bufs[0], batch, bufs[1], bufs[2], bufs[3])
        eng.sync()
        k1, k2, launches = eng.read_timing(reset=True)
        eng.timing(False)
        k1 /= max(launches, 1)
        fused = eng.last_fused()
        rsd = model["rsd_model"]
        F = model_flops(rsd, len(fit.s), 100, 50, len(fit.poles_s), aniso=not model["assume_isotropic"], niter=int(model.get("niter", 5)),
                        n_data=eng.n_data if fused else 0, linearised=bool(model.get("kaiser_approximation", False)),
                        coord_shift=bool(model.get("kaiser_coord_shift", True)))
        frac = F * batch / (k1 * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS if k1 > 0 else None
        res[label] = {"evals_per_s": batch / dt, "ms_per_batch": dt * 1e3, "kernel": eng.last_kernel(), "fused": fused,
                      "theory_kernel_ms": k1, "flops_per_eval": F, "fp64_valu_frac": frac, **clock_fields(F * batch, clocks, label)}
        for b in bufs:
            eng.free(b)

    boss = victor_amd.CCFFit(*cases.boss_options("config"))
    for label, kw in MODEL_OPTIONS:
        run(boss, label, **kw)
    run(victor_amd.CCFFit(*from_data_options()), FROM_DATA_LABEL)
    return {"batch": batch, "config": "BOSS DR12 CMASS", "rates": res}


def dsplit_measurement(batch=16384, steps=10):
    """BASELINE config 5 on one GPU: five table sets sharing one parameter batch (block-diagonal covariance, N = 5 x 120),
    one upload, the blocks on their own streams, sums on the device (vk_joint_eval_device_async)."""
    import numpy as np
    import victor_amd
    from victor_amd.joint import JointFit
    import workloads as cases
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


class Slot:
    """One GPU of the run: a context with its shard of the global Halton sequence resident in HBM."""

    def __init__(self, g, device, model, data, hp_all, B, total, gathered):
        import victor_amd
        self.g = g
        self.fit = victor_amd.CCFFit(model, data, device=device)
        self.eng = self.fit._get_engine()
        self.opts = self.eng.make_opts(self.fit.model, self.fit.fit_options)
        self.mine = {k: v[g * B:(g + 1) * B] for k, v in hp_all.items()}
        rows = self.fit._fit_rows(self.mine, self.fit.model)
        eng = self.eng
        self.d_rows, self.d_lnl, self.d_chi = eng.alloc(rows.size), eng.alloc(B), eng.alloc(B)
        self.d_ws = eng.alloc(B * eng.n_data)
        self.d_all = eng.alloc(B * total) if gathered else None        # every GPU's copy of the gathered lnL vector
        eng.upload(self.d_rows, rows)

    def launch(self, n):
        self.eng.eval_device_async(self.opts, self.d_rows, n, self.d_lnl, self.d_chi, self.d_ws)

    def free(self):
        for p in (self.d_rows, self.d_lnl, self.d_chi, self.d_ws, self.d_all):
            if p:
                self.eng.free(p)


def make_gather(dist, engines, launched, d_send, d_recv, n):
    """Build the all-gather for the engines this process drives (victor_amd/sharding.py: DeviceGather) and run the first
    collective - it builds RCCL's rings and logs - with stdout routed to stderr.  A rendezvous that never completes is fatal."""
    from victor_amd import _native
    from victor_amd.sharding import DeviceGather
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # RCCL logs go to stdout by default; stdout is the JSON line's
    with stdout_to_stderr():
        try:
            g = DeviceGather(dist, engines, launched, log=lambda msg: print(msg, file=sys.stderr))
        except _native.CommInitTimeout as exc:   # a thread is stuck inside RCCL on this context: fatal, no fallback
            print(f"rank {dist.rank}: {exc}", file=sys.stderr)
            sys.stderr.flush()
            os._exit(4)
        if g.mode in ("rank", "group"):
            failed = False
            try:
                g(d_send, d_recv, n)
                for e in engines:
                    e.sync()
            except Exception as exc:       # noqa: BLE001
                print(f"rank {dist.rank}: first RCCL all-gather failed ({exc})", file=sys.stderr)
                failed = True
            g.degrade(failed)
    return g


def timed_steps(dist, engines, step, steps):
    """``steps`` calls of ``step()`` bracketed by a synchronisation of every engine and a barrier on both sides; the MAX over
    the ranks of the elapsed seconds."""
    for e in engines:
        e.sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for e in engines:
        e.sync()
    dist.barrier()
    return dist.max_float(time.perf_counter() - t0)


def dsplit_sharded(dist, launched, n_local, n_dev, total, steps, warmup, global_batch=16384):
    """BASELINE config 5 on N GPUs: the density-split joint fit (five table sets sharing one parameter batch, block-diagonal
    covariance, N = 5 x 120) with its GLOBAL batch of 16384 Halton points sharded over the GPUs - rank g evaluates rows
    [g B/N, (g+1) B/N) on its five contexts (vk_joint_eval_device_async: the blocks on their own streams, sums on the device) -
    and one all-gather of the joint lnL per step on the lead context's stream, no host synchronisation in between."""
    import numpy as np
    import workloads as cases
    import victor_amd
    from victor_amd.joint import JointFit
    rank = dist.rank
    Bs = max(1, global_batch // total)
    hp_all = cases.halton_params(Bs * total)
    js = []
    for i in range(n_local):
        g = rank * n_local + i
        device = (dist.local_rank if launched else i) % n_dev
        joint = JointFit([victor_amd.CCFFit(*cases.dsplit_options(q), device=device) for q in range(5)])
        engines, opts = joint._plan({})
        mine = {k: v[g * Bs:(g + 1) * Bs] for k, v in hp_all.items()}
        _, (d_rows, d_out, d_ws) = joint._device_buffers(engines, Bs)
        lead = engines[0]
        lead.upload(d_rows, joint.fits[0]._fit_rows(mine, joint.fits[0].model))
        js.append({"g": g, "joint": joint, "engines": engines, "opts": opts, "lead": lead, "d_rows": d_rows, "d_out": d_out,
                   "d_chi": d_out + 8 * Bs, "d_ws": d_ws, "d_all": lead.alloc(Bs * total)})
    leads = [j["lead"] for j in js]
    gat = make_gather(dist, leads, launched, [j["d_out"] for j in js], [j["d_all"] for j in js], Bs)

    def step():
        for j in js:
            j["joint"].eval_device_async(j["engines"], j["opts"], j["d_rows"], Bs, j["d_out"], j["d_chi"], j["d_ws"])
        gat([j["d_out"] for j in js], [j["d_all"] for j in js], Bs)

    warm_up(leads[0], step, 0.2)
    for _ in range(warmup):
        step()
    el = timed_steps(dist, leads, step, steps)
    kernel = leads[0].last_kernel()
    # every GPU checks the WHOLE gathered vector: its own shard bit for bit, rows of every other shard against its own evaluation
    good = True
    for j in js:
        mine_l = j["lead"].download(j["d_out"], Bs)
        gathered = j["lead"].download(j["d_all"], Bs * total)
        good = good and bool(np.array_equal(gathered[j["g"] * Bs:(j["g"] + 1) * Bs], mine_l)) and bool(np.all(np.isfinite(gathered)))
        probe = np.unique(np.linspace(0, Bs - 1, 4).astype(int))
        for other in range(total):
            if other == j["g"]:
                continue
            theirs = {k: v[other * Bs + probe] for k, v in hp_all.items()}
            own_l, _ = j["joint"].log_likelihood_batch(theirs)
            good = good and bool(np.max(np.abs(gathered[other * Bs + probe] - own_l)) <= 1e-9 * np.max(np.abs(own_l)))
    good = bool(dist.min_float(1.0 if good else 0.0))
    gat.close()
    for j in js:
        j["lead"].free(j["d_all"])
    return {"workload": "BASELINE config 5: density-split joint fit, 5 table sets x N = 120, block-diagonal covariance",
            "global_batch": Bs * total, "batch_per_gpu": Bs, "blocks": 5, "steps": steps,
            "joint_evals_per_s": Bs * total * steps / el, "block_evals_per_s": 5 * Bs * total * steps / el,
            "ms_per_step": 1e3 * el / steps, "scaling": "strong", "kernel": kernel, "gather": gat.NAMES[gat.mode],
            "collectives_per_step": 0 if gat.mode == "none" else 1, "gather_matches_local": good}


def walkers_distributed(dist, launched, n_local, n_dev, total, walkers=8, steps=640):
    """BASELINE config 4 on N GPUs: 8 Metropolis walkers per GPU on config/boss_cobaya_config.yaml.  One process per GPU (the
    driver's layout; the reference's own scale-out is N chains under mpirun, README.md:30): every rank advances its own walkers
    on its own GPU and the ranks exchange the log-likelihoods of a 64-step block in ONE RCCL all-gather, enqueued on a context of
    its own behind the block and collected one block later (victor_amd/sampler.py: DistributedEnsemble) - timed with that
    gather and, beside it, without any.  One process driving all
    GPUs: a single ensemble of 8 N walkers whose proposals are sharded over the devices, gathered on the GPUs every step."""
    import numpy as np
    import workloads as cases
    import victor_amd
    from victor_amd import _native
    from victor_amd.sampler import DistributedEnsemble, EnsembleMetropolis, parse_cobaya_params
    from victor_amd.sharding import MultiGPUFit, RcclGather
    info = cases.cobaya_info()
    lk = info["likelihood"]["CCFLikelihood"]
    specs, fixed = parse_cobaya_params(info["params"])
    cwd = os.getcwd()
    os.chdir(ROOT)                       # the data paths in the config are relative to the repository root
    try:
        if launched:
            fit = victor_amd.CCFFit(lk["model"], lk["data"], device=dist.local_rank % n_dev)
        else:
            multi = MultiGPUFit(lk["model"], lk["data"], devices=[i % n_dev for i in range(n_local)])
    finally:
        os.chdir(cwd)
    out = {"workload": "BASELINE config 4: Metropolis walkers on config/boss_cobaya_config.yaml, host proposals",
           "walkers_per_gpu": walkers, "walkers_total": walkers * total, "steps": steps}
    if not launched:
        with stdout_to_stderr():
            rccl = multi.enable_rccl()
        ens = EnsembleMetropolis(multi.log_likelihood_gathered, specs, walkers * total, seed=1, fixed=fixed)
        ens.initialise()
        # what the devices gathered among themselves is what the host gets by concatenating the shards: bit for bit
        probe = ens._batch(ens.x)
        good = bool(np.array_equal(multi.log_likelihood_gathered(probe), multi.log_likelihood_batch(probe)[0]))
        t_end = time.perf_counter() + 0.4
        while time.perf_counter() < t_end:
            ens.run(10)
        e0 = ens.n_evals
        t0 = time.perf_counter()
        ens.run(steps)
        dt = time.perf_counter() - t0
        multi.close()
        out.update({"layout": "one process, ONE ensemble whose proposals are sharded over the devices: the host needs every "
                              "log-likelihood every step, so the (grouped) all-gather is part of the step; a step of 8 walkers per "
                              "GPU is host-bound (22 of 26 us are NumPy) and one interpreter cannot scale it - one process per GPU "
                              "(the launched layout, independent walkers, block gather) is the layout for this workload",
                    "gather": "rccl (grouped)" if rccl else "host", "evals_per_s": (ens.n_evals - e0) / dt,
                    "us_per_step": 1e6 * dt / steps, "acceptance": ens.acceptance, "gather_matches_local": good})
        return out
    engine = fit._get_engine()
    block = EnsembleMetropolis.BLOCK
    gather, ok = None, 1.0
    with stdout_to_stderr():
        try:
            gather = RcclGather.own_context(fit, dist, walkers * block)      # a context (stream) of its own
        except _native.CommInitTimeout as exc:
            print(f"rank {dist.rank}: {exc}", file=sys.stderr)
            sys.stderr.flush()
            os._exit(4)
        except Exception as exc:       # noqa: BLE001 - every rank must take the same branch
            print(f"rank {dist.rank}: RCCL gather unavailable ({exc}); using the socket group", file=sys.stderr)
            ok = 0.0
        if dist.min_float(ok) == 0.0:
            if gather is not None:
                gather.close()
            gather = None

    def evaluate(batch):
        return fit.log_likelihood_batch(batch)[0]

    # the check first, on fresh chains: this rank's slice of the gathered history is its own history bit for bit, and the next
    # rank's slice is the chain this rank gets when it runs that rank's walkers (same seed) itself
    chk = DistributedEnsemble(evaluate, specs, walkers, dist, seed=1, fixed=fixed, gather=gather, fit=fit, gather_block=block)
    n_chk = 2 * block
    _, lnl_own, all_chk = chk.run(n_chk)
    good = bool(np.array_equal(all_chk[:, dist.rank * walkers:(dist.rank + 1) * walkers], lnl_own)) and chk.n_collectives == 2
    other = (dist.rank + 1) % dist.world
    twin = EnsembleMetropolis(evaluate, specs, walkers, seed=1 + 7919 * other, fixed=fixed, fit=fit)
    _, lnl_twin = twin.run(n_chk)
    theirs = all_chk[:, other * walkers:(other + 1) * walkers]
    good = good and bool(np.all(np.abs(theirs - lnl_twin) <= 1e-9 * np.abs(lnl_twin)))
    good = bool(dist.min_float(1.0 if good else 0.0))
    # timing: the same ensemble, first without any gather, then with the block gather
    ens = chk
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        ens.local.run(10)
    dist.barrier()
    t0 = time.perf_counter()
    ens.local.run(steps)
    dt_plain = dist.max_float(time.perf_counter() - t0)
    c0, e0 = ens.n_collectives, ens.local.n_evals
    dist.barrier()
    t0 = time.perf_counter()
    ens.run(steps)
    dist.barrier()
    dt = dist.max_float(time.perf_counter() - t0)
    evals = dist.allgather_host(np.array([float(ens.local.n_evals - e0)]), 1).sum()
    out.update({"layout": "one process per GPU, independent walkers, block gather of lnL",
                "gather": "rccl" if gather is not None else "host (socket group)", "gather_block": block,
                "collectives": ens.n_collectives - c0, "collectives_per_step": (ens.n_collectives - c0) / steps,
                "evals_per_s": float(evals) / dt, "us_per_step": 1e6 * dt / steps,
                "us_per_step_without_gather": 1e6 * dt_plain / steps, "gather_cost_ratio": dt / dt_plain,
                "gather_us_per_collective": 1e6 * (dt - dt_plain) / max(ens.n_collectives - c0, 1),
                "gather_overlapped": bool(ens.overlap),
                "acceptance": ens.local.acceptance, "gather_matches_local": good})
    if gather is not None:
        gather.close()
    return out
