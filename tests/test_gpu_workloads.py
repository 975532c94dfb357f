"""BASELINE.json configs [3] and [4] as parity cases on the GPU (the bench line is config [2])."""

import os
import subprocess
import sys

import numpy as np
import pytest

from tests import cases
from tests.tolerances import assert_same_chi2, assert_same_lnl, chi2_bound
from victor_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def test_walker_ensemble_matches_oracle_driven_chain():
    """Config [3]: 8 Metropolis walkers on the BOSS cobaya configuration.  The same seeded chain is driven once
    by the HIP likelihood and once by the CPU oracle: every accept/reject decision and position must coincide."""
    import victor_amd
    import victor_oracle as vo
    from victor_amd.sampler import EnsembleMetropolis, parse_cobaya_params
    info = cases.cobaya_info()
    specs, fixed = parse_cobaya_params(info["params"])
    lk = info["likelihood"]["CCFLikelihood"]
    fit = victor_amd.CCFFit(lk["model"], lk["data"])
    ora = vo.OracleFit(*cases.boss_options("cobaya"))

    def gpu_eval(batch):
        return fit.log_likelihood_batch(batch)[0]

    def cpu_eval(batch):
        n = len(batch["fsigma8"])
        return np.array([ora.log_likelihood(cases.point(batch, i))[0] for i in range(n)])

    g = EnsembleMetropolis(gpu_eval, specs, 8, seed=2024, fixed=fixed).initialise()
    c = EnsembleMetropolis(cpu_eval, specs, 8, seed=2024, fixed=fixed).initialise()
    cg, lg = g.run(5)
    cc, lc = c.run(5)
    assert np.array_equal(cg, cc)
    assert np.max(np.abs(lg / lc - 1)) < 1e-9
    assert g.n_accept == c.n_accept and g.n_evals == c.n_evals
    # the direct route (what bench.py and run_walkers.py use): rows written in place, the ensemble as two halves on two
    # contexts, run() pipelined over the steps.  Same random numbers, same rows: the same chain - positions and decisions
    # identical, log-likelihoods to rounding (half-batches of 4 take another work split than one batch of 8) - over enough steps
    # to cross a block of pre-drawn random numbers and to meet proposals outside the prior; step() and run() evaluate the
    # same half-batches and agree bit for bit.
    d = EnsembleMetropolis(None, specs, 8, seed=2024, fixed=fixed, fit=fit, native=False).initialise()
    assert d._direct is not None and len(d._direct["engines"]) == 2 and not d._walk
    g2 = EnsembleMetropolis(gpu_eval, specs, 8, seed=2024, fixed=fixed).initialise()
    cd, ld = d.run(150)
    cg2, lg2 = g2.run(150)
    assert np.array_equal(cd[:5], cg) and np.array_equal(cd, cg2)
    assert np.max(np.abs(ld - lg2)) < 1e-9 * np.max(np.abs(lg2))
    assert d.n_accept == g2.n_accept and d.n_evals == g2.n_evals and 0 < d.n_accept < 150 * 8 and d.n_steps == 150
    s1 = EnsembleMetropolis(None, specs, 8, seed=2024, fixed=fixed, fit=fit, native=False).initialise()
    for t in range(70):
        s1.step()
        assert np.array_equal(s1.x, cd[t]) and np.array_equal(s1.lnl, ld[t]), t
    # The same loop inside the library (vk_walk_run; the default with fit=): the same half-batches, the same rows, the same
    # launches - positions, decisions, counters AND log-likelihoods identical to the Python loop, bit for bit (every route forms
    # apar = eps^(-2/3) with the library's one routine since round 6) -, run in pieces that do not line up with the blocks
    # of random numbers, with the per-step callback seeing the state after every step.
    n = EnsembleMetropolis(None, specs, 8, seed=2024, fixed=fixed, fit=fit, speculate=False).initialise()
    assert n._walk and not n.speculate
    seen = []
    parts = [n.run(37, on_step=lambda t, ens: seen.append((ens.x.copy(), ens.lnl.copy()))), n.run(100), n.run(13)]
    cn = np.concatenate([p[0] for p in parts])
    ln = np.concatenate([p[1] for p in parts])
    assert np.array_equal(cn, cd) and np.array_equal(ln, ld)
    assert (n.n_accept, n.n_evals, n.n_steps) == (d.n_accept, d.n_evals, d.n_steps)
    assert len(seen) == 37 and all(np.array_equal(sx, cd[t]) and np.array_equal(sl, ln[t]) for t, (sx, sl) in enumerate(seen))
    assert np.array_equal(n.x, cd[-1])
    # Two steps per launch (the default for small ensembles): per walker the proposal of step t and both candidates of step
    # t + 1 in one launch.  Other launches (three rows per walker: other work splits), so the log-likelihoods differ in their
    # last bits - the positions, the decisions and the counters are those of the step-by-step loop, in pieces of odd and even
    # length, across the blocks of random numbers.
    sp = EnsembleMetropolis(None, specs, 8, seed=2024, fixed=fixed, fit=fit).initialise()
    assert sp._walk and sp.speculate
    parts = [sp.run(37), sp.run(100), sp.run(1), sp.run(12)]
    cs = np.concatenate([p[0] for p in parts])
    ls = np.concatenate([p[1] for p in parts])
    assert np.array_equal(cs, cd) and np.max(np.abs(ls - ld)) <= 1e-9 * np.max(np.abs(ld))
    assert (sp.n_accept, sp.n_evals, sp.n_steps) == (d.n_accept, d.n_evals, d.n_steps)
    # ... and however the run is cut - one piece, pieces of odd length, single steps - the history is the same bit for bit: a
    # left-over single step travels in a launch of the same shape as the two-step launches (its candidate rows idle)
    for cut in ((150,), (1, 1, 1, 147), (75, 75), (149, 1)):
        sq = EnsembleMetropolis(None, specs, 8, seed=2024, fixed=fixed, fit=fit).initialise()
        parts = [sq.run(k) for k in cut]
        assert np.array_equal(np.concatenate([p[0] for p in parts]), cs), cut
        assert np.array_equal(np.concatenate([p[1] for p in parts]), ls), cut
        assert (sq.n_accept, sq.n_evals, sq.n_steps) == (sp.n_accept, sp.n_evals, sp.n_steps)
    # an odd number of walkers (halves of 3 and 4), a single walker, and an ensemble whose halves take the cells kernel
    for w in (7, 1, 64):
        a = EnsembleMetropolis(None, specs, w, seed=5, fixed=fixed, fit=fit).initialise()
        p = EnsembleMetropolis(None, specs, w, seed=5, fixed=fixed, fit=fit, native=False).initialise()
        b = EnsembleMetropolis(gpu_eval, specs, w, seed=5, fixed=fixed).initialise()
        ca, la = a.run(30)
        cp, lp = p.run(30)
        cb, lb = b.run(30)
        assert a._walk and not p._walk
        assert np.array_equal(ca, cb) and np.max(np.abs(la - lb)) < 1e-9 * np.max(np.abs(lb)), w
        assert np.array_equal(ca, cp) and np.max(np.abs(la - lp)) <= 1e-9 * np.max(np.abs(lp)), w          # (two steps per launch)
        assert (a.n_accept, a.n_evals) == (p.n_accept, p.n_evals) == (b.n_accept, b.n_evals), w


def test_cobaya_plugin_calculate_on_gpu():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "victor", "likelihoods"))
    from CCFLikelihood import CCFLikelihood
    info = cases.cobaya_info()["likelihood"]["CCFLikelihood"]
    lk = CCFLikelihood(dict(model=info["model"], data=info["data"]))
    state = {}
    assert lk.calculate(state, want_derived=True, fsigma8=0.47, beta=0.37, sigma_v=380, epsilon=1.0) is None
    g, _ = cases.golden_outputs()
    assert abs(state["logp"] - g["boss_cobaya_lnl"][0]) < 1e-9 * abs(state["logp"])
    assert abs(state["derived"]["chi2_ccf_correct"] - g["boss_cobaya_chi2"][0]) < 1e-9 * g["boss_cobaya_chi2"][0]
    lnl, chi2 = lk.calculate_batch({"fsigma8": np.array([0.47, 0.5]), "beta": 0.37, "sigma_v": 380, "epsilon": 1.0})
    assert lnl.shape == (2,)
    bound = chi2_bound(lk.ccf, {"fsigma8": np.array([0.47, 0.5]), "beta": 0.37, "sigma_v": 380, "epsilon": 1.0})
    assert_same_lnl(lnl[0], state["logp"], bound[0], what="calculate_batch vs calculate")


def _cobaya_inputs(lk, sampled):
    """What cobaya's model hands to ``calculate``: EVERY input parameter the plug-in's YAML declares - sampled ones, fixed
    numbers, and the dynamically defined ones (``value: "lambda alpha, epsilon: ..."``) evaluated from the others."""
    import inspect
    known, later = dict(sampled), {}
    for name in lk.input_params:
        spec = lk.params[name]
        if name in known:
            continue
        if isinstance(spec, dict) and isinstance(spec.get("value"), str) and spec["value"].lstrip().startswith("lambda"):
            later[name] = eval(spec["value"])            # the plug-in's own YAML, as cobaya evaluates it
        elif isinstance(spec, dict) and "value" in spec:
            known[name] = float(spec["value"])
        elif spec is None:
            raise KeyError(f"sampled parameter {name} needs a value")
        else:
            known[name] = float(spec)
    for name, fn in later.items():
        known[name] = float(fn(*[known[a] for a in inspect.signature(fn).parameters]))
    return known


def test_cobaya_shaped_construction_and_full_parameter_set():
    """The plug-in driven the way cobaya drives it (CCFLikelihood.py:32-42 of the reference, its YAML lines 9-41): constructed
    as ``Class(info, name, timing, packages_path, initialize, standalone)`` with its class defaults coming from
    CCFLikelihood.yaml, and ``calculate(state, want_derived, **params)`` receiving every declared input - the
    lambda-derived aperp / apar / alpha NEXT TO epsilon, bias, the excursion-set and cosmology extras.  logp must be
    bit-identical to the four-parameter call (epsilon takes precedence over aperp / apar, ccf_model.py:589-596; unknown
    names are ignored) and equal to the reference's golden."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "victor", "likelihoods"))
    from CCFLikelihood import CCFLikelihood
    run = cases.cobaya_info()["likelihood"]["CCFLikelihood"]
    lk = CCFLikelihood({"model": run["model"], "data": run["data"]}, "CCFLikelihood", None, None, True, False)
    assert lk.get_name() == "CCFLikelihood" and lk.config_file == "config/boss_config.yaml"      # a class default
    assert "chi2_ccf_correct" in lk.output_params and "chi2_ccf_correct" not in lk.input_params
    declared = {"fsigma8", "beta", "epsilon", "b", "alpha", "aperp", "apar", "astar", "sigma_v", "Av", "f", "sigma_8_0", "b10",
                "b01", "Rp", "Rx", "Omega_m", "Omega_b", "H0", "ns", "mnu", "Omega_k", "delta_c", "M", "Q"}
    assert set(lk.input_params) == declared                       # the reference YAML's list, name for name
    g, meta = cases.golden_outputs()
    for i, pt in enumerate(meta["boss_points"][:2]):              # epsilon = 1.0 and epsilon = 1.04
        sampled = {k: pt[k] for k in ("fsigma8", "beta", "epsilon")}
        full = _cobaya_inputs(lk, dict(sampled, sigma_v=pt["sigma_v"]))
        assert set(full) == declared
        assert abs(full["aperp"] - full["alpha"] * full["epsilon"] ** (1 / 3)) < 1e-15 and "apar" in full
        s_full, s_four = {}, {}
        lk.calculate(s_full, want_derived=True, **full)
        lk.calculate(s_four, want_derived=True, **dict(sampled, sigma_v=pt["sigma_v"]))
        assert s_full["logp"] == s_four["logp"] and s_full["derived"] == s_four["derived"]          # bit for bit
        assert abs(s_full["logp"] - g["boss_cobaya_lnl"][i]) < 1e-9 * abs(s_full["logp"])
        assert abs(s_full["derived"]["chi2_ccf_correct"] - g["boss_cobaya_chi2"][i]) < 1e-9 * g["boss_cobaya_chi2"][i]
    # initialize=False leaves the object unbuilt, as in cobaya
    bare = CCFLikelihood({"model": run["model"], "data": run["data"]}, initialize=False)
    assert not hasattr(bare, "ccf")


def test_config_file_route_of_plugin(tmp_path, monkeypatch):
    """model/data omitted -> the plug-in loads config/boss_config.yaml relative to the working directory."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.chdir(root)
    sys.path.insert(0, os.path.join(root, "victor", "likelihoods"))
    from CCFLikelihood import CCFLikelihood
    lk = CCFLikelihood(dict(model=None, data=None, config_file="config/boss_config.yaml"))
    state = {}
    lk.calculate(state, fsigma8=0.47, beta=0.37, sigma_v=380, epsilon=1.0)
    assert round(state["derived"]["chi2_ccf_correct"], 2) == 65.01 and round(state["logp"], 2) == 284.76


def test_density_split_joint_fit():
    """Config [4]: five stacked data vectors with block-diagonal precision (N = 5 x 120), one parameter vector."""
    import victor_amd
    from victor_amd.joint import JointFit
    joint = JointFit([victor_amd.CCFFit(*cases.dsplit_options(q)) for q in range(5)])
    assert joint.n_data == 600
    g, meta = cases.golden_outputs()
    pts = meta["synth_points"][:6]
    batch = {k: np.array([p[k] for p in pts]) for k in pts[0]}
    lnl, chi2 = joint.log_likelihood_batch(batch)
    assert np.max(np.abs(chi2 / g["dsplit_chi2"] - 1)) < 1e-9
    assert np.max(np.abs(lnl / g["dsplit_lnl"] - 1)) < 1e-9
    # full-size batch of the config: properties that do not need the oracle
    hp = cases.halton_params(16384)
    lnl, chi2 = joint.log_likelihood_batch(hp)
    assert lnl.shape == (16384,) and np.all(np.isfinite(lnl))
    assert np.max(np.abs(lnl + 0.5 * chi2)) < 1e-9 * np.max(chi2)
    # the joint chi2 is the sum over the blocks, so is the bound on what another evaluation order may change (tests/tolerances.py)
    bound = sum(chi2_bound(f, hp) for f in joint.fits)
    parts = sum(f.log_likelihood_batch({k: v[:64] for k, v in hp.items()})[1] for f in joint.fits)
    assert_same_chi2(parts, chi2[:64], bound[:64], what="joint fit: sub-batch per block")      # another work split / summation order
    # the device-resident joint path (one upload, five table sets on their own streams, sums on the device) against one
    # host call per block; and per-block options that differ take the per-block route
    seq_l, seq_c = joint._sequential(hp, {})
    assert_same_chi2(seq_c, chi2, bound, what="joint fit: device path vs one call per block")
    assert_same_lnl(seq_l, lnl, bound, what="joint fit: device path vs one call per block")
    assert joint._plan({}) is not None
    lnl_h, chi_h = joint.log_likelihood_batch({k: v[:300] for k, v in hp.items()}, likelihood={"form": "hartlap", "nmocks": 2000})
    seq_l, seq_c = joint._sequential({k: v[:300] for k, v in hp.items()}, {"likelihood": {"form": "hartlap", "nmocks": 2000}})
    assert_same_lnl(seq_l, lnl_h, bound[:300], what="joint fit: hartlap form")
    bad = {k: v[:5].copy() for k, v in hp.items()}
    bad["sigma_v"][3] = np.nan
    lnl_b, chi_b = joint.log_likelihood_batch(bad)
    assert np.isneginf(lnl_b[3]) and np.isposinf(chi_b[3]) and np.all(np.isfinite(lnl_b[[0, 1, 2, 4]]))
    one = joint.log_likelihood(pts[0])
    assert abs(one[1] / g["dsplit_chi2"][0] - 1) < 1e-9


def _launch_ranks(script_args, world, extra_env=None, timeout=240):
    """Start ``world`` ranks the way any launcher does - RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment - as plain
    child processes (no torchrun, no mpirun); returns their CompletedProcess-like (returncode, stdout, stderr) tuples."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable] + script_args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=timeout)
            out.append((p.returncode, o, e))
    finally:
        for p in procs:                     # a rank that is still there (a hung rendezvous) must not outlive the test
            if p.poll() is None:
                p.kill()
    return out


def test_walker_example_as_one_launched_rank():
    """examples/run_walkers.py as the single rank of a launcher-style start (RANK=0, WORLD_SIZE=1): with one rank no gather is
    built; the chain itself must behave."""
    import json
    (rc, out, err), = _launch_ranks([os.path.join(ROOT, "examples", "run_walkers.py"), "--steps", "40"], 1)
    assert rc == 0, err[-2000:]
    res = json.loads(out.strip().splitlines()[-1])
    assert res["walkers_total"] == 8 and res["gathered_shape"] == [40, 8]
    assert 0.02 < res["acceptance"] < 0.95
    assert 0.2 <= res["mean"]["beta"] <= 0.6 and 100 <= res["mean"]["sigma_v"] <= 500
    assert res["best_lnl_over_all_ranks"] > 250          # the reference point alone gives lnL = 284.8


def test_walker_example_two_ranks_and_two_contexts_on_one_gpu():
    """The same example with two launched ranks (socket rendezvous, RCCL refused for two ranks on one device -> gather through
    the socket group) and as ONE process with two contexts (``--gpus 2``: ncclCommInitAll refused for a shared device -> host
    concatenation): both layouts end to end on the one-GPU box."""
    import json
    res = _launch_ranks([os.path.join(ROOT, "examples", "run_walkers.py"), "--steps", "20"], 2)
    for rc, out, err in res:
        assert rc == 0, err[-2000:]
    r0 = json.loads(res[0][1].strip().splitlines()[-1])
    assert r0["walkers_total"] == 16 and r0["gathered_shape"] == [20, 16] and r0["gather"] in ("host", "rccl")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "run_walkers.py"), "--steps", "20", "--gpus", "2"],
                         capture_output=True, text=True, timeout=240)
    assert one.returncode == 0, one.stderr[-2000:]
    r1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert r1["walkers_total"] == 16 and r1["gathered_shape"] == [20, 16] and r1["best_lnl_over_all_ranks"] > 200


def _check_two_gpu_line(out):
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8192 and out["scaling"] == "weak"
    assert out["gather_matches_local"] is True and out["outputs_finite"] is True
    assert "host allgather" in out["config"]["gather"] or "rccl" in out["config"]["gather"]
    rccl = out["config"]["rccl"]
    assert rccl["hip_runtime"] and rccl["rccl"] and rccl["rccl_version"] > 0 and rccl["rccl_next_to_hip_runtime"] is True
    assert "torch" not in rccl["hip_runtime"]            # /opt/rocm's runtime: nothing imported torch's bundled copy first
    assert len(out["theory_kernel_ms_per_rank"]) == 2
    strong = out["strong_scaling"]
    assert strong["global_batch"] == 4096 and strong["batch_per_gpu"] == 2048 and strong["value"] > 0
    assert len(strong["theory_kernel_ms_per_rank"]) == 2


def test_bench_two_launched_ranks_on_one_gpu():
    """``bench.py --gpus 2`` as two launched ranks on the one-GPU box (what the driver's torchrun start looks like to the
    ranks): standard-library rendezvous, both ranks on device 0, RCCL refuses the communicator and the gather falls back to
    the socket group - which exercises everything around the collective: every rank's check of EVERY slot of the gathered
    vector against its own recomputation of the other rank's rows, the fixed-global-batch leg with per-rank kernel times, and
    the library diagnostics in the JSON line."""
    import json
    res = _launch_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4096",
                         "--no-cpu-baseline", "--no-boss"], 2)
    for rc, out, err in res:
        assert rc == 0, err[-2000:]
    lines = [ln for ln in res[0][1].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in res[1][1].splitlines() if ln.startswith("{")], res[0][1][-2000:]
    out = json.loads(lines[0])
    _check_two_gpu_line(out)
    assert out["config"]["processes"] == 2 and out["config"]["contexts_per_process"] == 1
    assert "socket" in out["config"]["rendezvous"]


def test_bench_under_the_drivers_launcher_command():
    """The driver's own N > 1 command line - ``python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 2 ...`` - on the one-GPU box.  torch is the LAUNCHER only (its agent owns TCP port
    P, which is why the ranks meet on a Unix socket named after it); the ranks must produce the same record as under the plain
    launcher of the test above, and must not have mapped torch's HIP runtime."""
    import json
    import socket
    try:
        import importlib.util
        if importlib.util.find_spec("torch") is None:
            pytest.skip("torch (the driver's launcher) is not installed")
    except ImportError:
        pytest.skip("torch (the driver's launcher) is not installed")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "4096", "--no-cpu-baseline", "--no-boss"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=360)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    _check_two_gpu_line(out)
    assert out["config"]["processes"] == 2 and "socket" in out["config"]["rendezvous"]


def test_bench_single_launched_rank_gathers_through_rccl():
    """One launched rank: the whole RCCL path of the multi-process layout - unique id, ncclCommInitRank on the watchdog thread,
    ncclAllGather on the context's stream in every timed step, every slot checked - with /opt/rocm's RCCL next to /opt/rocm's HIP
    runtime (no torch in the process)."""
    import json
    (rc, out, err), = _launch_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "4096",
                                     "--no-cpu-baseline", "--no-boss"], 1)
    assert rc == 0, err[-2000:]
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["gather_matches_local"] is True
    assert line["config"]["gather"].startswith("rccl allgather of lnL (ncclCommInitRank"), line["config"]["gather"]
    assert "torch" not in line["config"]["rccl"]["hip_runtime"] and line["config"]["rccl"]["rccl_next_to_hip_runtime"] is True
    # what the real RCCL says about the communicator it built: one rank, rank 0, on the device whose bus id the record names
    (rec,) = line["config"]["rccl"]["ranks"]
    assert rec["count"] == 1 and rec["rank"] == 0 and rec["device"] == 0 and len(rec["bus_id"]) >= 7, rec


def test_bench_one_process_two_contexts_on_one_gpu():
    """``python bench.py --gpus 2`` started on its own: ONE process drives both contexts (no launcher, no rendezvous); on the
    one-GPU box ncclCommInitAll is refused for the shared device and the gather is the host concatenation."""
    import json
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "4096", "--no-cpu-baseline", "--no-boss"], capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    _check_two_gpu_line(out)
    assert out["config"]["processes"] == 1 and out["config"]["contexts_per_process"] == 2


def test_bench_one_process_two_contexts_with_the_config_4_and_5_legs():
    """The same start with the BASELINE config 4 / 5 legs: the density-split joint fit sharded over the two contexts (five
    contexts each, global batch 16384) and one walker ensemble sharded over them - on the one-GPU box both gathers degrade to the
    host, and every check of what was gathered must still hold."""
    import json
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "4096", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])
    _check_two_gpu_line(out)
    d5, w = out["dsplit5"], out["walker_ensembles"]
    assert d5["global_batch"] == 16384 and d5["batch_per_gpu"] == 8192 and d5["gather_matches_local"] is True
    assert d5["kernel"] == "vk_theory_cells_kernel"
    assert w["walkers_total"] == 16 and w["gather_matches_local"] is True


def test_integration_stub_runs_as_written():
    """The ctypes stub printed in INTEGRATION.md (route B) is executed verbatim against a CCFFit and must reproduce
    the package's own batch API."""
    import os
    import re
    import numpy as np
    import victor_amd
    from tests import cases
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "class HipLikelihood" in b]
    assert len(stub) == 1
    ns = {}
    exec(compile(stub[0], "INTEGRATION.md", "exec"), ns)
    fit = victor_amd.CCFFit(*cases.boss_options("config"))
    like = ns["HipLikelihood"](fit)
    rows = fit._fit_rows(cases.halton_params(257, with_beta=True), fit.model)
    lnl, chi2 = like(rows)
    want_l, want_c = fit.log_likelihood_batch(rows)
    assert np.array_equal(lnl, want_l) and np.array_equal(chi2, want_c)


def test_two_dimensional_model_grids():
    """theory_xi_2D / xi_2D_from_multipoles (ccf_model.py:862-934): grid nodes against scalar oracle evaluations, and
    the call convention of the returned interpolant."""
    import os
    import sys
    import numpy as np
    import victor_amd
    from tests import cases
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import victor_oracle as vo
    fit = victor_amd.CCFFit(*cases.boss_options("config"))
    ora = vo.OracleFit(*cases.boss_options("config"))
    p = {"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.02}
    f2d = fit.theory_xi_2D(dict(p), rmax=85)
    sperp, spar = np.linspace(0.01, 85), np.linspace(-85, 85)
    assert f2d.z.shape == (50, 50) and np.array_equal(f2d.x, sperp) and np.array_equal(f2d.y, spar)
    for i, j in ((0, 0), (7, 3), (24, 25), (49, 49), (30, 12)):
        s = np.hypot(sperp[i], spar[j])
        want = ora.theory_xi(np.array([s]), np.array([spar[j] / s]), dict(p))[0, 0]
        assert abs(f2d.z[j, i] - want) < 1e-9 * max(abs(want), 1e-3), (i, j)
        assert abs(f2d(sperp[i], spar[j])[0] - f2d.z[j, i]) < 1e-15
    assert f2d(np.array([1.0, 2.0, 3.0]), np.array([-5.0, 5.0])).shape == (2, 3)
    m2d = fit.xi_2D_from_multipoles(dict(p), rmax=85)
    s1 = np.linspace(0.01, 85)
    poles, _ = ora.theory_multipoles(s1, dict(p), poles=[0, 2, 4])
    from victor_amd import tables as T
    i, j = 11, 40
    s = np.hypot(sperp[i], spar[j])
    want = sum(T.notaknot(s1, poles[f"{l}"])(min(s, 85.0)) * T.legendre_values(l, spar[j] / s) for l in (0, 2, 4))
    assert abs(m2d.z[j, i] - want) < 1e-9 * abs(want)


def test_plain_c_client_of_the_abi(tmp_path):
    """examples/c_abi_client.c (no Python, no C++): load a dumped vk_tables, vk_create, vk_eval_batch - must reproduce
    the Python path bit for bit."""
    import os
    import subprocess
    import numpy as np
    import victor_amd
    from tests import cases
    from victor_amd.engine import build_tables, dump_tables
    from victor_amd import _native as N
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_client")
    subprocess.run(["gcc", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_abi_client.c"),
                    "-ldl", "-o", exe], check=True)
    fit = victor_amd.CCFFit(*cases.boss_options("config"))
    tables, keep = build_tables(fit, fit)
    dump_tables(tables, str(tmp_path / "tables.bin"))
    rows = fit._fit_rows(cases.halton_params(300, with_beta=True), fit.model)
    rows.tofile(str(tmp_path / "params.bin"))
    like = fit.fit_options["likelihood"]
    args = [exe, N.library_path(), str(tmp_path / "tables.bin"), str(tmp_path / "params.bin"), str(tmp_path / "out.bin"),
            str(int(not fit.model["velocity_independent_of_AP"])), str(int(fit.model["assume_isotropic"])),
            str(N.LIKE[like["form"].lower()]), str(like.get("nmocks", 1)), str(like.get("nparams", 0))]
    done = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert done.returncode == 0, done.stderr
    out = np.fromfile(str(tmp_path / "out.bin"))
    want_l, want_c = fit.log_likelihood_batch(rows)
    assert np.array_equal(out[:300], want_l) and np.array_equal(out[300:], want_c)


def test_small_batches_replay_a_captured_graph():
    """Host-buffer batches of up to 4096 points are launch-bound: from their second use on, (H2D, theory kernel,
    likelihood kernel, D2H) is one hipGraph launch.  Results must be bit-identical to the eager path, follow the
    parameters of every call, and survive a change of batch size and options."""
    import os
    import numpy as np
    import victor_amd
    from tests import cases
    fit = victor_amd.CCFFit(*cases.boss_options("config"))
    hp = cases.halton_params(64, with_beta=True)
    rows = fit._fit_rows(hp, fit.model)
    _native.set_knob("VICTOR_HIP_NO_ZERO_COPY", "1")     # the in-place path would serve these sizes first: this test is about the graphs
    try:
        _graph_replay_checks(fit, rows, hp)
    finally:
        _native.set_knob("VICTOR_HIP_NO_ZERO_COPY", None)
        _native.set_knob("VICTOR_HIP_NO_GRAPH", None)


def _graph_replay_checks(fit, rows, hp):
    import numpy as np
    from tests import cases
    _native.set_knob("VICTOR_HIP_NO_GRAPH", "1")
    try:
        want = {n: fit.log_likelihood_batch(rows[:n]) for n in (1, 8, 64)}
        want_shift = fit.log_likelihood_batch(rows[8:16])
        want_kaiser = fit.log_likelihood_batch(rows[:8], rsd_model="kaiser")
    finally:
        _native.set_knob("VICTOR_HIP_NO_GRAPH", None)
    for rep in range(4):                       # call 1 eager, call 2 captures, calls 3-4 replay
        for n in (1, 8, 64):
            got = fit.log_likelihood_batch(rows[:n])
            assert np.array_equal(got[0], want[n][0]) and np.array_equal(got[1], want[n][1]), (rep, n)
        got = fit.log_likelihood_batch(rows[8:16])                  # same shape, other parameters
        assert np.array_equal(got[0], want_shift[0]) and np.array_equal(got[1], want_shift[1]), rep
        got = fit.log_likelihood_batch(rows[:8], rsd_model="kaiser")  # same shape, other options -> another graph
        assert np.array_equal(got[0], want_kaiser[0]), rep
    big = fit.log_likelihood_batch(fit._fit_rows(cases.halton_params(5000, with_beta=True), fit.model))   # regrows scratch
    assert np.all(np.isfinite(big[0]))
    got = fit.log_likelihood_batch(rows[:8])
    assert np.array_equal(got[0], want[8][0])
    p = cases.point(hp, 3)
    single = [fit.log_likelihood(dict(p)) for _ in range(5)]
    assert all(s == single[0] for s in single)
    _native.set_knob("VICTOR_HIP_NO_ZERO_COPY", None)
    # default: host-buffer batches of up to 4096 points are read and written in place (pinned, device-mapped memory): same bits
    for n in (1, 8, 64):
        for rep in range(2):
            got = fit.log_likelihood_batch(rows[:n])
            assert np.array_equal(got[0], want[n][0]) and np.array_equal(got[1], want[n][1]), (n, rep)
    assert fit.log_likelihood(dict(p)) == single[0]


def test_in_place_calls_poll_for_their_results():
    """In-place host-buffer calls of up to 256 points do not wait for the end of the launch but for the results themselves
    (slots pre-set to a NaN pattern, polled in pinned memory).  Same bits as the synchronised call for every size, rows
    that fail (NaN parameter -> (-inf, inf)) included, through calls of changing size and content."""
    import numpy as np
    import victor_amd
    from tests import cases
    fit = victor_amd.CCFFit(*cases.boss_options("config"))
    rows = fit._fit_rows(cases.halton_params(300, with_beta=True), fit.model)
    rows[5, _native.P_SIGMAV] = np.nan
    rows[200, _native.P_FSIGMA8] = np.inf
    sizes = (1, 6, 64, 256, 300)
    try:
        _native.set_knob("VICTOR_HIP_SPIN_MAX", "0")
        want = {n: fit.log_likelihood_batch(rows[:n]) for n in sizes}
        want_tail = fit.log_likelihood_batch(rows[190:210])
        _native.set_knob("VICTOR_HIP_SPIN_MAX", None)
        for rep in range(3):
            for n in sizes:
                got = fit.log_likelihood_batch(rows[:n])
                assert np.array_equal(got[0], want[n][0]) and np.array_equal(got[1], want[n][1]), (rep, n)
            got = fit.log_likelihood_batch(rows[190:210])
            assert np.array_equal(got[0], want_tail[0]) and np.array_equal(got[1], want_tail[1]), rep
        assert np.isneginf(want[6][0][5]) and np.isposinf(want[6][1][5]) and np.isneginf(want_tail[0][10])
        p = cases.point(cases.halton_params(4, with_beta=True), 3)
        singles = {fit.log_likelihood(dict(p)) for _ in range(200)}
        assert len(singles) == 1
    finally:
        _native.set_knob("VICTOR_HIP_SPIN_MAX", None)


def test_one_process_driving_several_contexts():
    """victor_amd.sharding.MultiGPUFit: contiguous shards of a batch evaluated concurrently from host threads, one
    context per listed device (the one-GPU box lists device 0 three times); same as the single-context result."""
    import numpy as np
    import victor_amd
    from victor_amd.sharding import MultiGPUFit
    from tests import cases
    opts = cases.boss_options("config")
    multi = MultiGPUFit(*opts, devices=[0, 0, 0])
    single = victor_amd.CCFFit(*opts)
    for n in (1, 2, 1000, 20001):
        hp = cases.halton_params(n, with_beta=True)
        got = multi.log_likelihood_batch(hp)
        want = single.log_likelihood_batch(hp)
        # shards of a different size may take another kernel mapping: agreement to rounding, not bit for bit
        # (lnL = -1/2 log det - n/2 log(1 + chi2/(n-1)) crosses zero: compare it on the scale of its two terms)
        bound = chi2_bound(single, hp)
        assert_same_chi2(got[1], want[1], bound, what=f"three shards on one GPU, n={n}")
        assert_same_lnl(got[0], want[0], bound, what=f"three shards on one GPU, n={n}")
    th = multi.theory_vector_batch(cases.halton_params(77, with_beta=True), rsd_model="dispersion")
    assert th.shape == (77, 60) and np.all(np.isfinite(th))
    multi.close()


def test_grouped_rccl_gather_of_one_process_at_one_context():
    """The RCCL calls of the one-process layout - ncclCommInitAll, ncclGroupStart / ncclAllGather / ncclGroupEnd on the context's
    stream (vk_comm_init_all, vk_comm_allgather_group_async) - with the one context a one-GPU box allows: MultiGPUFit's gathered
    evaluation (shard upload, launch, grouped all-gather, download from device 0) against the plain batch API, for sizes around
    the padding and buffer-growth edges; a second device entry for the same GPU is refused without calling RCCL."""
    import victor_amd
    from victor_amd.sharding import MultiGPUFit
    opts = cases.boss_options("config")
    multi = MultiGPUFit(*opts, devices=[0])
    assert multi.enable_rccl() is True, getattr(multi, "_rccl_error", "")
    single = victor_amd.CCFFit(*opts)
    for n in (1, 63, 64, 65, 3000):
        hp = cases.halton_params(n, with_beta=True)
        got = multi.log_likelihood_gathered(hp)
        want = single.log_likelihood_batch(hp)[0]
        assert got.shape == (n,) and np.array_equal(got, want), n
    multi.close()
    shared = MultiGPUFit(*opts, devices=[0, 0])
    assert shared.enable_rccl() is False and "share device" in shared._rccl_error
    hp = cases.halton_params(101, with_beta=True)
    assert_same_lnl(shared.log_likelihood_gathered(hp), single.log_likelihood_batch(hp)[0], chi2_bound(single, hp),
                    what="host fallback of the grouped gather")
    shared.close()


def test_c_abi_rejects_bad_calls_without_crashing():
    """Error conventions of the boundary: NULL handles and buffers, negative sizes, unknown option values and grids the
    kernels cannot take all come back as VK_E_ARG with a message - nothing is launched."""
    import ctypes as C
    import numpy as np
    import victor_amd
    from victor_amd import _native as N
    from tests import cases
    lib = N.load()
    fit = victor_amd.CCFFit(*cases.synth_options(2))
    eng = fit._get_engine()
    ctx = eng._ctx
    opts = eng.make_opts(fit.model, fit.fit_options)
    rows = fit._fit_rows(cases.halton_params(4), fit.model)
    out = np.empty(4)
    dp = N.as_dp
    assert lib.vk_eval_batch(None, C.byref(opts), dp(rows), 4, dp(out), dp(out), None) == -1
    assert lib.vk_eval_batch(ctx, None, dp(rows), 4, dp(out), dp(out), None) == -1
    assert lib.vk_eval_batch(ctx, C.byref(opts), None, 4, dp(out), dp(out), None) == -1
    assert lib.vk_eval_batch(ctx, C.byref(opts), dp(rows), -1, dp(out), dp(out), None) == -1
    assert b"NULL" in lib.vk_last_error(ctx) or lib.vk_last_error(ctx)
    assert lib.vk_eval_batch(ctx, C.byref(opts), dp(rows), 0, None, None, None) == 0          # empty batch is legal
    bad = N.vk_eval_opts.from_buffer_copy(bytes(opts))
    bad.rsd_model = 9
    assert lib.vk_eval_batch(ctx, C.byref(bad), dp(rows), 4, dp(out), dp(out), None) == -1
    bad = N.vk_eval_opts.from_buffer_copy(bytes(opts))
    bad.like_form = -3
    assert lib.vk_eval_batch(ctx, C.byref(bad), dp(rows), 4, dp(out), dp(out), None) == -1
    s = np.linspace(5.0, 100.0, 6)
    mu = np.linspace(0.0, 1.0, 100)
    w = np.ones((2, 100))
    th = np.empty((4, 2, 6))
    assert lib.vk_theory_batch(ctx, C.byref(opts), dp(rows), 4, dp(s), 6, dp(mu), 1, dp(w), 2, dp(th)) == -1
    assert lib.vk_theory_batch(ctx, C.byref(opts), dp(rows), 4, dp(s), 6, dp(mu), 100, dp(w), 7, dp(th)) == -1
    assert lib.vk_theory_batch(ctx, C.byref(opts), dp(rows), 4, dp(s), 6, dp(mu), 100, dp(w), 2, dp(th)) == 0
    assert np.all(np.isfinite(th))
    # the walker loop: bad descriptions are refused at creation, bad calls come back as VK_E_ARG
    ctxs = (C.c_void_p * 1)(ctx)
    cols = np.array([N.P_FSIGMA8, N.P_SIGMAV], dtype=np.int32)
    lo, hi = np.array([0.05, 100.0]), np.array([1.5, 500.0])
    base = np.ascontiguousarray(rows[:4])
    err = C.create_string_buffer(256)
    ip = C.POINTER(C.c_int32)

    def create(n_ctx=1, n_walkers=4, n_params=2, columns=cols, lo_=lo, hi_=hi):
        return lib.vk_walk_create(ctxs, n_ctx, C.byref(opts), n_walkers, n_params, columns.ctypes.data_as(ip), dp(lo_), dp(hi_), dp(base),
                                  1.0, 1, err, len(err))

    assert not create(n_ctx=3) and err.value
    assert not create(n_walkers=0) and b"walkers" in err.value
    assert not create(columns=np.array([N.P_FSIGMA8, N.P_APAR], dtype=np.int32)) and b"column" in err.value       # derived column
    assert not create(columns=np.array([N.VK_WALK_EPSILON, N.VK_WALK_EPSILON], dtype=np.int32)) and b"twice" in err.value
    assert not create(lo_=np.array([0.05, 600.0])) and b"hi > lo" in err.value
    walk = create()
    assert walk
    x = np.array([[0.4, 350.0], [0.5, 380.0], [0.6, 300.0], [0.45, 420.0]])
    l0 = fit.log_likelihood_batch(dict(fsigma8=x[:, 0], sigma_v=x[:, 1], aperp=rows[:4, N.P_APERP], apar=rows[:4, N.P_APAR]))[0]
    dz = np.zeros((3, 4, 2))
    logu = np.zeros((3, 4))
    n_acc, n_ev = C.c_int64(0), C.c_int64(0)
    assert lib.vk_walk_run(None, 3, dp(x), dp(l0), dp(dz), dp(logu), None, None, C.byref(n_acc), C.byref(n_ev)) == -1
    assert lib.vk_walk_run(walk, 3, None, dp(l0), dp(dz), dp(logu), None, None, C.byref(n_acc), C.byref(n_ev)) == -1
    assert lib.vk_walk_run(walk, 3, dp(x), dp(l0), None, dp(logu), None, None, C.byref(n_acc), C.byref(n_ev)) == -1
    assert lib.vk_walk_run(walk, 0, dp(x), dp(l0), None, None, None, None, C.byref(n_acc), C.byref(n_ev)) == 0          # no steps: legal
    # zero increments, acceptance level log(1) = 0: every proposal is the position itself, lnl_prop - lnl = 0 is not > 0: nothing moves
    x0 = x.copy()
    assert lib.vk_walk_run(walk, 3, dp(x), dp(l0), dp(dz), dp(logu), None, None, C.byref(n_acc), C.byref(n_ev)) == 0
    assert np.array_equal(x, x0) and n_acc.value == 0 and n_ev.value == 12
    lib.vk_walk_destroy(walk)
    # the context is still healthy
    lnl, chi2 = fit.log_likelihood_batch(rows)
    assert np.all(np.isfinite(lnl))


def test_ten_million_point_batch_is_chunked_inside_the_library():
    """A batch beyond the 32-bit work-item range of one launch (6.7 M points at 40 s bins; round 2 returned VK_E_ARG there):
    the library cuts it into launches of whole 65536-point blocks on the context's stream.  10 M points of config 2 in ONE call
    against the same rows in two calls of 5 M: bit for bit, every row finite."""
    import victor_amd
    fit = victor_amd.CCFFit(*cases.synth_options(2))
    n = 10_000_000
    rows = fit._fit_rows(cases.halton_params(n), fit.model)
    lnl, chi2 = fit.log_likelihood_batch(rows)
    assert lnl.shape == (n,) and np.all(np.isfinite(lnl)) and np.all(chi2 > 0)
    half = n // 2
    for lo, hi in ((0, half), (half, n)):
        l2, c2 = fit.log_likelihood_batch(rows[lo:hi])
        assert np.array_equal(l2, lnl[lo:hi]) and np.array_equal(c2, chi2[lo:hi])
    # and it is the same function of the row as a small batch through another kernel mapping (to rounding)
    probe = np.linspace(0, n - 1, 50).astype(int)
    l3, c3 = fit.log_likelihood_batch(rows[probe])
    assert_same_chi2(c3, chi2[probe], chi2_bound(fit, rows[probe]), what="probe rows of the 10 M batch")


@pytest.mark.gpu
def test_contexts_give_their_memory_back():
    """Forty contexts created, used (single point, small batch, large batch: device tables, LDS images, scratch, pinned and
    mapped host buffers, events, a stream) and closed: the device's free memory, read from the HIP runtime itself, ends where
    it started (a chain that rebuilds its engine per option set must not creep)."""
    import ctypes
    import victor_amd
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()

    def free_bytes():
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    hp = cases.halton_params(3000, with_beta=True)

    def use_once():
        fit = victor_amd.CCFFit(*cases.boss_options("config"))
        fit.log_likelihood(cases.point(hp, 1))
        fit.log_likelihood_batch({k: v[:50] for k, v in hp.items()})
        fit.log_likelihood_batch(hp)
        for eng in list(fit._engines.values()) if hasattr(fit, "_engines") else [fit._get_engine()]:
            eng.close()

    use_once()                                   # the runtime's own one-off allocations (code object, pools) happen here
    start = free_bytes()
    for _ in range(40):
        use_once()
    assert abs(free_bytes() - start) < 64 << 20, (start, free_bytes())


@pytest.mark.gpu
def test_live_counter_passes_of_the_bench():
    """What bench.py does before it touches the GPU - three child runs of itself under `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE,
    GRBM_GUI_ACTIVE; counters only, the interpreter directly behind `--`) - on a small batch: the HBM traffic of the theory
    kernel and, from ONE child that launches every workload the line quotes a roofline fraction for, the shader cycles and
    the duration of the same dispatches.  Structure only: which kernel, how many dispatches, that counters and timestamps came
    back for every workload - the figures themselves are the bench's to report."""
    import bench
    import bench_pmc
    if not bench_pmc.rocprof_path():
        pytest.skip("rocprofv3 not installed")
    batch = 4096
    t = bench_pmc.live_traffic(batch, "simpson")
    assert t is not None, "a FETCH_SIZE / WRITE_SIZE pass failed (gpurun_out/live_*.err)"
    assert t["kernel"] == "vk::vk_theory_cells_kernel<3, 3, 0, 0, 0>" and t["launches_averaged"] >= 4          # pre-warm + 1 warm-up + 3 steps
    assert t["written_bytes"] >= batch * 120 * 8 and t["bytes_per_launch"] == t["read_bytes"] + t["written_bytes"]   # the theory workspace at least
    c = bench_pmc.live_clocks(batch, "simpson")
    assert c is not None, "the GRBM_GUI_ACTIVE pass failed or its dispatch records do not match what the child launched"
    labels = ["config3", "boss_cmass"] + [name for name, _ in bench.MODEL_OPTIONS] + [bench.FROM_DATA_LABEL]
    assert list(c) == labels
    for name, rec in c.items():
        assert "vk_theory_cells_kernel" in rec["kernel"] and rec["dispatches"] == (6 if name in ("config3", "boss_cmass") else 4), (name, rec)
        assert rec["cycles_per_dispatch"] > 0, (name, rec)
        rec["child_event_ms"], rec["dispatch_ms"], rec["sustained_clock_ghz"]                 # (present; their values are the bench's to report)
    assert c["config3"]["kernel"] == "vk::vk_theory_cells_kernel<3, 3, 0, 0, 0>" and c["kaiser"]["kernel"] == c["euclid_special"]["kernel"]
    f = bench.clock_fields(1.0e9, c, "config3")
    assert f["clock_source"] == "this run" and f["frac_at_sustained_clock"] is not None
