"""The GPU owner process and its chains on a real MI355X: vk_serve_mailboxes (native loop) behind the unchanged cobaya plug-in."""

import os
import subprocess
import sys
import time

import numpy as np
import pytest

from tests import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _chain(idx, name, points, barrier, queue):
    """A cobaya-style chain in a child process: the plug-in, one point per calculate(), VICTOR_HIP_BROKER in the environment."""
    try:
        os.chdir(ROOT)
        os.environ["VICTOR_HIP_BROKER"] = name
        sys.path.insert(0, os.path.join(ROOT, "victor", "likelihoods"))
        from CCFLikelihood import CCFLikelihood
        info = cases.cobaya_info()["likelihood"]["CCFLikelihood"]
        lk = CCFLikelihood({"model": info["model"], "data": info["data"]})
        barrier.wait(timeout=300)             # all chains start together: their requests share launches
        out = []
        for rep in range(3):
            for p in points:
                state = {}
                lk.calculate(state, want_derived=True, **p)
                out.append((state["logp"], state["derived"]["chi2_ccf_correct"]))
        from victor_amd import _native
        queue.put((idx, out, lk.ccf._engine is None and _native._lib is None, None))
    except Exception as exc:       # noqa: BLE001
        try:
            barrier.abort()
        except Exception:
            pass
        queue.put((idx, [], False, repr(exc)))


def _points(meta):
    pts = [{k: pt[k] for k in ("fsigma8", "beta", "sigma_v", "epsilon")} for pt in meta["boss_points"][:2]]
    h = cases.halton(14, bases=(2, 3, 5, 7, 11))
    for a, b, c, d, _ in h.tolist():
        pts.append({"fsigma8": 0.05 + 1.45 * a, "beta": 0.2 + 0.4 * b, "sigma_v": 100 + 400 * c, "epsilon": 0.8 + 0.4 * d})
    return pts


@pytest.mark.parametrize("depth", [1, 4])
def test_four_chains_through_one_broker_are_bit_identical_to_the_single_process_values(depth):
    """Four child chains (processes that never load the HIP library) attach to one owner process; every logp / chi2 they get
    equals - bit for bit - what this process computes alone for the same point, and the reference's goldens to 1e-9.
    depth 1: one launch at a time, so requests that wait share the next one; depth 4: up to four launches in flight."""
    import multiprocessing as mp
    import victor_amd
    from victor_amd import broker as B
    g, meta = cases.golden_outputs()
    pts = _points(meta)
    name = f"victor_test_{os.getpid()}_{depth}"
    env = dict(os.environ, PYTHONPATH=ROOT + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else ""))
    env.pop("VICTOR_HIP_BROKER", None)
    srv = subprocess.Popen([sys.executable, "-m", "victor_amd.broker", "--config", "config/boss_cobaya_config.yaml", "--name", name,
                            "--slots", "8", "--depth", str(depth)], cwd=ROOT, env=env, stdin=subprocess.DEVNULL)
    try:
        ctx = mp.get_context("spawn")
        barrier, queue = ctx.Barrier(4), ctx.Queue()
        procs = [ctx.Process(target=_chain, args=(i, name, pts[i::2] if i < 2 else pts, barrier, queue)) for i in range(4)]
        for p in procs:
            p.start()
        res = sorted(queue.get(timeout=600) for _ in procs)
        for p in procs:
            p.join(timeout=30)
        assert [r[3] for r in res] == [None] * 4, res
        assert all(r[2] for r in res)                      # no engine, no libvictor_hip.so in any chain process

        # the single-process values: this process, its own context, the same plug-in options
        cwd = os.getcwd()
        os.chdir(ROOT)
        try:
            info = cases.cobaya_info()["likelihood"]["CCFLikelihood"]
            fit = victor_amd.CCFFit(info["model"], info["data"], broker=False)
        finally:
            os.chdir(cwd)
        alone = {i: fit.log_likelihood(dict(p)) for i, p in enumerate(pts)}
        for idx, out, _, _ in res:
            mine = list(range(idx, len(pts), 2)) if idx < 2 else list(range(len(pts)))
            assert len(out) == 3 * len(mine)
            for k, (lnl, chi2) in enumerate(out):
                assert (lnl, chi2) == alone[mine[k % len(mine)]]                         # bit for bit, every repetition
        for i in range(2):
            assert abs(alone[i][0] - g["boss_cobaya_lnl"][i]) < 1e-9 * abs(alone[i][0])
            assert abs(alone[i][1] - g["boss_cobaya_chi2"][i]) < 1e-9 * g["boss_cobaya_chi2"][i]

        seg = B._Segment(B.shm_path(name))
        st = seg.header.stats
        n_calls = sum(len(r[1]) for r in res)
        deadline = time.time() + 5
        while int(seg.header.stats.evals) < n_calls + 1 and time.time() < deadline:      # the header is refreshed every 0.25 s
            time.sleep(0.05)
        assert int(st.evals) >= n_calls and int(seg.header.depth) == depth, (int(st.evals), int(st.batches), int(st.max_batch))
        if depth == 1:                                     # four chains, one launch at a time: requests did share launches
            assert int(st.max_batch) >= 2 and int(st.batches) < int(st.evals), (int(st.evals), int(st.batches), int(st.max_batch))
        seg.header.stop = 1
        seg.close()
        assert srv.wait(timeout=30) == 0
        assert not os.path.exists(B.shm_path(name))
    finally:
        if srv.poll() is None:
            srv.kill()


def test_auto_broker_is_started_by_the_first_chain_and_goes_away_after_the_last(tmp_path):
    """VICTOR_HIP_BROKER=auto: no broker exists; two chains of one 'job' elect one of themselves to start it, both attach, both
    get the single-process value; the owner process leaves a few seconds after they have gone and removes its segment."""
    from victor_amd import broker as B
    code = r'''
import os, sys, json
root = sys.argv[1]
os.chdir(root)
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "victor", "likelihoods"))
from CCFLikelihood import CCFLikelihood
from tests import cases
info = cases.cobaya_info()["likelihood"]["CCFLikelihood"]
lk = CCFLikelihood({"model": info["model"], "data": info["data"]})
state = {}
lk.calculate(state, want_derived=True, fsigma8=0.47, beta=0.37, sigma_v=380, epsilon=1.0)
from victor_amd import _native
print(json.dumps({"logp": state["logp"], "chi2": state["derived"]["chi2_ccf_correct"], "name": lk.ccf._broker_client.name,
                  "gpu_free": lk.ccf._engine is None and _native._lib is None}))
'''
    env = dict(os.environ, VICTOR_HIP_BROKER="auto", VICTOR_HIP_BROKER_LOG=str(tmp_path / "broker.log"))
    procs = [subprocess.Popen([sys.executable, "-c", code, ROOT], env=env, stdout=subprocess.PIPE, text=True) for _ in range(2)]
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, (tmp_path / "broker.log").read_text() if (tmp_path / "broker.log").exists() else o
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    g, _ = cases.golden_outputs()
    assert outs[0]["name"] == outs[1]["name"] and outs[0]["gpu_free"] and outs[1]["gpu_free"]
    assert outs[0]["logp"] == outs[1]["logp"] and outs[0]["chi2"] == outs[1]["chi2"]
    assert abs(outs[0]["logp"] - g["boss_cobaya_lnl"][0]) < 1e-9 * abs(outs[0]["logp"])
    path = B.shm_path(outs[0]["name"])
    deadline = time.time() + 30
    while os.path.exists(path) and time.time() < deadline:
        time.sleep(0.25)
    assert not os.path.exists(path), "the auto-started broker did not leave after its chains had gone"


def test_broker_survives_a_killed_chain_and_hands_its_mailbox_on(tmp_path):
    """A chain that dies without detaching (SIGKILL in the middle of its loop) must cost nothing but its mailbox for a moment:
    the owner's reaper frees it (the pid is gone), the other chain's values stay correct throughout, and a chain that comes
    later gets a mailbox even though the segment has only two."""
    import signal
    import victor_amd
    from victor_amd import broker as B
    name = f"victor_test_kill_{os.getpid()}"
    env = dict(os.environ, PYTHONPATH=ROOT + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else ""))
    env.pop("VICTOR_HIP_BROKER", None)
    srv = subprocess.Popen([sys.executable, "-m", "victor_amd.broker", "--config", "config/boss_cobaya_config.yaml", "--name", name,
                            "--slots", "2"], cwd=ROOT, env=env, stdin=subprocess.DEVNULL)
    loop = r'''
import os, sys, json, time
root, name, seconds = sys.argv[1], sys.argv[2], float(sys.argv[3])
os.chdir(root); sys.path.insert(0, root)
os.environ["VICTOR_HIP_BROKER"] = name
import victor_amd
from tests import cases
info = cases.cobaya_info()["likelihood"]["CCFLikelihood"]
fit = victor_amd.CCFFit(info["model"], info["data"])
p = {"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0}
first = fit.log_likelihood(dict(p))
print(json.dumps({"ready": first}), flush=True)
n, t_end = 0, time.time() + seconds
while time.time() < t_end:
    assert fit.log_likelihood(dict(p)) == first
    n += 1
print(json.dumps({"calls": n, "slot": fit._broker_client.slot}), flush=True)
'''
    try:
        victim = subprocess.Popen([sys.executable, "-c", loop, ROOT, name, "60"], env=env, stdout=subprocess.PIPE, text=True)
        steady = subprocess.Popen([sys.executable, "-c", loop, ROOT, name, "4"], env=env, stdout=subprocess.PIPE, text=True)
        ready = json_line(victim.stdout.readline())
        assert "ready" in ready
        time.sleep(0.5)
        victim.send_signal(signal.SIGKILL)                 # in the middle of its loop, mailbox attached, maybe a request pending
        victim.wait(timeout=30)
        out, _ = steady.communicate(timeout=120)
        lines = [json_line(ln) for ln in out.strip().splitlines()]
        assert steady.returncode == 0 and lines[0]["ready"] == ready["ready"] and lines[-1]["calls"] > 1000
        # both mailboxes were taken; the reaper must have freed the victim's by now (it runs every 0.25 s)
        late = subprocess.run([sys.executable, "-c", loop, ROOT, name, "0.2"], env=env, capture_output=True, text=True, timeout=120)
        assert late.returncode == 0, late.stderr[-2000:]
        assert json_line(late.stdout.strip().splitlines()[0])["ready"] == ready["ready"]
        g, _ = cases.golden_outputs()
        assert abs(ready["ready"][0] - g["boss_cobaya_lnl"][0]) < 1e-9 * abs(ready["ready"][0])
        seg = B._Segment(B.shm_path(name))
        seg.header.stop = 1
        seg.close()
        assert srv.wait(timeout=30) == 0
    finally:
        for p in (srv,):
            if p.poll() is None:
                p.kill()


def json_line(text):
    import json
    return json.loads(text)


def test_chains_example_gives_the_same_chains_brokered_and_direct():
    """examples/run_chains.py - independent Metropolis chains, one process each, one ``calculate`` per step, the reference's way
    of sampling: through the shared GPU owner (VICTOR_HIP_BROKER=auto) and with a GPU context per chain the SAME chains come
    out, sample for sample (every likelihood value is bit-identical, so every accept / reject decision is)."""
    import json
    runs = {}
    for tag, extra in (("brokered", []), ("direct", ["--direct"])):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "run_chains.py"), "--chains", "3", "--steps", "250"] + extra,
                             capture_output=True, text=True, timeout=600, env={k: v for k, v in os.environ.items() if k != "VICTOR_HIP_BROKER"})
        assert res.returncode == 0, res.stderr[-3000:] + res.stdout[-500:]
        runs[tag] = json.loads(res.stdout.strip().splitlines()[-1])
    a, b = runs["brokered"], runs["direct"]
    assert a["chains_never_loaded_the_gpu_library"] is True and b["chains_never_loaded_the_gpu_library"] is False
    assert a["last_logp"] == b["last_logp"] and a["mean"] == b["mean"] and a["acceptance"] == b["acceptance"]
    assert a["likelihood_evaluations"] == b["likelihood_evaluations"] > 600 and 0.02 < a["acceptance"] < 0.9
    assert all(250 < v < 300 for v in a["last_logp"])            # the chains sit near the maximum (lnL = 284.8 at the reference point)


_MAILBOX_CHAIN = r'''
import json, os, sys, time
root, name, digest, rows_file, reps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
sys.path.insert(0, root)
from victor_amd import broker as B
rows = json.load(open(rows_file))
cl = B.BrokerClient(name, digest, timeout=120)
print("attached", flush=True)
sys.stdin.readline()                          # the parent releases all chains together: their requests share launches
out = [cl.eval_point(r) for _ in range(reps) for r in rows]
cl.close()
print(json.dumps(out))
'''


@pytest.mark.parametrize("rsd_model", ["kaiser", "euclid_special"])
def test_models_on_the_cells_kernel_are_served_with_the_single_point_split(rsd_model, tmp_path):
    """kaiser / euclid_special always take the cells kernel, whose ranges per point follow the batch size: through the mailbox
    server they must follow the SINGLE-point rule whatever shares the launch (round-4 advice: with a.n the ranges went from 256
    to 1024 cells once eight requests shared a launch - a chain's value then depended, at rounding level, on how many other
    chains posted at the same moment).  Sixteen chains on one launch at a time fill launches of eight; every value equals the
    single-process one bit for bit."""
    import json
    import victor_amd
    from victor_amd import broker as B
    info = cases.cobaya_info()["likelihood"]["CCFLikelihood"]
    model = dict(info["model"], rsd_model=rsd_model, dir=ROOT)          # (the file paths of the configuration are relative)
    data = dict(info["data"], dir=ROOT)
    fit = victor_amd.CCFFit(model, data, broker=False)
    h = cases.halton(12, bases=(2, 3, 5, 7, 11))
    pts = [{"fsigma8": 0.05 + 1.45 * a, "beta": 0.2 + 0.4 * b, "sigma_v": 100 + 400 * c, "epsilon": 0.8 + 0.4 * d, "M": 0.9 + 0.2 * e,
            "Q": 1.1 - 0.2 * e} for a, b, c, d, e in h.tolist()]
    alone = [fit.log_likelihood(dict(p)) for p in pts]
    assert fit._get_engine().last_kernel() == "vk_theory_cells_kernel"
    rows = [[float(x) for x in fit._fit_rows(dict(p), fit.model)[0]] for p in pts]
    rows_file = tmp_path / "rows.json"
    rows_file.write_text(json.dumps(rows))
    cfg = tmp_path / "cfg.json"
    cfg.write_text(json.dumps({"model": model, "data": data}, default=str))
    name = f"victor_test_{rsd_model}_{os.getpid()}"
    digest = B.config_digest(model, data)
    env = dict(os.environ, PYTHONPATH=ROOT + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else ""))
    env.pop("VICTOR_HIP_BROKER", None)
    srv = subprocess.Popen([sys.executable, "-m", "victor_amd.broker", "--config-json", str(cfg), "--name", name, "--slots", "16",
                            "--depth", "1", "--max-batch", "8", "--digest", digest], cwd=ROOT, env=env, stdin=subprocess.DEVNULL)
    chains = []
    try:
        chains = [subprocess.Popen([sys.executable, "-c", _MAILBOX_CHAIN, ROOT, name, digest, str(rows_file), "40"], env=env,
                                   stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for _ in range(16)]
        for c in chains:
            assert c.stdout.readline().strip() == "attached"
        for c in chains:
            c.stdin.write("go\n")
            c.stdin.flush()
        for c in chains:
            out, _ = c.communicate(timeout=300)
            assert c.returncode == 0
            got = json.loads(out.strip().splitlines()[-1])
            assert len(got) == 40 * len(pts)
            for k, (lnl, chi2) in enumerate(got):
                assert (lnl, chi2) == alone[k % len(pts)], (k, lnl, chi2, alone[k % len(pts)])       # bit for bit
        seg = B._Segment(B.shm_path(name))
        deadline = time.time() + 5
        while int(seg.header.stats.evals) < 16 * 40 * len(pts) and time.time() < deadline:      # the header is refreshed every 0.25 s
            time.sleep(0.05)
        assert int(seg.header.stats.max_batch) == 8, int(seg.header.stats.max_batch)        # launches of eight did happen
        seg.header.stop = 1
        seg.close()
        assert srv.wait(timeout=30) == 0
    finally:
        for p in chains + [srv]:
            if p.poll() is None:
                p.kill()
