"""Soak of the GPU owner process (victor_amd/broker.py): for SECONDS seconds chains come and go - attach, evaluate points of a
fixed list in random order, compare EVERY value bit for bit with what this process computed for that point on a context of
its own, detach after a random time; a few are killed instead of detaching; more chains want in than there are mailboxes.
Usage: gpu_broker_soak.py [seconds=30] [mailboxes=6] [chains at a time=10]"""
import json
import os
import random
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHAIN = r'''
import os, sys, json, random, time
root, name, expect_file, seconds, seed = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), int(sys.argv[5])
os.chdir(root); sys.path.insert(0, root)
os.environ["VICTOR_HIP_BROKER"] = name
import victor_amd
from victor_amd import _native
from tests import cases
info = cases.cobaya_info()["likelihood"]["CCFLikelihood"]
pts, want = json.load(open(expect_file))
rng = random.Random(seed)
t_wait = time.time()
while True:
    try:
        fit = victor_amd.CCFFit(info["model"], info["data"])
        fit.log_likelihood(dict(pts[0]))
        break
    except _native.NativeError as exc:          # every mailbox taken: wait for one
        if "taken" not in str(exc) or time.time() - t_wait > 60:
            raise
        fit = None
        time.sleep(0.05)
n = bad = 0
t_end = time.time() + seconds
while time.time() < t_end:
    i = rng.randrange(len(pts))
    got = fit.log_likelihood(dict(pts[i]))
    n += 1
    if list(got) != want[i]:
        bad += 1
print(json.dumps({"calls": n, "mismatches": bad, "gpu_library_loaded": _native._lib is not None}), flush=True)
'''


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    slots = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    import victor_amd
    from tests import cases
    from victor_amd import broker as B
    os.chdir(ROOT)
    info = cases.cobaya_info()["likelihood"]["CCFLikelihood"]
    fit = victor_amd.CCFFit(info["model"], info["data"], broker=False)
    h = cases.halton(64, bases=(2, 3, 5, 7))
    pts = [{"fsigma8": 0.05 + 1.45 * a, "beta": 0.2 + 0.4 * b, "sigma_v": 100 + 400 * c, "epsilon": 0.8 + 0.4 * d} for a, b, c, d in h.tolist()]
    want = [list(fit.log_likelihood(dict(p))) for p in pts]
    expect_file = f"/tmp/victor_soak_{os.getpid()}.json"
    json.dump([pts, want], open(expect_file, "w"))
    name = f"victor_soak_{os.getpid()}"
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("VICTOR_HIP_BROKER", None)
    srv = subprocess.Popen([sys.executable, "-m", "victor_amd.broker", "--config", "config/boss_cobaya_config.yaml", "--name", name,
                            "--slots", str(slots)], cwd=ROOT, env=env, stdin=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rng = random.Random(7)
    live, done, killed, started = [], [], 0, 0
    t_end = time.time() + seconds
    try:
        while time.time() < t_end or live:
            while len(live) < width and time.time() < t_end:
                p = subprocess.Popen([sys.executable, "-c", CHAIN, ROOT, name, expect_file, str(rng.uniform(0.3, 3.0)), str(started)],
                                     env=env, stdout=subprocess.PIPE, text=True)
                live.append((p, time.time(), rng.random() < 0.15))
                started += 1
            time.sleep(0.1)
            for item in list(live):
                p, t0, doomed = item
                if doomed and time.time() - t0 > 1.5 and p.poll() is None:
                    p.send_signal(signal.SIGKILL)
                    p.wait()
                    killed += 1
                    live.remove(item)
                elif p.poll() is not None:
                    out = p.stdout.read().strip().splitlines()
                    done.append((p.returncode, json.loads(out[-1]) if out and p.returncode == 0 else None))
                    live.remove(item)
        seg = B._Segment(B.shm_path(name))
        st = seg.header.stats
        stats = {"batches": int(st.batches), "evals": int(st.evals), "max_batch": int(st.max_batch)}
        seg.header.stop = 1
        seg.close()
        srv.wait(timeout=30)
    finally:
        if srv.poll() is None:
            srv.kill()
        os.unlink(expect_file)
    ok = [d for rc, d in done if rc == 0 and d]
    print(json.dumps({"seconds": seconds, "mailboxes": slots, "chains_started": started, "finished": len(ok), "failed": len(done) - len(ok),
                      "killed_with_sigkill": killed, "calls": sum(d["calls"] for d in ok), "mismatches": sum(d["mismatches"] for d in ok),
                      "any_chain_loaded_the_gpu_library": any(d["gpu_library_loaded"] for d in ok), "owner": stats,
                      "owner_exit_code": srv.returncode}))


if __name__ == "__main__":
    main()
