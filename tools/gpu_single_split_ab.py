"""Work split of a single-point evaluation (development knob VICTOR_HIP_SPLIT=1,4,parts: workgroups per (mu, v) plane) against
(a) the latency of one CCFFit.log_likelihood call and (b) the aggregate rate of 8 / 16 chains through the GPU owner process,
whose launches evaluate every request with that same split.  Fewer workgroups per point = a longer kernel for one point, more
points resident at once."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CALL = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1]); os.chdir(sys.argv[1])
import victor_amd
from tests import cases
info = cases.cobaya_info()["likelihood"]["CCFLikelihood"]
fit = victor_amd.CCFFit(info["model"], info["data"])
p = {"fsigma8": 0.47, "beta": 0.37, "sigma_v": 380, "epsilon": 1.0}
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end: fit.log_likelihood(p)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(2000): fit.log_likelihood(p)
    best = min(best, (time.perf_counter() - t0) / 2000)
print(best * 1e6)
'''


def main():
    import bench
    import multiprocessing as mp
    from victor_amd import broker as B
    for split in (None, "1,4,3", "1,4,2", "1,4,1"):
        env = dict(os.environ, PYTHONPATH=ROOT)
        env.pop("VICTOR_HIP_BROKER", None)
        if split:
            env.update(VICTOR_HIP_DEV="1", VICTOR_HIP_SPLIT=split)
        us = float(subprocess.run([sys.executable, "-c", CALL, ROOT], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1])
        name = f"victor_split_{os.getpid()}_{(split or 'default').replace(',', '_')}"
        srv = subprocess.Popen([sys.executable, "-m", "victor_amd.broker", "--config", "config/boss_cobaya_config.yaml", "--name", name,
                                "--slots", "32"], cwd=ROOT, env=env, stdin=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        rates = {}
        try:
            for P in (4, 8, 16):
                ctx = mp.get_context("spawn")
                barrier, queue = ctx.Barrier(P), ctx.Queue()
                procs = [ctx.Process(target=bench.chain_worker, args=(i, P, 0.8, name, barrier, queue)) for i in range(P)]
                for p in procs:
                    p.start()
                res = [queue.get(timeout=600) for _ in procs]
                for p in procs:
                    p.join(timeout=30)
                rates[P] = sum(r[1] for r in res) / max(r[2] for r in res) if not any(r[4] for r in res) else None
            seg = B._Segment(B.shm_path(name))
            seg.header.stop = 1
            seg.close()
            srv.wait(timeout=20)
        finally:
            if srv.poll() is None:
                srv.kill()
        print(f"split {split or 'default (1,4,2)':16s} single call {us:6.2f} us   chains: " +
              "  ".join(f"P={P} {r / 1e3:6.1f} k/s" for P, r in rates.items()), flush=True)


if __name__ == "__main__":
    main()
