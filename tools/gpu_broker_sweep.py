"""Mailbox server (victor_amd/broker.py) parameter sweep on one GPU: serving threads, launches in flight per thread (--depth),
requests per launch (--max-batch) and the gather window against the aggregate evaluations/s of P cobaya-style chains (bench.chain_worker).  Run before anything
touches the GPU in this process (it only starts child processes)."""
import json
import multiprocessing as mp
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(P, name, seconds=0.7):
    ctx = mp.get_context("spawn")
    barrier, queue = ctx.Barrier(P), ctx.Queue()
    procs = [ctx.Process(target=bench.chain_worker, args=(i, P, seconds, name, barrier, queue)) for i in range(P)]
    for p in procs:
        p.start()
    res = [queue.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    if any(r[4] for r in res):
        return None
    return sum(r[1] for r in res) / max(r[2] for r in res)


def main():
    chains = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("4", "8", "16"))]
    grid = [(d, b, w, t) for t in (1, 2, 4) for d in (2, 4) for b in (2, 4, 8) for w in (0.0, 3.0)]
    if len(sys.argv) > 2:
        grid = [tuple(float(x) if "." in x else int(x) for x in g.split(":")) for g in sys.argv[2].split(",")]
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("VICTOR_HIP_BROKER", None)
    print("threads depth max_batch window_us | " + "  ".join(f"P={p:<2d} kevals/s (mean batch)" for p in chains), flush=True)
    for row in grid:
        depth, cap, window = row[:3]
        threads = row[3] if len(row) > 3 else 1
        name = f"victor_sweep_{os.getpid()}_{threads}_{depth}_{cap}_{int(window * 10)}"
        srv = subprocess.Popen([sys.executable, "-m", "victor_amd.broker", "--config", "config/boss_cobaya_config.yaml", "--name", name,
                                "--slots", "32", "--depth", str(depth), "--max-batch", str(cap), "--window-us", str(window),
                                "--threads", str(threads)],
                               cwd=ROOT, env=env, stdin=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        try:
            from victor_amd import broker as B
            cells = []
            prev = (0, 0)
            for P in chains:
                rate = run(P, name)
                time.sleep(0.3)
                seg = B._Segment(B.shm_path(name))
                st = seg.header.stats
                now = (int(st.evals), int(st.batches))
                seg.close()
                mb = (now[0] - prev[0]) / max(now[1] - prev[1], 1)
                prev = now
                cells.append(f"{(rate or 0) / 1e3:7.1f} ({mb:4.1f})")
            print(f"{threads:7d} {depth:5d} {cap:9d} {window:9.1f} | " + "   ".join(cells), flush=True)
            seg = B._Segment(B.shm_path(name))
            seg.header.stop = 1
            seg.close()
            srv.wait(timeout=20)
        finally:
            if srv.poll() is None:
                srv.kill()


if __name__ == "__main__":
    main()
