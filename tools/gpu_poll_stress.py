"""Several processes with a context each, all making split launches of 1 ... 8 points that hand over by polling, for SECONDS:
every result must repeat the process's own first evaluation of the same sub-batch bit for bit, and no call may fail (the
polling hand-off waits inside the kernel - vk_common.h: kPollEmpty - so this is the load under which a flaw in it would show
as a time-out).  Usage: gpu_poll_stress.py [seconds] [processes <= 5]"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, seconds, queue, barrier):
    import numpy as np
    import victor_amd
    from tests import cases
    boss = rank % 2 == 0
    fit = victor_amd.CCFFit(*(cases.boss_options("config") if boss else cases.synth_options(3)))
    hp = cases.halton_params(8, with_beta=boss)
    rng = np.random.default_rng(rank)
    ref = {}
    for n in range(1, 9):
        sub = {k: v[:n] for k, v in hp.items()}
        ref[n] = np.concatenate(fit.log_likelihood_batch(sub))
    barrier.wait(timeout=300)
    calls = bad = 0
    t_end = time.perf_counter() + seconds
    try:
        while time.perf_counter() < t_end:
            n = int(rng.integers(1, 9))
            sub = {k: v[:n] for k, v in hp.items()}
            got = np.concatenate(fit.log_likelihood_batch(sub))
            calls += 1
            if not np.array_equal(got, ref[n]):
                bad += 1
        queue.put((rank, calls, bad, fit._get_engine().last_kernel(), ""))
    except Exception as exc:       # noqa: BLE001
        queue.put((rank, calls, bad, "", repr(exc)))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    procs = min(int(sys.argv[2]) if len(sys.argv) > 2 else 5, 5)
    ctx = mp.get_context("spawn")
    queue, barrier = ctx.Queue(), ctx.Barrier(procs)
    ps = [ctx.Process(target=worker, args=(r, seconds, queue, barrier)) for r in range(procs)]
    for p in ps:
        p.start()
    res = [queue.get(timeout=seconds + 600) for _ in ps]
    for p in ps:
        p.join(timeout=30)
    total = sum(r[1] for r in res)
    print(f"{procs} processes x {seconds:.0f} s: {total} calls of 1-8 points ({total / seconds:.0f} calls/s), "
          f"mismatches {sum(r[2] for r in res)}, errors {[r[4] for r in res if r[4]]}, kernels {sorted({r[3] for r in res})}")
    sys.exit(1 if any(r[2] or r[4] for r in res) else 0)


if __name__ == "__main__":
    main()
