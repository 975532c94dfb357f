"""Same-box A/B of library builds in the small-batch regime (the fused / split launches whose workgroups hand partial sums to
each other): resident batches of 1, 8, 64 and 1024 points, config 3 and BOSS, microseconds per call.
Usage: gpu_handoff_ab.py libA.so libB.so ...   (each build in its own process, two rounds)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOOP = os.path.join(ROOT, "tools", "small_batch_loop.py")
libs = sys.argv[1:]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ, VICTOR_HIP_LIB=os.path.abspath(lib))
        cells = []
        for which in ("3", "boss"):
            for batch in (1, 8, 64, 1024):
                res = subprocess.run([sys.executable, LOOP, which, str(batch), "resident", "2000"], env=env, capture_output=True, text=True)
                try:
                    d = json.loads(res.stdout.strip().splitlines()[-1])
                    cells.append(f"{which}/{batch}: {d['us_per_call']:.2f}")
                except Exception:
                    cells.append(f"{which}/{batch}: FAILED {res.stderr[-200:]}")
        print(f"round {rnd} {os.path.basename(lib):24s} us per call  " + "  ".join(cells), flush=True)
