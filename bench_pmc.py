"""The counter passes of bench.py: child runs of the bench under `rocprofv3 --pmc`, made BEFORE the parent touches the GPU.

Three bounded passes, each one counter set, counters only (no trace domain), the interpreter itself directly after `--`:

    FETCH_SIZE, WRITE_SIZE   HBM bytes per launch of the dominant (theory) kernel      -> roofline.traffic
    GRBM_GUI_ACTIVE          shader clock every measured kernel sustains IN THIS RUN    -> *.sustained_clock_ghz,
                                                                                           *.frac_at_sustained_clock

The clock of a dispatch is GRBM_GUI_ACTIVE / 8 / its duration (rocprofv3 reports the sum over the 8 XCDs; MI355X_MICROARCH.md,
"DVFS give-back"): counter value and start / end timestamps are columns of ONE row of the profiler's counter_collection.csv,
so cycles and time belong to the same dispatch on the same box.  Nothing here reads a figure of another lease: a pass that
cannot be made yields None and the bench line carries null.

The parsing and the arithmetic are pure functions of CSV rows (tests/test_bench_pmc.py runs them on canned files).
"""

import csv
import glob
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.abspath(__file__))
PASS_TIMEOUT_S = 120
XCDS = 8                          # GRBM_GUI_ACTIVE is reported summed over the XCDs


def rocprof_path():
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    return exe if os.path.isfile(exe) else None


def run_bounded(cmd, cwd, env, timeout):
    """subprocess.run with captured output whose time-out ends the child's whole process GROUP (the profiler and the program
    it started), so that nothing of a hung pass keeps the GPU."""
    proc = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.communicate()
        raise
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)


def read_counter_rows(out_dir, counter, kernel_filter="vk_theory"):
    """Rows of every *counter_collection.csv under ``out_dir`` for one counter and the kernels whose name contains
    ``kernel_filter``, in dispatch order: dicts {dispatch, kernel, value, start_ns, end_ns}."""
    rows = []
    for f in sorted(glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter or kernel_filter not in r.get("Kernel_Name", ""):
                    continue
                rows.append({"dispatch": int(r["Dispatch_Id"]), "kernel": r["Kernel_Name"], "value": float(r["Counter_Value"]),
                             "start_ns": int(r["Start_Timestamp"]) if r.get("Start_Timestamp") else None,
                             "end_ns": int(r["End_Timestamp"]) if r.get("End_Timestamp") else None})
    rows.sort(key=lambda r: r["dispatch"])
    return rows


def short_kernel_name(name):
    return name.replace("void ", "").split("(")[0]


def pmc_pass(counter, child_args, log_dir=None, log_tag=None):
    """One child run of bench.py under `rocprofv3 --pmc <counter>`.  Returns (rows, child stdout) or None."""
    exe = rocprof_path()
    if not exe:
        return None
    out_dir = tempfile.mkdtemp(prefix="victor_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--", sys.executable,
           os.path.join(ROOT, "bench.py"), *child_args]
    try:
        res = run_bounded(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=PASS_TIMEOUT_S)
        rows = read_counter_rows(out_dir, counter)
        if res.returncode != 0 or not rows:
            if log_dir:
                with open(os.path.join(log_dir, f"live_{log_tag or counter}.err"), "w") as fh:
                    fh.write(res.stdout[-4000:] + "\n" + res.stderr[-4000:])
            return None
        return rows, res.stdout
    except (OSError, subprocess.TimeoutExpired, ValueError, KeyError):
        return None
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


# ---- HBM traffic -----------------------------------------------------------------------------------------------------------

def dominant_kernel_average(rows):
    """(kernel, average counter value per launch, launches) of the kernel with the largest total."""
    by = {}
    for r in rows:
        by.setdefault(r["kernel"], []).append(r["value"])
    kernel = max(by, key=lambda k: sum(by[k]))
    return kernel, sum(by[kernel]) / len(by[kernel]), len(by[kernel])


def traffic_from_rows(fetch_rows, write_rows):
    """HBM bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md prescribes (KiB -> bytes, FETCH_SIZE x 2
    on gfx950)."""
    fk, fv, fn = dominant_kernel_average(fetch_rows)
    _, wv, _ = dominant_kernel_average(write_rows)
    fetch, write = 2.0 * fv * 1024.0, wv * 1024.0
    return {"bytes_per_launch": fetch + write, "read_bytes": fetch, "written_bytes": write, "kernel": short_kernel_name(fk),
            "launches_averaged": fn,
            "method": "two child runs of this program under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, "
                      "counters only); KiB -> bytes, FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md)"}


def live_traffic(batch, simpson_even, log_dir=None):
    """HBM bytes per launch of the dominant (theory) kernel, measured NOW on the bench's own batch.  A dict, or None when the
    profiler is not there or a pass fails (the line then carries `traffic: null`)."""
    child = ["--steps", "3", "--warmup", "1", "--batch", str(batch), "--no-cpu-baseline", "--no-boss", "--no-live-traffic",
             "--simpson-even", simpson_even]
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        res = pmc_pass(counter, child, log_dir)
        if res is None:
            return None
        got[counter] = res[0]
    return traffic_from_rows(got["FETCH_SIZE"], got["WRITE_SIZE"])


# ---- sustained clocks ------------------------------------------------------------------------------------------------------

def clocks_from_rows(rows, sequence):
    """Per-workload sustained clock from the GRBM_GUI_ACTIVE rows of ONE child run.

    ``sequence``: what the child reports it launched, in order - dicts {label, warm, timed[, event_ms]}: `warm` untimed theory-
    kernel launches followed by `timed` measured ones.  ``rows``: the theory-kernel dispatches in dispatch order.  The counts
    must add up, and a workload's timed dispatches must all be of one kernel; otherwise None - a clock is never guessed."""
    want = sum(int(s["warm"]) + int(s["timed"]) for s in sequence)
    if want == 0 or len(rows) != want:
        return None
    out, at = {}, 0
    for s in sequence:
        at += int(s["warm"])
        mine = rows[at:at + int(s["timed"])]
        at += int(s["timed"])
        if not mine or len({r["kernel"] for r in mine}) != 1 or any(r["start_ns"] is None or r["end_ns"] is None for r in mine):
            return None
        dur_ns = sum(r["end_ns"] - r["start_ns"] for r in mine)
        cycles = sum(r["value"] for r in mine)
        if dur_ns <= 0 or cycles <= 0:
            return None
        out[s["label"]] = {"sustained_clock_ghz": cycles / XCDS / dur_ns, "cycles_per_dispatch": cycles / XCDS / len(mine),
                           "kernel": short_kernel_name(mine[0]["kernel"]), "dispatches": len(mine),
                           "dispatch_ms": dur_ns / len(mine) * 1e-6, "child_event_ms": s.get("event_ms")}
    return out


def parse_clock_child(stdout):
    """The child's own record: the last stdout line that is a JSON object with a `clock_pass` list."""
    for line in reversed(stdout.splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                rec = json.loads(line)
            except ValueError:
                continue
            if isinstance(rec.get("clock_pass"), list):
                return rec["clock_pass"]
    return None


def live_clocks(batch, simpson_even, log_dir=None):
    """Sustained shader clock of every kernel the bench line quotes a roofline fraction for, measured in THIS run: one child
    (`bench.py --clock-pass`) under `rocprofv3 --pmc GRBM_GUI_ACTIVE`.  {label: {...}} or None."""
    res = pmc_pass("GRBM_GUI_ACTIVE", ["--clock-pass", "--batch", str(batch), "--simpson-even", simpson_even], log_dir, "clock_pass")
    if res is None:
        return None
    rows, stdout = res
    sequence = parse_clock_child(stdout)
    if not sequence:
        return None
    return clocks_from_rows(rows, sequence)


CLOCK_METHOD = ("a child run of this program under rocprofv3 --pmc GRBM_GUI_ACTIVE (counters only), the same kernel on the same batch: "
                "sustained_clock_ghz = GRBM_GUI_ACTIVE / 8 / dispatch duration, counter and timestamps of the same dispatches "
                "(MI355X_MICROARCH.md: within 3 % of the in-kernel clock for dispatches of 10 ms or more, reads high below about "
                "0.3 ms); frac_at_sustained_clock = algorithmic flops per launch / (GRBM_GUI_ACTIVE / 8 shader cycles of a dispatch) / "
                "(peak flops per cycle = peak / 2.4 GHz) - cycles and work of the SAME dispatches, so it does not depend on which "
                "clock this box, or the timed loop beside it, happened to hold")
