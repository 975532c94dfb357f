"""profiles/traffic_latest.json from the PMC passes of tools/gpu_profile.sh (bench workload, config 3) and
tools/gpu_profile_boss.sh (BOSS CMASS): HBM bytes per launch of the dominant kernel, FETCH_SIZE / WRITE_SIZE collected in
separate rocprofv3 --pmc runs and corrected as MI355X_MICROARCH.md prescribes (KiB -> bytes, FETCH_SIZE x 2 on gfx950).
bench.py quotes these figures (`roofline.traffic_profiled`, `boss_cmass.traffic_profiled`) next to its own timings.
Usage: python tools/update_traffic.py <gpurun_out dir of the bench profile> <gpurun_out dir of the BOSS profile> <label>
       [dispersion=<dir> kaiser=<dir> euclid_special=<dir>]      (profiles of tools/gpu_profile_model.sh: sustained clocks)
Run it at the commit the profiles were taken at: the hash of the kernel sources is stored with the counters."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def effective_clock_ghz(sm, kernel):
    """Shader clock the kernel sustained: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / its average duration under the same
    counter pass's neighbour, the kernel-trace run (rocprofv3 --kernel-trace --stats)."""
    cs = sm.get("pmc", {}).get(kernel, {})
    if "GRBM_GUI_ACTIVE" not in cs:
        return None
    for k, st in sm.get("kernel_trace", {}).items():
        if k.split("(")[0] == kernel.split("(")[0] and st.get("avg_ns"):
            return cs["GRBM_GUI_ACTIVE"]["avg"] / 8.0 / float(st["avg_ns"])
    return None


def one(dir_, batch, n_data, alg_note):
    with open(os.path.join(dir_, "summary.json")) as fh:
        sm = json.load(fh)
    best = None
    for k, cs in sm.get("pmc", {}).items():
        if "vk_theory" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            fetch, write = 2.0 * cs["FETCH_SIZE"]["avg"] * 1024.0, cs["WRITE_SIZE"]["avg"] * 1024.0
            if best is None or fetch + write > best["theory_kernel_hbm_bytes_per_launch"]:
                best = {"kernel": k.replace("void ", "").split("(")[0], "batch": batch,
                        "theory_kernel_hbm_bytes_per_launch": fetch + write, "read_bytes": fetch, "written_bytes": write,
                        "algorithmic_bytes_per_launch": (96 + 16) * batch, "note": alg_note,
                        "effective_clock_ghz": effective_clock_ghz(sm, k)}
    return best


def clock_only(dir_):
    """The sustained clock of the dominant theory kernel of a model-option profile (tools/gpu_profile_model.sh)."""
    path = os.path.join(dir_, "summary.json")
    if not os.path.isfile(path):
        return None
    with open(path) as fh:
        sm = json.load(fh)
    best = None
    for k, st in sm.get("kernel_trace", {}).items():
        total = float(st.get("avg_ns", 0)) * float(st.get("n", 0))
        if "vk_theory" in k and (best is None or total > best[1]):
            best = (k, total)
    if best is None:
        return None
    key = next((k for k in sm.get("pmc", {}) if k.split("(")[0] == best[0].split("(")[0]), None)
    return {"kernel": best[0].replace("void ", "").split("(")[0], "effective_clock_ghz": effective_clock_ghz(sm, key) if key else None}


def main():
    bench_dir, boss_dir, label = sys.argv[1], sys.argv[2], sys.argv[3]
    models = dict(a.split("=", 1) for a in sys.argv[4:])        # e.g. dispersion=gpurun_out/r05_disp kaiser=...
    sys.path.insert(0, ROOT)
    from victor_amd.build import sources_digest
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    except OSError:
        commit = ""
    out = {"commit": commit or os.environ.get("VICTOR_COMMIT", ""), "source": label,
           "sources_sha256": sources_digest(),          # csrc/*.h, csrc/*.hip, include/victor_hip.h as profiled (bench.py compares)
           "model_options": {name: clock_only(d) for name, d in models.items()},
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB -> bytes; FETCH_SIZE x 2 on gfx950 "
                     "(MI355X_MICROARCH.md); per launch of the theory kernel",
           "config3": one(bench_dir, 65536, 120, "96 B parameter row in, lnL + chi2 out per evaluation; the theory workspace "
                          "(960 B per evaluation) makes a round trip when the chi-square is a launch of its own"),
           "boss_cmass": one(boss_dir, 65536, 60, "96 B parameter row in, lnL + chi2 out per evaluation; fused launch: no theory "
                             "workspace traffic; the per-point blended precision and the beta-polynomial tables are read from L2")}
    with open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
