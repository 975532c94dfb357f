"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) of tools/gpu_profile.sh into text + JSON."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir = sys.argv[1]
summary = {}
# --points N: integrand points (or cells) one launch of the theory kernel processes -> instructions per point; --label TEXT
points = None
label = None
for i, a in enumerate(sys.argv):
    if a == "--points":
        points = float(sys.argv[i + 1])
    if a == "--label":
        label = sys.argv[i + 1]
if label:
    print("==", label)
    summary["label"] = label
for f in glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out_dir))
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows:
        print("  {Name:60.60s} calls={Calls} total_ns={TotalDurationNs} avg_ns={AverageNs} pct={Percentage}".format(**r))
    summary["kernel_stats"] = rows
for f in glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    by = defaultdict(list)
    meta = {}
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        by[r["Kernel_Name"]].append(d)
        meta[r["Kernel_Name"]] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                                       "Scratch_Size", "Grid_Size_X", "Workgroup_Size_X")}
    print("== kernel trace (per-dispatch durations, ns)")
    for k, v in by.items():
        print(f"  {k[:60]:60s} n={len(v)} avg={sum(v)/len(v):.0f} min={min(v)} max={max(v)} {meta[k]}")
    summary["kernel_trace"] = {k: {"n": len(v), "avg_ns": sum(v) / len(v), "min_ns": min(v), "max_ns": max(v), **meta[k]}
                               for k, v in by.items()}
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== PMC (average per dispatch)")
summary["pmc"] = {}
for k, cs in pmc.items():
    summary["pmc"][k] = {}
    for c, v in sorted(cs.items()):
        avg = sum(v) / len(v)
        summary["pmc"][k][c] = {"avg": avg, "n": len(v)}
        print(f"  {k[:50]:50s} {c:28s} avg={avg:.6g} n={len(v)}")
# HBM traffic of the theory kernel per launch
for k, cs in summary["pmc"].items():
    if "vk_theory" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        # MI355X_MICROARCH.md (HBM): counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes read -> x2
        traffic = (2.0 * cs["FETCH_SIZE"]["avg"] + cs["WRITE_SIZE"]["avg"]) * 1024.0
        summary["theory_kernel_hbm_bytes_per_launch"] = traffic
        print(f"== theory kernel HBM traffic per launch: {traffic/1e6:.2f} MB "
              f"(fetch 2 x {cs['FETCH_SIZE']['avg']*1024/1e6:.2f} MB, write {cs['WRITE_SIZE']['avg']*1024/1e6:.2f} MB)")
# instruction counts per integrand point (per lane): SQ_INSTS_* count wave-level instructions, one per 64 lanes
if points:
    for k, cs in summary["pmc"].items():
        if "vk_theory" not in k:
            continue
        per = {c: cs[c]["avg"] * 64.0 / points for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD") if c in cs}
        busy = {}
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs' clocks
        if "SQ_ACTIVE_INST_VALU" in cs and "GRBM_GUI_ACTIVE" in cs:
            cycles = cs["GRBM_GUI_ACTIVE"]["avg"] / 8.0
            busy["valu_busy"] = 4.0 * cs["SQ_ACTIVE_INST_VALU"]["avg"] / (1024.0 * cycles)
            if "SQ_ACTIVE_INST_LDS" in cs:
                busy["lds_busy"] = 4.0 * cs["SQ_ACTIVE_INST_LDS"]["avg"] / (256.0 * cycles)
        if "SQ_LDS_BANK_CONFLICT" in cs and "SQ_LDS_IDX_ACTIVE" in cs and cs["SQ_LDS_IDX_ACTIVE"]["avg"] > 0:
            busy["lds_bank_conflict_frac"] = cs["SQ_LDS_BANK_CONFLICT"]["avg"] / cs["SQ_LDS_IDX_ACTIVE"]["avg"]
        summary.setdefault("per_point", {})[k] = {"points_per_launch": points, **per, **busy}
        print(f"== {k[:70]}: per point of {points:.4g} per launch: " + ", ".join(f"{c[8:]} {v:.1f}" for c, v in per.items())
              + ("; " + ", ".join(f"{c} {v:.3f}" for c, v in busy.items()) if busy else ""))
with open(os.path.join(out_dir, "summary.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
