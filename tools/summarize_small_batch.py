"""K1 / K2 / gap split of a rocprofv3 --kernel-trace of tools/small_batch_loop.py.  Usage: summarize_small_batch.py DIR"""
import csv
import glob
import json
import os
import sys

out = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)):
    with open(f) as fh:
        rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
    rows = [r for r in rows if r["Kernel_Name"].startswith("vk_") or "vk::" in r["Kernel_Name"]]
    rows = rows[len(rows) // 4:]                                # skip warm-up and first-touch launches
    k1 = [r for r in rows if "theory" in r["Kernel_Name"]]
    k2 = [r for r in rows if "like" in r["Kernel_Name"]]
    dur = lambda rs: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / max(len(rs), 1) / 1e3  # noqa: E731
    gap12, gap21 = [], []
    for a, b in zip(rows[:-1], rows[1:]):
        g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        if "theory" in a["Kernel_Name"] and "like" in b["Kernel_Name"]:
            gap12.append(g)
        elif "like" in a["Kernel_Name"] and "theory" in b["Kernel_Name"]:
            gap21.append(g)
    med = lambda v: sorted(v)[len(v) // 2] if v else None      # noqa: E731
    tag = os.path.relpath(f, sys.argv[1]).split(os.sep)[0]
    out[tag] = {"k1_name": k1[0]["Kernel_Name"][:60] if k1 else None, "k1_us": dur(k1), "k2_us": dur(k2), "n": len(k1),
                "gap_k1_to_k2_us_median": med(gap12), "gap_k2_to_next_k1_us_median": med(gap21),
                "grid_k1": k1[0].get("Grid_Size_X") if k1 else None, "lds_k1": k1[0].get("LDS_Block_Size") if k1 else None,
                "vgpr_k1": k1[0].get("VGPR_Count") if k1 else None, "grid_k2": k2[0].get("Grid_Size_X") if k2 else None}
    print(tag, json.dumps(out[tag]))
json.dump(out, open(os.path.join(sys.argv[1], "small_batch_summary.json"), "w"), indent=1)
