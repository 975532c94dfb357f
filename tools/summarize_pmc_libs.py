"""Condense the counter CSVs of tools/gpu_pmc_libs.sh: one column per library build, theory kernel only, averages per launch."""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "*_set*", "**", "*counter_collection.csv"), recursive=True):
    build = os.path.relpath(f, out).split(os.sep)[0].rsplit("_set", 1)[0]
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if "vk_theory" in r["Kernel_Name"]:
                vals[r["Counter_Name"]][build].append(float(r["Counter_Value"]))
builds = sorted({b for c in vals.values() for b in c})
print(f"{'counter (avg per launch)':32s}" + "".join(f"{b:>18s}" for b in builds))
for c in sorted(vals):
    print(f"{c:32s}" + "".join(f"{(sum(vals[c][b]) / len(vals[c][b]) if vals[c][b] else float('nan')):18.5g}" for b in builds))
