"""Collect per-kernel counter averages from the rocprofv3 CSV passes written by tools/pmc.sh."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# PMC summary (per-dispatch averages)\n")
for k, ctrs in agg.items():
    if "madicp" not in k:
        continue
    print(f"## `{k}`\n")
    print("| counter | dispatches | avg per dispatch |")
    print("|---|---|---|")
    for c, v in sorted(ctrs.items()):
        print(f"| {c} | {len(v)} | {sum(v) / len(v):.1f} |")
    print()
