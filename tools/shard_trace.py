"""Kernel timeline of tools/shard_probe.py under rocprofv3 --kernel-trace (development tool): for every phase of the probe
(separated by its long host-side pauses) the kernels per registration, their average duration and the average idle gap in
front of each.  usage: python tools/shard_trace.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("icp_", "ccl", "Kernel", "AllReduce", "copy", "fill"))]
# registrations end with icp_final; look at windows of 40 registrations around the middle of each third of the trace
fin = [i for i, r in enumerate(rows) if "icp_final" in r["Kernel_Name"]]
print("kernels %d, icp_final launches %d" % (len(rows), len(fin)))
def name(r):
    n = r["Kernel_Name"].split("(")[0]
    return n[-40:]
def report(a, b, title):
    seg = rows[a:b]
    stat = collections.OrderedDict()
    prev_end = int(seg[0]["Start_Timestamp"])
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        k = stat.setdefault(name(r), [0, 0.0, 0.0])
        k[0] += 1
        k[1] += (e - s) / 1e3
        k[2] += max(0, s - prev_end) / 1e3
        prev_end = max(prev_end, e)
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    nreg = sum(1 for r in seg if "icp_final" in r["Kernel_Name"])
    print("\n== %s: %d registrations, %.1f us per registration (span)" % (title, nreg, span / max(nreg, 1)))
    for n_, (c, d, g) in stat.items():
        print("   %-42s x%5d  avg %7.2f us  avg gap before %6.2f us" % (n_, c, d / c, g / c))
n = len(fin)
for frac, title in ((0.05, "early (fused B=1)"), (0.5, "middle"), (0.93, "late")):
    i = int(n * frac)
    report(fin[max(i - 20, 0)] + 1, fin[min(i + 20, n - 1)] + 1, title)
