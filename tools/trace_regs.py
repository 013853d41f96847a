"""Registration timeline from a rocprofv3 kernel trace of bench.py (development tool): per registration the gap before
its first round, its duration, and what ran on other queues meanwhile.  usage: trace_regs.py <kernel_trace.csv> [memcopy.csv]"""
import csv, sys
import numpy as np
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
fin = [i for i, r in enumerate(rows) if "icp_final" in r["Kernel_Name"]]
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
recs = []
for a, b in zip(fin[:-1], fin[1:]):
    seg = rows[a + 1:b + 1]
    rounds = [r for r in seg if "icp_round" in r["Kernel_Name"]]
    if len(rounds) != 15:
        continue
    others = [r["Kernel_Name"].split("(")[0][-20:] for r in seg if "icp_" not in r["Kernel_Name"]]
    recs.append(((int(rounds[0]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3,
                 (int(rows[b]["End_Timestamp"]) - int(rounds[0]["Start_Timestamp"])) / 1e3, dur(rows[b]),
                 [round(dur(r), 1) for r in rounds], others))
g = np.array([r[0] for r in recs]); d = np.array([r[1] for r in recs]); f = np.array([r[2] for r in recs])
n = len(recs)
for lo, hi in ((0, n // 4), (n // 4, n // 2), (n // 2, 3 * n // 4), (3 * n // 4, n)):
    print("regs %4d-%4d: gap median %.1f  reg median %.1f  final %.1f  period %.1f | others %s" % (
        lo, hi, np.median(g[lo:hi]), np.median(d[lo:hi]), np.median(f[lo:hi]), np.median(g[lo:hi] + d[lo:hi]), recs[(lo + hi) // 2][4][:4]))
print("sample rounds:", recs[n // 8][3])
if len(sys.argv) > 2:
    mc = list(csv.DictReader(open(sys.argv[2])))
    print("memcopies:", len(mc), mc[0].keys() if mc else "")
    k = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Direction", "")) for r in mc]
    import collections
    by = collections.defaultdict(list)
    for t, dd in k: by[dd].append(t)
    for dd, v in by.items(): print(dd, len(v), "median us", np.median(v))
