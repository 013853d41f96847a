"""Split a rocprofv3 kernel trace of `bench.py` into the timed region and per-GN-round averages.

usage: python tools/trace_split.py <kernel_trace.csv> <registrations in the timed region (warmup + steps)> [rounds=15]
"""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_reg = int(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 15


def dur(r):
    return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3


rnd = sorted([r for r in rows if "icp_round" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
fin = sorted([r for r in rows if "icp_final" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
d = [dur(r) for r in rnd[: n_reg * rounds]]
e = [dur(r) for r in fin[:n_reg]]
print("| kernel | launches | avg us | median us |\n|---|---|---|---|")
print("| icp_round (timed region) | %d | %.2f | %.2f |" % (len(d), statistics.mean(d), statistics.median(d)))
print("| icp_final (timed region) | %d | %.2f | %.2f |" % (len(e), statistics.mean(e), statistics.median(e)))
print()
print("| round | " + " | ".join(str(i) for i in range(rounds)) + " |")
print("|" + "---|" * (rounds + 1))
print("| avg us | " + " | ".join("%.1f" % statistics.mean(d[i::rounds]) for i in range(rounds)) + " |")
