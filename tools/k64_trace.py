"""Per-round kernel durations of BASELINE configs[4] (64 keyframes, 8 scans in flight), from a rocprofv3 kernel trace of
tools/k64_probe.py:

  K64_ONLY8=1 rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/k64_probe.py
  python tools/k64_trace.py DIR/**/t_kernel_trace.csv

Every icp_round launch of that run has the 8-scan geometry and a batch is 15 rounds + one icp_final, so launch `position %
15` is round `position` of a batch.  Which rounds the time goes to decides what is worth changing in the kernel: rounds 0-1
(every pair walks), rounds 2-8 (a shrinking fraction walks, but nearly every wavefront holds a walker), rounds 9-14 (nobody
walks)."""
import csv
import statistics
import sys

ROUNDS = 15


def main(path):
    rows = list(csv.DictReader(open(path)))
    rnd = sorted([r for r in rows if "icp_round" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
    fin = sorted([r for r in rows if "icp_final" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rnd]
    n = len(d) // ROUNDS
    skip = min(3, n // 4)
    d = d[skip * ROUNDS: n * ROUNDS]
    print("batches in the trace: %d (first %d skipped); icp_round launches %d, avg %.1f us; icp_final avg %.1f us"
          % (n, skip, len(d), statistics.mean(d), statistics.mean((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in fin)))
    print("| round | " + " | ".join(str(i) for i in range(ROUNDS)) + " | sum |")
    print("|" + "---|" * (ROUNDS + 2))
    per = [statistics.mean(d[i::ROUNDS]) for i in range(ROUNDS)]
    print("| avg us | " + " | ".join("%.1f" % x for x in per) + " | %.0f |" % sum(per))


if __name__ == "__main__":
    main(sys.argv[1])
