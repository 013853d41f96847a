"""Registration span per frame from a kernel trace of the look-ahead drive: first icp_round start -> icp_final end, the rounds'
durations, the gaps between consecutive rounds, and what else ran inside the span.
  python tools/registration_span.py DIR/**/t_kernel_trace.csv"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
k = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
is_round = lambda n: "icp_round" in n  # noqa: E731
finals = [x for x in k if "icp_final" in x[2]]
spans, slow, gaps, inside = [], 0, [], Counter()
n_rounds = 0
prev_final_end = None
for f in finals[4:]:
    rs = [x for x in k if is_round(x[2]) and x[1] <= f[0] + 1000 and (prev_final_end is None or x[0] >= prev_final_end)]
    rs = rs[-15:]
    prev_final_end = f[1]
    if len(rs) < 15:
        continue
    spans.append((f[1] - rs[0][0]) / 1e3)
    for a, b in zip(rs[:-1], rs[1:]):
        gaps.append((b[0] - a[1]) / 1e3)
    for r in rs:
        n_rounds += 1
        if r[1] - r[0] > 20000:
            slow += 1
    for x in k:
        if x[0] < f[1] and x[1] > rs[0][0] and not is_round(x[2]) and "icp_final" not in x[2]:
            inside[x[2].split("(")[0][-26:] + "@" + x[3]] += 1
spans.sort()
gaps.sort()
print("frames %d: registration span (first round -> final) median %.1f us, p90 %.1f, max %.1f; rounds slower than 20 us: %d of %d; "
      "gap between rounds median %.2f us, p99 %.2f, sum per frame %.1f"
      % (len(spans), spans[len(spans) // 2], spans[int(len(spans) * 0.9)], spans[-1], slow, n_rounds, gaps[len(gaps) // 2],
         gaps[int(len(gaps) * 0.99)], sum(gaps) / max(1, len(spans))))
print("kernels inside the span (per frame):", ", ".join("%s x%.1f" % (n, c / len(spans)) for n, c in inside.most_common(10)))
