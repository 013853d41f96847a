"""Per-registration kernel trace of the streamed headline workload — the rocprofv3 evidence `roofline.avg_launch_us` of
bench.py must agree with.

  rocprofv3 --kernel-trace --stats --output-format csv -d DIR -o t -- python tools/round_trace.py run [N]
  python tools/round_trace.py split DIR/**/t_kernel_trace.csv

`run` does ONLY what bench.py's timed region does: N (default 200) streamed single-scan registrations of BASELINE configs[2]
(8 distinct scans cycled, one submission ahead of the collection), so every icp_round launch in the trace is round
`position % 15` of such a registration.  `split` prints the average duration per round position, over all positions (=
what bench.py reports live as roofline.avg_launch_us, measured there with HIP events around a graph of the same
registration), and icp_final's."""
import csv
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROUNDS = 15


def run(n):
    from mad_icp_amd import capi, synth

    K = 16
    pb = synth.make_problem(K, seed=1, n_queries=1)
    scans, gts, gs = synth.make_query_streams(K, seed=1, n_streams=8)
    ctx = capi.Context(0)
    for kv in os.environ.get("MADICP_OPTIONS", "").split(","):  # development: library options for an A/B trace, "key=value,..."
        if "=" in kv:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
    tids = []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        ht = capi.HostTree(s, 0.2, 0.1, 3)
        ht.transform(T[:3, :3], T[:3, 3])
        tids.append(ctx.upload(ht))
    qts = [capi.HostTree(s, 0.2, 0.1, 3) for s in scans]
    leaves = [q.leaf_means() for q in qts]
    guess = [capi.pose12(T) for T in gs]
    prev = None
    for i in range(n):
        q = i % 8
        tk = ctx.stream_submit(leaves[q], tids, guess[q], (0.2, 0.1, 0.02), ROUNDS)
        if prev is not None:
            ctx.stream_collect(prev[0], prev[1])
        prev = (tk, qts[q].num_leaves)
    ctx.stream_collect(prev[0], prev[1])
    ctx.close()


def split(path):
    rows = list(csv.DictReader(open(path)))

    def dur(r):
        return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3

    rnd = sorted([r for r in rows if "icp_round" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
    fin = sorted([r for r in rows if "icp_final" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
    n_reg = len(rnd) // ROUNDS
    skip = min(20, n_reg // 4)  # warm-up registrations
    d = [dur(r) for r in rnd[skip * ROUNDS: n_reg * ROUNDS]]
    e = [dur(r) for r in fin[skip:n_reg]]
    gaps = []
    for i in range(skip * ROUNDS, n_reg * ROUNDS - 1):
        if (i + 1) % ROUNDS:
            gaps.append((int(rnd[i + 1]["Start_Timestamp"]) - int(rnd[i]["End_Timestamp"])) / 1e3)
    print("streamed single-scan registrations in the trace: %d (first %d skipped as warm-up)" % (n_reg, skip))
    print("| kernel | launches | avg us | median us |\n|---|---|---|---|")
    print("| icp_round, all rounds of a registration | %d | %.2f | %.2f |" % (len(d), statistics.mean(d), statistics.median(d)))
    print("| icp_final | %d | %.2f | %.2f |" % (len(e), statistics.mean(e), statistics.median(e)))
    print("| idle between two rounds of a registration (end -> next start) | %d | %.2f | %.2f |" % (len(gaps), statistics.mean(gaps), statistics.median(gaps)))
    per_reg = [sum(d[i:i + ROUNDS]) for i in range(0, len(d), ROUNDS)]
    print("| sum of a registration's 15 rounds | %d | %.1f | %.1f |" % (len(per_reg), statistics.mean(per_reg), statistics.median(per_reg)))
    print()
    print("| round | " + " | ".join(str(i) for i in range(ROUNDS)) + " |")
    print("|" + "---|" * (ROUNDS + 1))
    print("| avg us | " + " | ".join("%.1f" % statistics.mean(d[i::ROUNDS]) for i in range(ROUNDS)) + " |")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "split":
        split(sys.argv[2])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 200)
