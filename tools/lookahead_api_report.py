"""Merged host / device timeline of the look-ahead drive from the traces of tools/lookahead_api_trace.sh.

  python tools/lookahead_api_report.py DIR [frame ...]

DIR holds t_kernel_trace.csv, t_hip_api_trace.csv (and t_memory_copy_trace.csv) somewhere below it.  For the chosen steady
frames (a frame = from one icp_final to the next) it prints, in time order: the device's busy segments (kernels merged when
less than 3 us apart, with the families inside), the copies, and every HIP call of at least 10 us with its thread; then a
table over ALL steady frames: host time per HIP function per frame, and the device's idle time."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    g = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return g[0] if g else None


def col(row, *names):
    for n in names:
        if n in row:
            return row[n]
    raise KeyError(names)


def fam(n):
    for f in ("icp_round", "icp_final", "icp_publish", "tb_chip_stats", "tb_chip_scatter", "tb_level", "tb_init", "tb_finish",
              "tb_emit", "tree_compact", "moving_from_leaves", "tree_transform", "deskew", "ingest"):
        if f in n:
            return f
    return n.split("(")[0][-24:]


def main(d, frames):
    kern = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam(r["Kernel_Name"]), r.get("Queue_Id", "?"))
            for r in csv.DictReader(open(find(d, "*kernel_trace.csv")))]
    api_path = find(d, "*hip_api_trace.csv")
    api = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), col(r, "Function", "Name"), r.get("Thread_Id", "?"))
           for r in csv.DictReader(open(api_path))] if api_path else []
    cp_path = find(d, "*memory_copy_trace.csv")
    cps = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), col(r, "Direction", "Name"))
           for r in csv.DictReader(open(cp_path))] if cp_path else []
    finals = sorted(s for s, e, f, q in kern if f == "icp_final")
    if len(finals) < 10:
        print("too few registrations")
        return
    print("frame periods (us):", " ".join("%.0f" % ((b - a) / 1e3) for a, b in zip(finals[3:-1], finals[4:])))
    threads = sorted({t for _, _, _, t in api})
    tname = {t: "T%d" % i for i, t in enumerate(threads)}
    for fr in frames:
        t0, t1 = finals[fr], finals[fr + 1]
        print("\n== frame %d: %.0f us ==" % (fr, (t1 - t0) / 1e3))
        ev = []
        ks = sorted(k for k in kern if k[1] > t0 and k[0] < t1)
        seg = None
        for s, e, f, q in ks:
            if seg and s - seg[1] < 3000:
                seg[1] = max(seg[1], e)
                seg[2][f + "@" + q] += 1
            else:
                if seg:
                    ev.append((seg[0], "GPU  busy %7.1f us  %s" % ((seg[1] - seg[0]) / 1e3, " ".join("%s x%d" % kv for kv in seg[2].items()))))
                seg = [s, e, defaultdict(int)]
                seg[2][f + "@" + q] += 1
        if seg:
            ev.append((seg[0], "GPU  busy %7.1f us  %s" % ((seg[1] - seg[0]) / 1e3, " ".join("%s x%d" % kv for kv in seg[2].items()))))
        for s, e, n in cps:
            if e > t0 and s < t1:
                ev.append((s, "COPY      %7.1f us  %s" % ((e - s) / 1e3, n)))
        for s, e, n, t in api:
            if e > t0 and s < t1 and e - s >= 10000:
                ev.append((s, "HOST %s %7.1f us  %s" % (tname[t], (e - s) / 1e3, n)))
        for s, txt in sorted(ev):
            print("  %+8.1f  %s" % ((s - t0) / 1e3, txt))
    # totals over the steady frames
    t0, t1 = finals[3], finals[-1]
    n = len(finals) - 4
    tot = defaultdict(lambda: [0, 0.0])
    for s, e, f, t in api:
        if s >= t0 and e <= t1:
            tot[(f, tname[t])][0] += 1
            tot[(f, tname[t])][1] += (e - s) / 1e3
    print("\n| HIP call | thread | calls per frame | us per frame |\n|---|---|---|---|")
    for (f, t), (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
        print("| %s | %s | %.1f | %.1f |" % (f, t, c / n, us / n))


if __name__ == "__main__":
    main(sys.argv[1], [int(a) for a in sys.argv[2:]] or [10, 11])
