"""How much of the look-ahead tree construction really runs BESIDE the registration: overlap of the builder's kernels (tb_*,
build stream) with the registration's (icp_*, compute stream) in a rocprofv3 kernel trace of a Pipeline drive with the
device front-end on and prefetch(i + 1) issued before compute(i).

  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/overlap_trace.py run [frames]
  python tools/overlap_trace.py report DIR/**/t_kernel_trace.csv

`report` prints, per frame (steady frames only) and in total: the wall time covered by builder kernels, by registration
kernels, by both at once, and the device's busy time — so `both / min(build, registration)` is the fraction of the shorter
activity that was hidden behind the longer one."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n):
    from mad_icp_amd import _build, synth

    _build.build_pybind()
    from mad_icp.src.pybind import pypeline

    scene = synth.Scene(0)
    drive = [synth.render_scan(scene, synth.path_pose(1.0 * i), 100 + i) for i in range(n)]
    pl = pypeline.Pipeline(10.0, False, 0.2, 0.1, 0.8, 0.1, 0.02, 16, 16, False)
    pl.setDeviceFrontEnd(True)
    pl.prefetch(drive[0])
    for i in range(n):
        if i + 1 < n:
            pl.prefetch(drive[i + 1])
        pl.compute(0.1 * i, drive[i])


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    out = []
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def report(path):
    rows = list(csv.DictReader(open(path)))
    k = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    finals = sorted(s for n, s, e in k if "icp_final" in n)
    if len(finals) < 8:
        print("too few registrations in the trace")
        return
    # steady part: from the 4th registration's end to the last one's
    t0, t1 = finals[3], finals[-1]
    build = union([(s, e) for n, s, e in k if n.startswith("tb_") or "tb::" in n or "compact" in n or "tb_" in n])
    reg = union([(s, e) for n, s, e in k if "icp_" in n or "moving_from_leaves" in n or "tree_transform" in n])
    clip = lambda iv: [[max(a, t0), min(b, t1)] for a, b in iv if min(b, t1) > max(a, t0)]  # noqa: E731
    build, reg = clip(build), clip(reg)
    both = intersect(build, reg)
    busy = union(build + reg)
    frames = len(finals) - 4
    span = (t1 - t0) / 1e3
    print("| steady frames | frame period (device clock) | builder kernels | registration kernels | both at once | device busy | idle |")
    print("|---|---|---|---|---|---|---|")
    print("| %d | %.1f us | %.1f us | %.1f us | %.1f us | %.1f us | %.1f us |"
          % (frames, span / frames, length(build) / 1e3 / frames, length(reg) / 1e3 / frames, length(both) / 1e3 / frames,
             length(busy) / 1e3 / frames, (span - length(busy) / 1e3) / frames))
    print("\nper frame; `both at once` = %.0f %% of the registration's kernel time ran while a builder kernel was running"
          % (100.0 * length(both) / max(1, length(reg))))
    # where the kernel time goes, family by family (launches and microseconds per steady frame), and on which queues
    fam = {}
    for r in rows:
        n, s_, e_ = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s_ < t0 or e_ > t1:
            continue
        key = next((f for f in ("icp_round", "icp_final", "icp_publish", "tb_chip_stats", "tb_chip_scatter", "tb_level", "tb_init",
                                "tb_finish", "tb_emit", "tree_compact", "moving_from_leaves", "tree_transform") if f in n), "other")
        d = fam.setdefault(key, [0, 0.0, set()])
        d[0] += 1
        d[1] += (e_ - s_) / 1e3
        d[2].add(r.get("Queue_Id", "?"))
    print("\n| kernels | launches per frame | us per frame | avg us | queues |\n|---|---|---|---|---|")
    for key, (c, us, qs) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("| %s | %.1f | %.1f | %.2f | %s |" % (key, c / frames, us / frames, us / c, ",".join(sorted(qs))))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    elif len(sys.argv) >= 3 and sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        print(__doc__)
