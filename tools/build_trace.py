"""Per-kernel timeline of ONE device tree build from a rocprofv3 kernel trace of tools/build_probe.py (development tool).

  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/build_probe.py 5
  python tools/build_trace.py <dir>/**/*kernel_trace.csv > profiles/<tag>_tree_build_trace.txt

Prints, for the LAST build of the trace, every kernel with its workgroups, start, duration and the gap to the previous
kernel (microseconds), then the totals per kernel name."""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "tb_init" in r["Kernel_Name"]]
if not starts:
    sys.exit("no tb_init in the trace")
seg = rows[starts[-1]:]
# the build ends with the emission of the tree (tb_emit: node array, screening / leaf records and the staged top in one launch)
end = max((i for i, r in enumerate(seg) if "tb_emit" in r["Kernel_Name"]), default=len(seg) - 1)
seg = seg[: end + 1]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
tot = collections.OrderedDict()
print("# one madicp_tree_build; start / duration / gap to the previous kernel in microseconds (traced run)")
for r in seg:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wgs = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1))
    print("%-24s wgs=%5d start=%7.1f dur=%6.1f gap=%5.1f" % (name[:24], wgs, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
    k = tot.setdefault(name, [0, 0.0])
    k[0] += 1
    k[1] += (e - s) / 1e3
print("# total %.1f us from the first kernel's start to the last one's end" % ((prev_end - t0) / 1e3))
for name, (n, us) in tot.items():
    print("#   %-24s x%3d  %7.1f us" % (name[:24], n, us))
