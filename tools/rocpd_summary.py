"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a small markdown table for profiles/.

  python tools/rocpd_summary.py gpurun_out/prof/xxx_results.db > profiles/r1_xxx_kernel_stats.md
"""
import sqlite3
import sys


def main(path, title):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                          "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                          "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1
    print(f"# {title}\n")
    print("rocprofv3 --kernel-trace --stats (durations in ns, from the dispatch timestamps)\n")
    print("| kernel | calls | total ns | avg ns | min ns | max ns | % | vgpr | sgpr | lds B | scratch B | grid_x | wg_x |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| `{r[0]}` | {r[1]} | {r[2]} | {r[3]:.0f} | {r[4]} | {r[5]} | {100.0 * r[2] / total:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
