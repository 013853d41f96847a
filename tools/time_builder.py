"""Host MAD-tree build time of one 120k-point synthetic scan vs max_parallel_level (min of 7 runs)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_icp_amd import capi, synth  # noqa: E402

scan = synth.render_scan(synth.Scene(0), synth.path_pose(0.0), 1)
print("points", scan.shape[0], "cpus", os.cpu_count())
for lvl in (0, 1, 2, 3, 4, 5, 6):
    ts = []
    for _ in range(7):
        t = time.perf_counter()
        h = capi.HostTree(scan, 0.2, 0.1, lvl)
        ts.append(time.perf_counter() - t)
    print("max_parallel_level %d: %.2f ms  (%d nodes)" % (lvl, 1e3 * min(ts), h.num_nodes))
