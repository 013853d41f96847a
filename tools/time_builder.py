"""Host MAD-tree build time of 120k-point synthetic scans vs max_parallel_level, the flag-driven partition against the
reference's swap loop (MADICP_HOST_PARTITION=loop), after a warm-up; median and min of 15 runs each."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np

    from mad_icp_amd import capi, synth

    scans = [synth.render_scan(synth.Scene(0), synth.path_pose(1.0 * i), 1 + i) for i in range(3)]
    lim = int(os.environ.get("MADICP_POOL_LIMIT", "16"))
    capi.host_lib().madicp_host_set_threads(lim)
    for _ in range(30):
        capi.HostTree(scans[0], 0.2, 0.1, 4)
    for lvl in (0, 2, 3, 4, 5):
        ts = []
        for k in range(15):
            t = time.perf_counter()
            h = capi.HostTree(scans[k % 3], 0.2, 0.1, lvl)
            ts.append(time.perf_counter() - t)
        print("  max_parallel_level %d: median %.2f ms  min %.2f ms  (%d nodes)" % (lvl, 1e3 * np.median(ts), 1e3 * min(ts), h.num_nodes))
    sys.exit(0)

print("cpus", os.cpu_count())
for how, slice_min, aff in (("loop", None, "none"), ("flags", None, "none"), ("flags", None, "node"), ("flags", None, "l3"),
                            ("flags", 32768, "node"), ("flags", 32768, "l3")):
    print("partition =", how, " bbox slice min =", slice_min, " worker affinity =", aff)
    sys.stdout.flush()
    env = dict(os.environ, MADICP_HOST_PARTITION=how, MADICP_HOST_AFFINITY=aff)
    if slice_min:
        env["MADICP_HOST_BBOX_SLICE_MIN"] = str(slice_min)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child"], env=env)
