"""Timeline of one parallel host tree build (MADICP_HOST_TRACE): where the wall time goes.  GPU box host or anywhere."""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(tempfile.gettempdir(), "madicp_host_trace.txt")
os.environ["MADICP_HOST_TRACE"] = path
from mad_icp_amd import capi, synth  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
capi.host_lib().madicp_host_set_threads(threads)
scan = synth.render_scan(synth.Scene(0), synth.path_pose(0.0), 1)
import time
import numpy as np
scans = [synth.render_scan(synth.Scene(0), synth.path_pose(1.0 * i), 1 + i) for i in range(4)]
ts = []
for i in range(44):
    t = time.perf_counter(); capi.HostTree(scans[i % 4], 0.2, 0.1, lvl); ts.append(time.perf_counter() - t)
dt = ts[-1]
print("builds: median %.2f ms  min %.2f ms  (last, traced: %.2f ms)" % (1e3 * np.median(ts[4:]), 1e3 * min(ts[4:]), 1e3 * dt))
ev = [l.split() for l in open(path)]
ev = [(k, int(l), int(n), float(a), float(b), t) for k, l, n, a, b, t in ev]
end = max(e[4] for e in ev)
print("timeline end %.0f us; threads seen %d" % (end, len(set(e[5] for e in ev))))
print("forked nodes (critical path candidates):")
for k, l, n, a, b, t in sorted([e for e in ev if e[0] == "N"], key=lambda e: e[3])[:40]:
    print("  N level %d  n %6d  %7.0f -> %7.0f  (%5.0f us)  thread %s" % (l, n, a, b, b - a, t))
seq = [e for e in ev if e[0] == "S"]
print("sequential chunks: %d, total %.0f us of work, longest %.0f us (n %d), last ends %.0f us" % (
    len(seq), sum(e[4] - e[3] for e in seq), max(e[4] - e[3] for e in seq), max(seq, key=lambda e: e[4] - e[3])[2], max(e[4] for e in seq)))
late = sorted(seq, key=lambda e: -e[4])[:8]
for k, l, n, a, b, t in late:
    print("  S level %d  n %6d  %7.0f -> %7.0f  (%5.0f us) thread %s" % (l, n, a, b, b - a, t))
for k, l, n, a, b, t in [e for e in ev if e[0] == "C"]:
    print("  copy/layout %7.0f -> %7.0f (%.0f us)" % (a, b, b - a))
for k, l, n, a, b, t in [e for e in ev if e[0] in "LP"]:
    print("  %s n %6d %7.0f -> %7.0f (%.0f us) thread %s" % (k, n, a, b, b - a, t))
