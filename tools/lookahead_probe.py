"""Frame period of Pipeline.compute against the look-ahead depth (prefetch(i + d) before compute(i)), host path and device
front-end.
Usage: python tools/lookahead_probe.py [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_icp_amd import _build, synth  # noqa: E402

_build.build_pybind()
from mad_icp.src.pybind import pypeline  # noqa: E402

if os.environ.get("LOOKAHEAD_IMPORT_TORCH") == "1":  # (is it the framework's presence in the process that makes the look-ahead fast?)
    import torch

    torch.cuda.init()
    torch.zeros(8, device="cuda").sum().item()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
scene = synth.Scene(0)
drive = [synth.render_scan(scene, synth.path_pose(1.0 * i), 100 + i) for i in range(N)]
clouds = [pypeline.VectorEigen3d(s) for s in drive]
threads = min(os.cpu_count() or 1, 16)
ref = None
CASES = ((False, 0), (False, 1), (False, 2), (False, 3), (True, 0), (True, 1))
if os.environ.get("LOOKAHEAD_ONLY") == "device":
    CASES = ((True, 0), (True, 1))
for device, depth in CASES:
    if device and depth == 0:
        ref = None  # (device-built trees: their own reference trajectory)
    pl = pypeline.Pipeline(10.0, False, 0.2, 0.1, 0.8, 0.1, 0.02, 16, threads, False)
    pl.setDeviceFrontEnd(device)
    for d in range(depth):
        pl.prefetch(clouds[d])
    ts = []
    for i in range(N):
        t = time.perf_counter()
        if depth and i + depth < N:
            pl.prefetch(clouds[i + depth])
        pl.compute(0.1 * i, clouds[i])
        ts.append(time.perf_counter() - t)
    traj = np.asarray(pl.trajectory())
    if ref is None:
        ref = traj
    assert np.array_equal(traj, ref)
    if os.environ.get("LOOKAHEAD_SERIES") == "1":
        print("   ", " ".join("%.2f" % (1e3 * t) for t in ts))
    print("%s, look-ahead %d: mean %.2f ms per frame = %.0f frames/s (median %.2f: the series is bimodal with a look-ahead)"
          % ("device front-end" if device else "host path", depth, 1e3 * np.mean(ts[3:N - 3]), 1.0 / np.mean(ts[3:N - 3]),
             1e3 * np.median(ts[3:N - 3])))
