"""Where a look-ahead frame of the default Pipeline spends its time, phase by phase, on the host's clock (no profiler attached):
prefetch() call | compute(): tree ready (collect the construction: wait + size + emit) | registration submitted | look-ahead begun
(staging copy + the construction's launches) | wait for the registration | the rest of compute() (velocity, transform, window).
Medians over the steady frames, device front-end, look-ahead 0 and 1.
Usage: [GPU_MAX_HW_QUEUES=8] python tools/lookahead_phases.py [frames]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_icp_amd import _build, synth  # noqa: E402

_build.build_pybind()
from mad_icp.src.pybind import pypeline  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
CANARY = os.environ.get("LOOKAHEAD_CANARY") == "1"
scene = synth.Scene(0)
drive = [synth.render_scan(scene, synth.path_pose(1.0 * i), 100 + i) for i in range(N)]
clouds = [pypeline.VectorEigen3d(s) for s in drive]
buf = np.empty((max(len(s) for s in drive), 3))
print("GPU_MAX_HW_QUEUES=%s MADICP_PUBLISH_SIDE=%s" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), os.environ.get("MADICP_PUBLISH_SIDE", "default")))
print("| look-ahead | frame | prefetch() | tree ready | submit | begin look-ahead | wait for the result | rest of compute() | series of frames (ms) |")
print("|---|---|---|---|---|---|---|---|---|")
for depth in (0, 1):
    pl = pypeline.Pipeline(10.0, False, 0.2, 0.1, 0.8, 0.1, 0.02, 16, 16, False)
    pl.setDeviceFrontEnd(True)
    if depth:
        pl.prefetch(clouds[0])
    rows = []
    canary = []
    for i in range(N):
        t0 = time.perf_counter()
        if depth and i + 1 < N:
            pl.prefetch(clouds[i + 1])
        t1 = time.perf_counter()
        t_pf = t1 - t0
        if CANARY and depth:  # (host work of the same kind as the binding's by-value cloud, while the construction runs on the GPU)
            c0 = time.perf_counter()
            v = pypeline.VectorEigen3d(clouds[i])  # allocate 1.4 MB + copy
            c1 = time.perf_counter()
            del v  # free
            c2 = time.perf_counter()
            np.copyto(buf[:len(drive[i])], drive[i])  # copy into memory that exists
            c3 = time.perf_counter()
            canary.append((1e3 * (c1 - c0), 1e3 * (c2 - c1), 1e3 * (c3 - c2)))
            t1 = time.perf_counter()
        pl.compute(0.1 * i, clouds[i])
        t2 = time.perf_counter()
        ph = pl.lastIcpPhasesMs()
        b = pl.lastBuildMs()
        rows.append((1e3 * (t2 - t1 + t_pf), 1e3 * t_pf, b, ph[0], ph[1], ph[2], 1e3 * (t2 - t1) - b - sum(ph)))
    a = np.asarray(rows[4:N - 2])
    med = np.median(a, axis=0)
    print("| %d | %.3f (mean %.3f) | %s | %s |" % (depth, med[0], a[:, 0].mean(), " | ".join("%.3f" % v for v in med[1:]),
                                                  " ".join("%.2f" % v for v in a[:14, 0])))
    if canary:
        c = np.median(np.asarray(canary[4:]), axis=0)
        print("canary between prefetch() and compute() (not in the frame time above): allocate + copy a cloud %.3f ms, free it %.3f ms, "
              "copy into an existing buffer %.3f ms" % tuple(c))
