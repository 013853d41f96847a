"""Pipeline.compute per-frame host time on a short synthetic drive (GPU box): median frame, lastBuildMs, lastIcpMs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_icp_amd import _build, synth  # noqa: E402

_build.build_pybind()
from mad_icp.src.pybind import pypeline as pm  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene = synth.Scene(0)
drive = [synth.render_scan(scene, synth.path_pose(1.0 * i), 100 + i) for i in range(14)]
pl = pm.Pipeline(10.0, False, 0.2, 0.1, 0.8, 0.1, 0.02, 16, threads, False)
ts, bs, ics = [], [], []
for rep in range(3):
    for i, sc in enumerate(drive):
        v = pm.VectorEigen3d(sc)
        t = time.perf_counter()
        pl.compute(0.1 * (rep * 14 + i), v)
        ts.append(time.perf_counter() - t)
        bs.append(pl.lastBuildMs())
        ics.append(pl.lastIcpMs())
print("threads %d: frame median %.3f ms (min %.3f)  build+upload median %.3f  icp median %.3f" % (
    threads, 1e3 * np.median(ts[14:]), 1e3 * min(ts[14:]), np.median(bs[14:]), np.median(ics[14:])))
