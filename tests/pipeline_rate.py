"""End-to-end Pipeline.compute rate on full-size synthetic scans: the product (pybind pypeline, GPU hot path) next to
the CPU oracle pipeline (the restated reference).  GPU box only.  usage: python tests/pipeline_rate.py [frames=24]
(Lives under tests/ because it runs the oracle, which only tests and bench.py's cpu_baseline leg may do.)

Every frame: MAD-tree build of the incoming 120k-point scan on the host (both sides), then the 15-round registration
against up to 16 keyframes (GPU: one device call; CPU: the reference's OpenMP loop), keyframe bookkeeping.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib as O  # noqa: E402  (checker/baseline only)
from mad_icp_amd import synth  # noqa: E402
from mad_icp.src.pybind import pypeline  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cpu_frames = min(n_frames, 10)
threads = min(os.cpu_count() or 1, 16)
scene = synth.Scene(0)
scans = [synth.render_scan(scene, synth.path_pose(1.0 * i), 100 + i) for i in range(n_frames)]
args = (10.0, False, 0.2, 0.1, 0.8, 0.1, 0.02, 16, threads, False)

gp = pypeline.Pipeline(*args)
t_gpu = []
for i, s in enumerate(scans):
    v = pypeline.VectorEigen3d(s)
    t = time.perf_counter()
    gp.compute(0.1 * i, v)
    t_gpu.append(time.perf_counter() - t)
op = O.Pipeline(*args)
t_cpu = []
for i, s in enumerate(scans[:cpu_frames]):
    t = time.perf_counter()
    op.compute(0.1 * i, s)
    t_cpu.append(time.perf_counter() - t)
d = np.linalg.inv(op.currentPose()) @ np.asarray(gp.trajectory()[cpu_frames - 1])
gt = np.linalg.inv(synth.path_pose(0.0)) @ synth.path_pose(1.0 * (n_frames - 1))
print("frames %d (CPU: %d), %d points/scan, %d host threads" % (n_frames, cpu_frames, scans[0].shape[0], threads))
print("product  Pipeline.compute: median %.2f ms/frame (%.0f frames/s); tree build %.2f ms, registration %.2f ms (last frame)"
      % (1e3 * np.median(t_gpu[2:]), 1.0 / np.median(t_gpu[2:]), gp.lastBuildMs(), gp.lastIcpMs()))
print("oracle   Pipeline.compute: median %.2f ms/frame (%.1f frames/s)" % (1e3 * np.median(t_cpu[2:]), 1.0 / np.median(t_cpu[2:])))
print("pose difference product vs oracle at frame %d: %.2e m;  product drift vs ground truth after %d m: %.3f m"
      % (cpu_frames - 1, np.linalg.norm(d[:3, 3]), n_frames - 1, np.linalg.norm(gp.currentPose()[:3, 3] - gt[:3, 3])))
