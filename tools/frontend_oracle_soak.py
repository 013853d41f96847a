"""Soak behind tests/test_gpu_frontend_oracle.py: the DEFAULT Pipeline (device front-end for deskew = false) against the oracle
pipeline over many more frames than the suite affords — scenes x frames, every frame compared (pose, currentID, keyframeID,
isMapUpdated).  usage (GPU box): python tools/frontend_oracle_soak.py [scenes] [frames]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from mad_icp_amd import _build, synth  # noqa: E402

_build.build_pybind()
from mad_icp.src.pybind import pypeline  # noqa: E402

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(2025)
worst_all, frames_all, mism = 0.0, 0, 0
for sc in range(n_scenes):
    seed = 20 + sc
    step = float(rng.choice([0.5, 0.8, 1.0, 1.3, 1.7, 2.0]))
    p_th = float(rng.choice([0.8, 0.85, 0.9, 0.95]))
    kf = int(rng.choice([4, 8, 16]))
    scene = synth.Scene(seed)
    args = (10.0, False, 0.2, 0.1, p_th, 0.1, 0.02, kf, 16, False)
    gp, op = pypeline.Pipeline(*args), O.Pipeline(*args)
    assert gp.deviceFrontEnd()
    worst_t = worst_a = 0.0
    promotions = 0
    for i in range(n_frames):
        s = synth.render_scan(scene, synth.path_pose(step * i), 9000 + 131 * seed + i)
        gp.compute(0.1 * i, s)
        op.compute(0.1 * i, s)
        d = np.linalg.inv(op.currentPose()) @ np.asarray(gp.currentPose())
        worst_t = max(worst_t, float(np.linalg.norm(d[:3, 3])))
        worst_a = max(worst_a, float(np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1))))
        mism += int(gp.currentID() != op.currentID() or gp.keyframeID() != op.keyframeID() or gp.isMapUpdated() != op.isMapUpdated())
        promotions += int(gp.isMapUpdated())
    print("scene %2d: %3d frames x %d points, %.1f m/frame, p_th %.2f, %2d keyframes: worst %.2e m / %.2e rad, %d promotions"
          % (seed, n_frames, s.shape[0], step, p_th, kf, worst_t, worst_a, promotions), flush=True)
    worst_all = max(worst_all, worst_t)
    frames_all += n_frames
print("%d frames: worst deviation from the oracle pipeline %.2e m; frames with another currentID / keyframeID / isMapUpdated: %d"
      % (frames_all, worst_all, mism))
