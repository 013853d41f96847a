#!/usr/bin/env python3
"""Soak run of Pipeline.compute (device front-end, deskew on): a long synthetic drive, device memory in use and frame time
sampled along the way — leaks or growth of the pooled buffers would show as a drift.  GPU box only."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (only for mem_get_info)

from mad_icp.src.pybind import pypeline as pm  # noqa: E402
from mad_icp_amd import synth  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 120
MODE = sys.argv[2] if len(sys.argv) > 2 else "deskew"  # deskew (default) | plain (deskew off) | lookahead (deskew off,
LOOKAHEAD = MODE == "lookahead"                          # prefetch(next) before compute: the build stream's path)
scene = synth.Scene(0)
scans = [synth.render_scan(scene, synth.path_pose(0.5 * i), 300 + i) for i in range(24)]
pl = pm.Pipeline(10.0, MODE == "deskew", 0.2, 0.1, 0.8, 0.1, 0.02, 16, 8, False)
pl.setDeviceFrontEnd(True)
ts, mem = [], []


def scan_of(i):
    return scans[i % len(scans)] if (i // len(scans)) % 2 == 0 else scans[len(scans) - 1 - i % len(scans)]  # back and forth


for i in range(n_frames):
    s = scan_of(i)
    t0 = time.perf_counter()
    if LOOKAHEAD and i + 1 < n_frames and i % 97 != 96:  # (every 97th frame comes without a look-ahead: the fall-back path)
        pl.prefetch(scan_of(i + 1))
    pl.compute(0.1 * i, s)
    ts.append(time.perf_counter() - t0)
    if i % 20 == 19:
        free, total = torch.cuda.mem_get_info()
        mem.append((total - free) >> 20)
        print("frame %4d: median %.3f ms over the last 20, device memory in use %d MiB, keyframes %d, inliers %.3f"
              % (i + 1, 1e3 * np.median(ts[-20:]), mem[-1], pl.numKeyframes(), pl.lastInliersRatio()), flush=True)
print("memory drift over the run: %d MiB" % (mem[-1] - mem[1] if len(mem) > 2 else 0))
