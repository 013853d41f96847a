#!/usr/bin/env python3
"""Times madicp_tree_build on the bench scan (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mad_icp_amd import capi, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pb = synth.make_problem(1, seed=1, n_queries=1)
scan = pb["query_scans"][0]
ctx = capi.Context(0)
cid = ctx.cloud_upload(scan)
for _ in range(3):
    t, nl = ctx.tree_build(cid, 0.2, 0.1)
    ctx.tree_release(t)
ctx.synchronize()
ts = []
for _ in range(reps):
    t0 = time.perf_counter()
    t, nl = ctx.tree_build(cid, 0.2, 0.1)
    ctx.synchronize()
    ts.append(time.perf_counter() - t0)
    ctx.tree_release(t)
print("N=%d leaves=%d build median %.3f ms min %.3f ms" % (scan.shape[0], nl, 1e3 * np.median(ts), 1e3 * min(ts)))
print(ctx.tree_build_stats())
ctx.close()
