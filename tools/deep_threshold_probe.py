"""When does a launch pay as a DEEP one (every workgroup one range of the scan and ALL the trees of its XCD piece, leaf-major
rounds) instead of one (tree, range) unit per workgroup?  Resident registrations/s for K keyframes x B scans in flight, with the
option deep_min_leaves at its default (two passes of 768 leaves per range) and lower.
usage (GPU box): python tools/deep_threshold_probe.py [K,K,...] [B,B,...] [threshold,threshold,...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_icp_amd import capi, synth  # noqa: E402

Ks = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,32,64").split(",")]
Bs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
Ts = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1536,512,128").split(",")]
PARAMS = (0.2, 0.1, 0.02)
ctx = capi.Context(0)
Kmax = max(Ks)
pb = synth.make_problem(Kmax, seed=1, n_queries=1)
trees = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3)
    ht.transform(T[:3, :3], T[:3, 3])
    trees.append(ctx.upload(ht))
print("| keyframes | scans in flight | " + " | ".join("deep_min_leaves %d" % t for t in Ts) + " |")
print("|---|---|" + "---|" * len(Ts))
for K in Ks:
    scans, gts, guesses = synth.make_query_streams(K, seed=1, n_streams=max(Bs))
    qts = [capi.HostTree(s, 0.2, 0.1, 3) for s in scans]
    mids = [ctx.moving_upload(q.leaf_means()) for q in qts]
    X0 = np.stack([capi.pose12(T) for T in guesses])
    tids = trees[:K]
    for B in Bs:
        row, sums = [], []
        for t in Ts:
            ctx.set_option("deep_min_leaves", t)
            n = max(6, 48 // B)
            for _ in range(3):
                ctx.icp_register_batch_enqueue(mids[:B], tids, X0[:B], PARAMS, 15)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                ctx.icp_register_batch_enqueue(mids[:B], tids, X0[:B], PARAMS, 15)
            ctx.synchronize()
            row.append(B * n / (time.perf_counter() - t0))
            sums.append(float(np.abs(ctx.icp_fetch(B)["X"]).sum()))
        assert max(sums) - min(sums) < 1e-9, sums
        print("| %d | %d | " % (K, B) + " | ".join("%.0f" % v for v in row) + " |", flush=True)
    for m in mids:
        ctx.moving_release(m)
