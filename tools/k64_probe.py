import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from mad_icp_amd import capi, synth
K, NQ = 64, 8
PARAMS = (0.2, 0.1, 0.02)
pb = synth.make_problem(K, seed=1, n_queries=1)
scans, gts, guesses = synth.make_query_streams(K, seed=1, n_streams=NQ)
ctx = capi.Context(0)
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3); ht.transform(T[:3, :3], T[:3, 3]); tids.append(ctx.upload(ht))
qts = [capi.HostTree(s, 0.2, 0.1, 3) for s in scans]
mids = [ctx.moving_upload(q.leaf_means()) for q in qts]
X0 = np.stack([capi.pose12(T) for T in guesses])
def rate(nb, n):
    for _ in range(3): ctx.icp_register_batch_enqueue(mids[:nb], tids, X0[:nb], PARAMS, 15)
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(n): ctx.icp_register_batch_enqueue(mids[:nb], tids, X0[:nb], PARAMS, 15)
    ctx.synchronize(); return nb * n / (time.perf_counter() - t)
r = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
print("K=64: 1 scan %.0f/s   8 scans %.0f/s   checksum %.12f" % (rate(1, 30), rate(8, 12), float(np.abs(r["X"]).sum())))
