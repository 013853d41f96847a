import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from mad_icp_amd import capi, synth
K, NQ = int(os.environ.get("K64_K", "64")), 8
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
def once(label):
    r = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
    only8 = os.environ.get("K64_ONLY8", "0") == "1"  # (counter passes: every icp_round launch then has the 8-scan geometry)
    print(("K=%d " % K) + "%-28s: 1 scan %.0f/s   8 scans %.0f/s   checksum %.12f" % (label, 0.0 if only8 else rate(1, 30), rate(8, 12), float(np.abs(r["X"]).sum())), flush=True)
once("defaults")
if os.environ.get("K64_INTERLEAVE", "0") == "1":  # keyframes dealt round-robin over the eight XCD pieces instead of contiguously
    tids = [tids[(p % 8) * (K // 8) + p // 8] for p in range(K)]
    once("trees interleaved over XCDs")
# development A/B: MADICP_AB="key=value;key=value,key2=value" — one more line per ';'-separated option set (options stay set)
for group in filter(None, os.environ.get("MADICP_AB", "").split(";")):
    for kv in group.split(","):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    once(group)
