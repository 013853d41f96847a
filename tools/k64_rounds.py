"""BASELINE configs[4] (64 keyframes, 8 scans in flight) round by round, without a profiler: registrations of 1, 2, ... 15
rounds (a k-round registration IS the first k rounds of the 15-round one), timed with HIP events by the library
(madicp_icp_time_registration), differenced — microseconds of icp_round and nodes really walked in round k, per option set.
usage (GPU box): python tools/k64_rounds.py ["opt=v,opt=v" ...]      e.g.  python tools/k64_rounds.py queue_walks=0 queue_walks=1"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_icp_amd import capi, synth  # noqa: E402
capi = capi.measure_variant()  # (uses include/madicp_hip_measure.h's aids: the measurement build, mad_icp_amd/_measure)

K, NQ = int(os.environ.get("K64_K", "64")), int(os.environ.get("K64_NQ", "8"))
PARAMS = (0.2, 0.1, 0.02)
pb = synth.make_problem(K, seed=1, n_queries=1)
scans, gts, guesses = synth.make_query_streams(K, seed=1, n_streams=NQ)
ctx = capi.Context(0)
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3)
    ht.transform(T[:3, :3], T[:3, 3])
    tids.append(ctx.upload(ht))
qts = [capi.HostTree(s, 0.2, 0.1, 3) for s in scans]
mids = [ctx.moving_upload(q.leaf_means()) for q in qts]
X0 = np.stack([capi.pose12(T) for T in guesses])
pairs = sum(q.num_leaves for q in qts) * K
print("K=%d, %d scans in flight, %d (leaf, tree) pairs per round" % (K, NQ, pairs))
for group in (sys.argv[1:] or [""]):
    for kv in filter(None, group.split(",")):
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    cum_t, cum_w, cum_v = [0.0], [0.0], [0.0]
    for k in range(1, 16):
        lin, fin, visits, walked = ctx.icp_time_registration(mids, tids, X0, PARAMS, k, reps=6)
        cum_t.append(lin * k)
        cum_w.append(float(walked.astype(np.float64).sum()) * k)
        cum_v.append(float(visits.astype(np.float64).sum()) * k)
    dt, dw, dv = np.diff(cum_t), np.diff(cum_w), np.diff(cum_v)
    print("\n[%s]  15 rounds: %.0f us, i.e. %.1f us per launch" % (group or "defaults", cum_t[-1], cum_t[-1] / 15))
    print("| round | " + " | ".join(str(i) for i in range(15)) + " |")
    print("|" + "---|" * 16)
    print("| us | " + " | ".join("%.0f" % x for x in dt) + " |")
    print("| walked / visited nodes | " + " | ".join(("%.3f" % (w / max(v, 1.0))) for w, v in zip(dw, dv)) + " |")
