import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mad_icp_amd import capi, synth
pb = synth.make_problem(16, seed=1)
ctx = capi.Context(0)
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3); ht.transform(T[:3, :3], T[:3, 3]); tids.append(ctx.upload(ht))
h = capi.HostTree(pb["query_scans"][0], 0.2, 0.1, 3)
mids = [ctx.moving_upload(h.leaf_means())]
P = (0.2, 0.1, 0.02)
X0 = capi.pose12(pb["query_guess"][0])[None, :]
for q in (1, 2):
    for bpc in (1,):
        ctx.set_option("queries_per_lane", q); ctx.set_option("grid_blocks_per_cu", bpc)
        lin, sol, v = ctx.icp_time_registration(mids, tids, X0, P, 15, 30)
        first, _ = ctx.icp_time_linearize(mids, tids, X0, P, 40)
        print("qpt %d bpc %d: linearize avg %.2f us, first round %.2f, solve %.2f -> %.0f us/registration" % (q, bpc, lin, first, sol, 15 * (lin + sol)), flush=True)
