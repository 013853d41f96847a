"""Where does a converged (walk-free) icp_linearize round spend its time?  Start the registration AT the converged
pose, so 14 of 15 rounds reuse every correspondence, and time the registration with the profiling build's flags."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mad_icp_amd import capi, synth
capi._load_orig = capi._load
capi._load = lambda name: ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmadicp_hip_ablate.so")) if "hip" in name else capi._load_orig(name)
K = 16
pb = synth.make_problem(K, seed=1)
ctx = capi.Context(0)
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3); ht.transform(T[:3, :3], T[:3, 3]); tids.append(ctx.upload(ht))
h = capi.HostTree(pb["query_scans"][0], 0.2, 0.1, 3)
mids = [ctx.moving_upload(h.leaf_means())]
P = (0.2, 0.1, 0.02)
r = ctx.icp_register(mids[0], tids, pb["query_guess"][0], P, 15, h.num_leaves)
Xc = r["X"][None, :]
for bpc in (3,):
    ctx.set_option("grid_blocks_per_cu", bpc)
    for flags, name in ((0, "full"), (1024, "no J/H arithmetic"), (2048, "leaf fetch from 8 records"), (2048 + 1024, "neither"), (2, "no reduction"), (2 + 2048 + 1024, "only loads of moving+cache")):
        os.environ["MADICP_ABLATE_FLAGS"] = str(flags | 1)  # | kFlagNoUpdate: the pose stays converged whatever the flags break
        lin, sol, _ = ctx.icp_time_registration(mids, tids, Xc, P, 15, 30)
        print("bpc %d flags %5d: linearize avg %.2f us (solve %.2f)  %s" % (bpc, flags, lin, sol, name), flush=True)
