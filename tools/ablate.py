"""Ablation timing of icp_linearize with the profiling build (tools/libmadicp_hip_ablate.so, -DMADICP_ABLATE):
MADICP_ABLATE_FLAGS bit 1 (2) = skip the cross-lane reduction, bit 2 (4) = skip the descent.  Results are wrong
by design; only the kernel time matters."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mad_icp_amd import capi, synth
capi._load_orig = capi._load
capi._load = lambda name: ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmadicp_hip_ablate.so")) if "hip" in name else capi._load_orig(name)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pb = synth.make_problem(K, seed=1, n_queries=B)
ctx = capi.Context(0)
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3); ht.transform(T[:3, :3], T[:3, 3]); tids.append(ctx.upload(ht))
mids = []
for s in pb["query_scans"]:
    h = capi.HostTree(s, 0.2, 0.1, 3); mids.append(ctx.moving_upload(h.leaf_means()))
X0 = np.stack([capi.pose12(T) for T in pb["query_guess"]])
P = (0.2, 0.1, 0.02)
ctx.set_option("grid_blocks_per_cu", 3)
for flags, name in ((0, "full"), (4, "no walk (leaf 0 for all: coalesced fetch, ~all rejected)"), (1024, "walk + fetch, no J/H arithmetic"),
                    (2048, "walk, fetch from 8 records only"), (2048 + 1024, "walk only"), (4096, "accept/reject decided per wave (no divergence in the arithmetic)"),
                    (4 + 4096, "no walk, per-wave accept"), (2, "no reduction")):
    os.environ["MADICP_ABLATE_FLAGS"] = str(flags)
    us, _ = ctx.icp_time_linearize(mids, tids, X0, P, 40)
    print("flags %5d: %6.2f us  %s" % (flags, us, name), flush=True)
