"""Scratch timing of the registration path at full size (not the contract bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mad_icp_amd import capi, synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t0 = time.time()
pb = synth.make_problem(K, seed=1, n_queries=B)
print("gen %.1fs" % (time.time() - t0), flush=True)
ctx = capi.Context(0)
tids = []
t0 = time.time()
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3)
    ht.transform(T[:3, :3], T[:3, 3])
    tids.append(ctx.upload(ht))
print("trees %.2fs, nodes %d" % (time.time() - t0, ht.num_nodes), flush=True)
mids, Ls = [], []
for s in pb["query_scans"]:
    h = capi.HostTree(s, 0.2, 0.1, 3)
    mids.append(ctx.moving_upload(h.leaf_means()))
    Ls.append(h.num_leaves)
print("L", Ls)
X0 = np.stack([capi.pose12(T) for T in pb["query_guess"]])
P = (0.2, 0.1, 0.02)
for opts in [dict(grid_blocks_per_cu=1, lds_stage_min_leaves=m) for m in (1024, 0, 1 << 30)]:
    for k, v in opts.items():
        ctx.set_option(k, v)
    for _ in range(3):
        ctx.icp_register_batch_enqueue(mids, tids, X0, P, 15)
    ctx.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.icp_register_batch_enqueue(mids, tids, X0, P, 15)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / n
    r = ctx.icp_fetch(B)
    err = np.linalg.inv(pb["query_gt"][0]) @ capi.pose44(r["X"][0])
    print(opts, "%.1f us/batch  %.0f reg/s  visits/round/pair %.2f  nmatched %s  terr %.4f" % (
        dt * 1e6, B / dt, r["visits"][0] / (15 * K * Ls[0]), r["n_matched"], np.linalg.norm(err[:3, 3])), flush=True)
us, v = ctx.icp_time_linearize(mids, tids, X0, P, 60)
print("linearize avg %.2f us/launch, visits/launch %s" % (us, v))
