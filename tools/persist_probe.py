"""icp_persist (all GN rounds in one launch) against the per-round launches: same bits?  how fast?  (GPU box only)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mad_icp_amd import capi, synth  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
OPT = os.environ.get("MADICP_AB_OPTION", "persistent")  # the 0/1 library option compared (default: icp_persist)
NQ = 8
PARAMS = (0.2, 0.1, 0.02)
pb = synth.make_problem(K, seed=1, n_queries=1)
pb["query_scans"], pb["query_gt"], pb["query_guess"] = synth.make_query_streams(K, seed=1, n_streams=NQ)
ctx = capi.Context(0)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3)
    ht.transform(T[:3, :3], T[:3, 3])
    tids.append(ctx.upload(ht))
qts = [capi.HostTree(s, 0.2, 0.1, 3) for s in pb["query_scans"]]
leaves = [q.leaf_means() for q in qts]
Ls = [q.num_leaves for q in qts]
mids = [ctx.moving_upload(lm) for lm in leaves]
guess = [capi.pose12(T) for T in pb["query_guess"]]
X0 = np.stack(guess)


def run(persist, opt=OPT):
    ctx.set_option(opt, persist)
    out = {}
    r = ctx.icp_register(mids[0], tids, pb["query_guess"][0], PARAMS, 15, Ls[0])
    out["single"] = r
    out["batch8"] = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
    for nb in (1, 8):
        for _ in range(5):
            ctx.icp_register_batch_enqueue(mids[:nb], tids, X0[:nb], PARAMS, 15)
        ctx.synchronize()
        t = time.perf_counter()
        n = 200 if nb == 1 else 50
        for _ in range(n):
            ctx.icp_register_batch_enqueue(mids[:nb], tids, X0[:nb], PARAMS, 15)
        ctx.synchronize()
        out["resident%d" % nb] = nb * n / (time.perf_counter() - t)
    # streamed
    def streamed(n):
        prev = None
        for i in range(n):
            q = i % NQ
            tk = ctx.stream_submit(leaves[q], tids, guess[q], PARAMS, 15)
            if prev is not None:
                ctx.stream_collect(prev[0], Ls[prev[1]])
            prev = (tk, q)
        return ctx.stream_collect(prev[0], Ls[prev[1]])
    streamed(20)
    t = time.perf_counter()
    last = streamed(400)
    out["streamed"] = 400 / (time.perf_counter() - t)
    out["last"] = last
    return out


a = run(0)
b = run(1)
a2 = run(0)
b2 = run(1)
for key in ("resident1", "resident8", "streamed"):
    a[key] = max(a[key], a2[key])
    b[key] = max(b[key], b2[key])
print("option", OPT)
print("K=%d L=%d" % (K, Ls[0]))
for key in ("resident1", "resident8", "streamed"):
    print("%-10s option = 0 %8.1f   option = 1 %8.1f   (%+.1f %%)" % (key, a[key], b[key], 100 * (b[key] / a[key] - 1)))
sa, sb = a["single"], b["single"]
print("single: X equal %s  H equal %s  matched equal %s  X_iters equal %s  visits %d / %d" % (
    np.array_equal(sa["X"], sb["X"]), np.array_equal(sa["H"], sb["H"]), np.array_equal(sa["matched"], sb["matched"]),
    np.array_equal(sa["X_iters"], sb["X_iters"]), sa["visits"], sb["visits"]))
print("max |dX| %.3e" % np.abs(sa["X"] - sb["X"]).max())
ba, bb = a["batch8"], b["batch8"]
print("batch8: X equal %s  n_matched equal %s  visits equal %s" % (np.array_equal(ba["X"], bb["X"]), np.array_equal(ba["n_matched"], bb["n_matched"]), np.array_equal(ba["visits"], bb["visits"])))
print("streamed last: X equal %s matched equal %s" % (np.array_equal(a["last"]["X"], b["last"]["X"]), np.array_equal(a["last"]["matched"], b["last"]["matched"])))
gt = pb["query_gt"][0]
print("translation error vs ground truth: %.4f m" % np.linalg.norm((np.linalg.inv(gt) @ sb["T"])[:3, 3]))
