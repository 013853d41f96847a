import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from mad_icp_amd import capi, synth
ctx = capi.Context(0)
rng = np.random.default_rng(3)
def check(name, pts, b_max):
    cid = ctx.cloud_upload(pts)
    t0 = time.perf_counter(); tid, nl = ctx.tree_build(cid, b_max, 0.1); ctx.synchronize(); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); t2, _ = ctx.tree_build(cid, b_max, 0.1); ctx.synchronize(); dt2 = time.perf_counter() - t0
    nn, _ = ctx.tree_info(tid)
    nodes = ctx.tree_download(tid, nn)
    t3 = ctx.tree_upload(nodes, nl)   # validates structure
    leaf = nodes["right"] == 0
    r = ctx.nn_search(tid, nodes["mean"][leaf][::7], want=("leaf", "dist"))
    st = ctx.tree_build_stats()
    print(name, "N", pts.shape[0], "leaves", nl, "levels", st["max_level"], "first %.2f ms second %.2f ms" % (dt*1e3, dt2*1e3),
          "self-query ok", bool((r["dist"] == 0).all()), "repro", ctx.tree_download(t2, nn).tobytes() == nodes.tobytes(), flush=True)
    for t in (tid, t2, t3): ctx.tree_release(t)
    ctx.cloud_release(cid)
check("gauss500k", rng.normal(size=(500000, 3)) * [40, 30, 2.0], 0.2)
check("uniform1M", rng.uniform(-50, 50, size=(1000000, 3)) * [1, 1, 0.02], 0.2)
pb = synth.make_problem(8, seed=4, n_queries=1)
world = np.concatenate([(s @ T[:3, :3].T) + T[:3, 3] for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"])])
check("map8scans", world, 0.2)
check("dense120k", pb["query_scans"][0], 1e-5)
check("plane", np.c_[rng.uniform(-30, 30, size=(200000, 2)), np.zeros(200000)], 0.2)
ctx.close()
