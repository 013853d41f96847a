"""Where a drop-in frame's host time goes (GPU box): tree build, upload call (validate + top layout + staging copy +
enqueue), stream submission, collection, transform — medians over a short synthetic drive."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_icp_amd import capi, synth  # noqa: E402

K = 16
pb = synth.make_problem(K, seed=1, n_queries=1)
scene = synth.Scene(0)
drive = [synth.render_scan(scene, synth.path_pose(1.0 * i), 100 + i) for i in range(12)]
ctx = capi.Context(0)
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 4)
    ht.transform(T[:3, :3], T[:3, 3])
    tids.append(ctx.upload(ht))
PARAMS = (0.2, 0.1, 0.02)
rows = []
for rep in range(3):
    for sc in drive:
        t0 = time.perf_counter()
        cp = np.array(sc, copy=True)
        t1 = time.perf_counter()
        ht = capi.HostTree(cp, 0.2, 0.1, 4)
        t2 = time.perf_counter()
        tid = ctx.upload(ht)
        t3 = time.perf_counter()
        tk = ctx.stream_submit_tree(tid, tids, pb["query_guess"][0], PARAMS, 15)
        t4 = time.perf_counter()
        r = ctx.stream_collect(tk, ht.num_leaves)
        t5 = time.perf_counter()
        ctx.tree_transform(tid, r["T"][:3, :3], r["T"][:3, 3])
        ctx.tree_release(tid)
        t6 = time.perf_counter()
        rows.append([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t0])
m = np.median(np.array(rows[12:]), axis=0) * 1e3
print("copy %.3f  build %.3f  upload-call %.3f  submit %.3f  collect %.3f  transform+release %.3f  total %.3f ms" % tuple(m))
