"""What a 'bucketed deep phase' would have to stage: statistics of the descents below the LDS-staged top (11 levels) for the
bench workload's geometry, computed on the host from the product's own trees.  Per workgroup pass (768 consecutive moving
leaves against one keyframe tree): distinct sub-trees below the top that the pass enters and the screening records they hold;
per wavefront: hops below the top (mean, and the deepest lane's — what the wave waits for).
Usage: python tools/walk_stats.py [keyframes]   (DESIGN.md section 3.1 quotes the output)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_icp_amd import capi, synth  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
TOP, BLOCK = 11, 768
pb = synth.make_problem(K, seed=1, n_queries=1)
q = capi.HostTree(pb["query_scans"][0], 0.2, 0.1, 3)
T = pb["query_guess"][0]
Q = q.leaf_means() @ T[:3, :3].T + T[:3, 3]
n = len(Q)
subs, recs, deepest, mean_hops = [], [], [], []
for k in range(K):
    ht = capi.HostTree(pb["keyframe_scans"][k], 0.2, 0.1, 3)
    Tk = pb["keyframe_poses"][k]
    ht.transform(Tk[:3, :3], Tk[:3, 3])
    nd = ht.nodes
    right, mean, d = nd["right"], nd["mean"], nd["dir"]  # right: offset of the right child, 0 for a leaf (madicp_hip.h)
    idx, depth, exit_node = np.zeros(n, np.int64), np.zeros(n, np.int64), np.full(n, -1, np.int64)
    live = right[idx] != 0
    lvl = 0
    while live.any():
        if lvl == TOP:
            exit_node[live] = idx[live]
        ii = idx[live]
        left = ((Q[live] - mean[ii]) * d[ii]).sum(1) < 0
        idx[live] = np.where(left, ii + 1, ii + right[ii])
        depth[live] += 1
        live = live & (right[idx] != 0)
        lvl += 1
    size = np.ones(len(nd), np.int64)
    for i in range(len(nd) - 1, -1, -1):
        if right[i]:
            size[i] = 1 + size[i + 1] + size[i + right[i]]
    below = np.maximum(depth - TOP, 0)
    for b in range(0, n, BLOCK):
        ex = exit_node[b:b + BLOCK]
        u = np.unique(ex[ex >= 0])
        subs.append(len(u))
        recs.append(size[u].sum())
        for w in range(b, min(n, b + BLOCK), 64):
            deepest.append(below[w:w + 64].max())
            mean_hops.append(below[w:w + 64].mean())
    print("tree %d: %d leaves, depth mean %.2f max %d, below the top mean %.2f" % (k, ht.num_leaves, depth.mean(), depth.max(), below.mean()))
print("per workgroup pass (%d leaves x one tree): %.0f distinct sub-trees (max %d) holding %.0f screening records = %.1f KB (max %.1f KB)"
      % (BLOCK, np.mean(subs), np.max(subs), np.mean(recs), np.mean(recs) * 16 / 1024, np.max(recs) * 16 / 1024))
print("per wavefront: %.2f hops below the top on average, %.2f for its deepest lane (p95 %d)"
      % (np.mean(mean_hops), np.mean(deepest), np.percentile(deepest, 95)))
