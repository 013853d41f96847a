"""Why icp_round deals its ranges (kernels.hip.h, "Ranges") and what rejecting far keyframes early would be worth — CPU only.

BASELINE configs[4] (64 keyframes 3 m apart, the query past the last one), the launch geometry of a batch of eight: every scan
is cut into 4 ranges, a workgroup owns one range and the 8 trees of its XCD piece.  For every sampled keyframe this prints, per
CONTIGUOUS range of the scan's leaf order (= a stretch of space: the leaf order is the tree's depth-first order),
  * the pairs that pass the gate at the true pose (mad_icp.cpp:81-83) — what a converged round evaluates, and
  * the fraction of pairs further from the box of ALL the keyframe's leaf means than their ball (+ 1 m + 0.05 |p|): what a
    test of the tree's bounding box in front of the walk would remove,
then the same per range when the ranges are dealt in groups of 64 leaves (option interleave_ranges).
usage: python tools/range_balance.py [every-n-th keyframe, default 4]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_icp_amd import capi, synth  # noqa: E402

B_MAX, B_MIN, B_RATIO = 0.2, 0.1, 0.02
SEED, K, RPT = 1, 64, 4
step = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scene = synth.Scene(SEED)
scans, gts, guesses = synth.make_query_streams(K, seed=SEED, n_streams=1)
qt = capi.HostTree(scans[0], B_MAX, B_MIN, 3)
P = qt.leaf_means()
pn = np.linalg.norm(P, axis=1)
G = B_MIN + B_RATIO * pn
L = P.shape[0]
q = P @ gts[0][:3, :3].T + gts[0][:3, 3]
S = (L + RPT - 1) // RPT
contiguous = [np.arange(r * S, min(L, (r + 1) * S)) for r in range(RPT)]
groups = np.arange(L) >> 6
dealt = [np.nonzero(groups % RPT == r)[0] for r in range(RPT)]
print("L = %d leaves, %d ranges; x extent of the contiguous ranges (sensor frame): %s" % (
    L, RPT, [(int(P[ix, 0].min()), int(P[ix, 0].max())) for ix in contiguous]))
acc_c, acc_d, rej_c = np.zeros(RPT), np.zeros(RPT), []
print("| keyframe | accepted pairs per contiguous range | ... per dealt range | behind the tree's box, per contiguous range |")
print("|---|---|---|---|")
for k in range(0, K, step):
    T = synth.path_pose(k * 3.0)
    ht = capi.HostTree(synth.render_scan(scene, T, SEED * 1000 + k), B_MAX, B_MIN, 3)
    ht.transform(T[:3, :3], T[:3, 3])
    nodes = ht.nodes
    idx = np.zeros(L, dtype=np.int64)
    live = nodes["right"][idx] != 0
    while live.any():  # greedy descent (mad_tree.cpp:144-152)
        nd = nodes[idx]
        s = ((q - nd["mean"]) * nd["dir"]).sum(1)
        idx = np.where(live, np.where(s < 0, idx + 1, idx + nd["right"]), idx)
        live = live & (nodes["right"][idx] != 0)
    ok = np.linalg.norm(q - nodes["mean"][idx], axis=1) <= G
    leaves = nodes["mean"][nodes["right"] == 0]
    lo, hi = leaves.min(0), leaves.max(0)
    d = np.maximum(np.maximum(lo - q, q - hi), 0.0)
    far = np.sqrt((d * d).sum(1)) > G + 1.0 + 0.05 * pn
    a_c = np.array([ok[ix].sum() for ix in contiguous])
    a_d = np.array([ok[ix].sum() for ix in dealt])
    f_c = [far[ix].mean() for ix in contiguous]
    acc_c += a_c
    acc_d += a_d
    rej_c.append(f_c)
    print("| %d | %s | %s | %s |" % (k, " ".join("%5d" % v for v in a_c), " ".join("%5d" % v for v in a_d), " ".join("%.2f" % v for v in f_c)), flush=True)
print("\naccepted pairs per range over the sampled keyframes: contiguous %s (max / mean %.2f), dealt %s (max / mean %.2f)" % (
    acc_c.astype(int).tolist(), acc_c.max() / acc_c.mean(), acc_d.astype(int).tolist(), acc_d.max() / acc_d.mean()))
print("pairs behind the tree's box, mean over the sampled keyframes, per contiguous range: %s" % np.round(np.mean(rej_c, 0), 3).tolist())
