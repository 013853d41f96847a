"""Where exactly does the device builder's tree leave the host builder's on the sixty random small clouds of
tests/test_gpu_frontend_oracle.py?  For every cloud whose topology differs: the first node (DFS preorder) whose `right` link
differs, that node and its parent in both trees (mean, split direction / normal, bbox0), and how the host builder's own
topology reacts to perturbations of the input by 1, 4 and 64 ulps.   usage (GPU box): python tools/small_cloud_diff.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mad_icp_amd import capi  # noqa: E402

np.set_printoptions(precision=17, linewidth=200)
ctx = capi.Context(0)
r2 = np.random.default_rng(77)
for i in range(60):
    n = int(r2.integers(1, 400))
    kind = int(r2.integers(4))
    c = r2.normal(size=(n, 3)) * r2.choice([0.01, 0.3, 5.0], size=3)
    if kind == 1:
        c[:, 2] = 0.0
    elif kind == 2:
        c[:, 1:] = 0.0
    elif kind == 3:
        c = np.repeat(c[: max(1, n // 4)], 4, axis=0)
    c = c + r2.normal(size=3) * 10.0
    b_max, b_min = float(r2.choice([1e-5, 0.05, 0.2, 1.0])), float(r2.choice([0.01, 0.1, 0.5]))
    r2.integers(3)
    ht = capi.HostTree(c, b_max, b_min, 2)
    cid = ctx.cloud_upload(c)
    tid, nl = ctx.tree_build(cid, b_max, b_min)
    dn = ctx.tree_download(tid, 2 * nl - 1)
    ctx.tree_release(tid)
    ctx.cloud_release(cid)
    hn = ht.nodes
    if dn.shape[0] == hn.shape[0] and np.array_equal(dn["right"], hn["right"]):
        continue
    m = min(dn.shape[0], hn.shape[0])
    first = int(np.flatnonzero(dn["right"][:m] != hn["right"][:m])[0]) if (dn["right"][:m] != hn["right"][:m]).any() else m
    print("=== cloud %d: %d points, kind %d, b_max %g, b_min %g; leaves device %d host %d; first differing node %d"
          % (i, c.shape[0], kind, b_max, b_min, nl, ht.num_leaves, first))
    # parent of `first` in the host tree (preorder: the last node j < first whose sub-tree spans first)
    par = -1
    for j in range(first - 1, -1, -1):
        r = int(hn["right"][j])
        if r > 0:  # internal: left child j + 1, right child j + r; sub-tree size unknown here, walk down instead
            pass
    # walk down from the root following which child contains `first`
    j, size = 0, hn.shape[0]
    chain = []
    while j != first and hn["right"][j] > 0:
        chain.append(j)
        r = int(hn["right"][j])
        if first < j + r:
            j, size = j + 1, r - 1
        else:
            j, size = j + r, size - r
    for j in chain[-2:] + [first]:
        for name, a in (("host  ", hn), ("device", dn)):
            if j < a.shape[0]:
                print("  node %3d %s right %4d leaf_id %4d bbox0 %.17g\n      mean %s\n      dir  %s" % (j, name, a["right"][j], a["leaf_id"][j], a["bbox0"][j], a["mean"][j], a["dir"][j]))
    for ulps in (1, 4, 64, 4096):
        rng = np.random.default_rng(1000 + i)
        kept = 0
        for _ in range(8):
            q = c.copy()
            q.view(np.int64)[...] += rng.integers(-ulps, ulps + 1, size=q.shape)
            r = capi.HostTree(q, b_max, b_min, 2).nodes["right"]
            kept += int(r.shape == hn["right"].shape and np.array_equal(r, hn["right"]))
        print("  host builder keeps its topology under +-%d ulp perturbations of every coordinate: %d of 8" % (ulps, kept))
