"""Device tree builder vs host builder (development tool, GPU box): topology, leaf representatives, member order and
build time on a set of clouds.  usage: python tools/order_probe.py [n_scans]

Per cloud: nodes / leaves of both builders, fraction of `right` links equal, fraction of leaves whose mean is the host
builder's at the same leaf ordinal (bitwise), fraction of internal nodes whose mean / split normal are bitwise equal,
fraction of points at the same place of the final member order (the host order has every leaf's first member overwritten
by the representative — those rows are compared against the representative), wall-clock of a device build on a resident
cloud (median of 20)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mad_icp_amd import capi, synth  # noqa: E402
capi = capi.measure_variant()  # (uses include/madicp_hip_measure.h's aids: the measurement build, mad_icp_amd/_measure)

B_MAX, B_MIN = 0.2, 0.1


def compare(ctx, name, pts, b_max=B_MAX, b_min=B_MIN, timing=False):
    ht = capi.HostTree(pts, b_max, b_min, 2)
    horder, _ = capi.host_tree_points(pts, b_max, b_min, 2)
    cid = ctx.cloud_upload(pts)
    tid, nl = ctx.tree_build(cid, b_max, b_min)
    nodes = ctx.tree_download(tid, 2 * nl - 1)
    dorder = ctx.tree_build_points(pts.shape[0])
    hn = ht.nodes
    line = "%-12s n=%7d leaves dev %6d host %6d" % (name, pts.shape[0], nl, ht.num_leaves)
    if nl == ht.num_leaves:
        same_right = np.mean(nodes["right"] == hn["right"])
        line += " | right== %.5f" % same_right
        if same_right == 1.0:
            leaf = nodes["right"] == 0
            lm = np.all(nodes["mean"][leaf].view(np.uint64) == hn["mean"][leaf].view(np.uint64), axis=1)
            ln = np.all(nodes["dir"][leaf].view(np.uint64) == hn["dir"][leaf].view(np.uint64), axis=1)
            im = np.all(nodes["mean"][~leaf].view(np.uint64) == hn["mean"][~leaf].view(np.uint64), axis=1)
            idr = np.all(nodes["dir"][~leaf].view(np.uint64) == hn["dir"][~leaf].view(np.uint64), axis=1)
            line += " leaf-mean== %.5f (%d differ) leaf-normal== %.4f | internal mean== %.4f dir== %.4f" % (
                lm.mean(), int((~lm).sum()), ln.mean(), im.mean() if im.size else 1.0, idr.mean() if idr.size else 1.0)
            # member order: the host's container holds the representative in every leaf's first slot
            same = np.all(dorder.view(np.uint64) == horder.view(np.uint64), axis=1)
            line += " | order== %.5f" % same.mean()
            if not same.all():
                # rows that differ only because the host overwrote a leaf's first member
                bad = np.flatnonzero(~same)
                rep = set(map(bytes, hn["mean"][leaf].view(np.uint8).reshape(-1, 24)))
                expl = np.array([bytes(horder[i].view(np.uint8)) in rep for i in bad])
                line += " (of %d rows that differ %d hold a representative on the host side)" % (bad.size, int(expl.sum()))
        else:
            shared = len(set(map(bytes, nodes["mean"][nodes["right"] == 0].view(np.uint8).reshape(-1, 24))) &
                         set(map(bytes, hn["mean"][hn["right"] == 0].view(np.uint8).reshape(-1, 24)))) / ht.num_leaves
            line += " shared leaf means %.5f" % shared
    else:
        shared = len(set(map(bytes, nodes["mean"][nodes["right"] == 0].view(np.uint8).reshape(-1, 24))) &
                     set(map(bytes, hn["mean"][hn["right"] == 0].view(np.uint8).reshape(-1, 24)))) / ht.num_leaves
        line += " | shared leaf means %.5f" % shared
    if timing:
        ts = []
        for _ in range(20):
            t0 = time.perf_counter()
            t2, _ = ctx.tree_build(cid, b_max, b_min)
            ts.append(time.perf_counter() - t0)
            ctx.tree_release(t2)
        line += " | build %.3f ms (min %.3f)" % (1e3 * np.median(ts), 1e3 * min(ts))
    print(line, flush=True)
    ctx.tree_release(tid)
    ctx.cloud_release(cid)


def main():
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    ctx = capi.Context(0)
    rng = np.random.default_rng(5)
    from fixtures import street_problem
    small = {
        "1pt": np.array([[1.0, 2.0, 3.0]]),
        "2pt": np.array([[1.0, 2.0, 3.0], [1.5, 2.0, 3.0]]),
        "dup40": np.repeat(np.array([[1.0, 2.0, 3.0]]), 40, axis=0),
        "line100": np.stack([np.linspace(0, 10, 100), np.zeros(100), np.zeros(100)], 1),
        "expline36": np.stack([100.0 ** np.arange(36), np.zeros(36), np.zeros(36)], 1),
        "gauss33": rng.normal(size=(33, 3)) * [5, 3, 0.05],
        "gauss500": rng.normal(size=(500, 3)) * [5, 3, 0.05],
        "gauss3000": rng.normal(size=(3000, 3)) * [5, 3, 0.05],
        "gauss9000": rng.normal(size=(9000, 3)) * [5, 3, 1.0],
        "street19k": street_problem(2)["query_scans"][0],
    }
    for k, v in small.items():
        compare(ctx, k, v)
    compare(ctx, "street19k-d", small["street19k"], 1e-5, B_MIN)
    pb = synth.make_problem(n_scans, seed=1, n_queries=1)
    compare(ctx, "scan120k", pb["query_scans"][0], timing=True)
    for i, s in enumerate(pb["keyframe_scans"][:n_scans]):
        compare(ctx, "kf%d" % i, s, timing=(i == 0))
    r2 = np.random.default_rng(77)
    for i in range(12):  # the random small clouds of tests/test_gpu_frontend.py
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
        compare(ctx, "rand%d" % i, c, b_max, b_min)
    ctx.close()


if __name__ == "__main__":
    main()
