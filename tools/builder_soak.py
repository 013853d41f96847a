"""Device tree builder against the host builder over MANY full-size scans (development tool, GPU box): per scan the leaf
count, the `right` links, the leaf representatives at the same ordinals, and a second device build of the same cloud (the
bytes must repeat).  Scans: two scenes x poses along the drive x three rigid motions of each cloud (the tree is not
rotation invariant: other split planes, other ties).  usage: python tools/builder_soak.py [n_scans]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mad_icp_amd import capi, synth  # noqa: E402


def main():
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    ctx = capi.Context(0)
    rng = np.random.default_rng(11)
    tot_leaves = tot_diff = topo_diff = repro_diff = 0
    t_dev = []
    t0 = time.perf_counter()
    for i in range(n_scans):
        scene = synth.Scene(i % 5)
        pts = synth.render_scan(scene, synth.path_pose(1.7 * i), 4000 + i)
        if i % 3:  # a rigid motion of the cloud: another tree
            a = rng.uniform(-np.pi, np.pi)
            R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
            if i % 3 == 2:
                b = rng.uniform(-0.3, 0.3)
                R = R @ np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
            pts = np.ascontiguousarray(pts @ R.T + rng.uniform(-5, 5, 3))
        ht = capi.HostTree(pts, 0.2, 0.1, 3)
        cid = ctx.cloud_upload(pts)
        ta = time.perf_counter()
        tid, nl = ctx.tree_build(cid, 0.2, 0.1)
        t_dev.append(time.perf_counter() - ta)
        nodes = ctx.tree_download(tid, 2 * nl - 1)
        t2, nl2 = ctx.tree_build(cid, 0.2, 0.1)
        again = ctx.tree_download(t2, 2 * nl2 - 1)
        repro_diff += int(nl2 != nl or again.tobytes() != nodes.tobytes())
        hn = ht.nodes
        if nl != ht.num_leaves or not np.array_equal(nodes["right"], hn["right"]):
            topo_diff += 1
            print("scan %d: topology differs (leaves %d vs %d)" % (i, nl, ht.num_leaves), flush=True)
        else:
            leaf = nodes["right"] == 0
            same = np.all(nodes["mean"][leaf].view(np.uint64) == hn["mean"][leaf].view(np.uint64), axis=1)
            tot_leaves += int(leaf.sum())
            tot_diff += int((~same).sum())
            if not same.all():
                print("scan %d: %d of %d leaf representatives differ" % (i, int((~same).sum()), int(leaf.sum())), flush=True)
        for t in (tid, t2):
            ctx.tree_release(t)
        ctx.cloud_release(cid)
    print("%d scans in %.1f s: topology differs on %d; leaf representatives that differ: %d of %d; builds that did not repeat "
          "their bytes: %d; device build median %.3f ms" % (n_scans, time.perf_counter() - t0, topo_diff, tot_diff, tot_leaves,
                                                            repro_diff, 1e3 * float(np.median(t_dev))))
    ctx.close()


if __name__ == "__main__":
    main()
