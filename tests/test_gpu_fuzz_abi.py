"""Seeded random walks over the C ABI: uploads, device builds (synchronous, look-ahead begin / end / cancel), transforms,
searches, streamed and blocking registrations, releases — in random order on ONE context, with every result checked against a
model kept on the host:

  * a tree's node array (download) equals the model's — uploads round-trip bit for bit, transforms equal the host classes'
    (the bitwise claim of tests/test_gpu_parity.py), a look-ahead build equals the synchronous build of the same cloud;
  * a search equals a numpy descent over the model's nodes in the reference's operation order (mad_tree.cpp:144-152);
  * a streamed registration equals the blocking one on the same inputs bit for bit, whatever was in flight around it;
  * refused calls (the builder's scratch is taken) say so and change nothing.

What this is for: the library recycles device buffers through a pool ordered by events on three streams; a stale event, a
buffer handed out while a stream still reads it, a slot left pending would all show up here as a wrong array sooner or later."""
import numpy as np
import pytest

from fixtures import B_MAX, B_MIN, PARAMS
from mad_icp_amd import capi, synth

pytestmark = pytest.mark.gpu


DEFAULT_ROUTE = {"use_graph": 1, "eager_when_busy": 1, "seq_completion": 1, "host_feed_wait": 1, "wait_mode": 0, "nn_lds_top": 0,
                 "persistent": 0, "xcd_fold": 0}  # madicp_capi.hip


def descend(nodes, q):
    """bestMatchingLeafFast over a linear node array, the reference's arithmetic: (q - mean) . dir with the contiguous
    3-vector reduction order (oracle/linalg.h dotc), left when negative."""
    right, mean, d, leaf_id = nodes["right"], nodes["mean"], nodes["dir"], nodes["leaf_id"]
    idx = np.zeros(len(q), np.int64)
    live = right[idx] != 0
    while live.any():
        ii = idx[live]
        e = q[live] - mean[ii]
        s = (e[:, 0] * d[ii, 0] + e[:, 1] * d[ii, 1]) + e[:, 2] * d[ii, 2]
        idx[live] = np.where(s < 0, ii + 1, ii + right[ii])
        live = right[idx] != 0
    return leaf_id[idx].astype(np.uint32)


def rigid(rng, scale=1.0):
    w = rng.normal(size=3) * 0.2 * scale
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
    return R, rng.normal(size=3) * 2.0 * scale


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_walk_over_the_abi(ctx, seed):
    rng = np.random.default_rng(seed)
    scene = synth.Scene(seed)
    scans = [synth.render_scan(scene, synth.path_pose(1.5 * i), 40 + i, n_beams=16, n_azimuth=300) for i in range(6)]
    trees = {}      # id -> dict(nodes=model node array, leaves=int)
    pending = None  # look-ahead in flight: the scan it was given
    inflight = []   # streamed registrations: (ticket, expected result, L)
    counts = dict(upload=0, build=0, begin=0, end=0, cancel=0, transform=0, search=0, register=0, release=0, refused=0)

    def model_of_device_build(scan):
        cid = ctx.cloud_upload(scan)
        tid, nl = ctx.tree_build(cid, B_MAX, B_MIN)
        ctx.cloud_release(cid)
        nodes = ctx.tree_download(tid, 2 * nl - 1)
        ctx.tree_release(tid)
        return nodes, nl

    reference_builds = {}

    for step in range(260):
        op = rng.choice(["upload", "build", "begin", "end", "cancel", "transform", "search", "register", "collect", "release"],
                        p=[0.10, 0.08, 0.10, 0.10, 0.03, 0.14, 0.15, 0.14, 0.08, 0.08])
        k = int(rng.integers(len(scans)))
        if op == "upload" and len(trees) < 12:
            ht = capi.HostTree(scans[k], B_MAX, B_MIN, 1)
            tid = ctx.upload(ht, trusted=bool(rng.integers(2)))
            trees[tid] = dict(nodes=ht.nodes.copy(), leaves=ht.num_leaves)
            counts["upload"] += 1
        elif op == "build" and len(trees) < 12:
            if pending is not None:
                with pytest.raises(capi.MadIcpError, match="look-ahead tree build is in flight"):
                    ctx.tree_build(ctx.cloud_upload(scans[k]), B_MAX, B_MIN)
                counts["refused"] += 1  # (the cloud of the refused call stays resident: released with the context)
                continue
            cid = ctx.cloud_upload(scans[k])
            tid, nl = ctx.tree_build(cid, B_MAX, B_MIN)
            ctx.cloud_release(cid)
            trees[tid] = dict(nodes=ctx.tree_download(tid, 2 * nl - 1), leaves=nl)
            if k in reference_builds:  # device builds are bit-reproducible run to run
                assert trees[tid]["nodes"].tobytes() == reference_builds[k].tobytes()
            reference_builds[k] = trees[tid]["nodes"].copy()
            counts["build"] += 1
        elif op == "begin":
            if pending is not None:
                with pytest.raises(capi.MadIcpError, match="look-ahead tree build is in flight"):
                    ctx.tree_build_begin(scans[k], B_MAX, B_MIN)
                counts["refused"] += 1
                continue
            ctx.tree_build_begin(scans[k], B_MAX, B_MIN)
            pending = k
            counts["begin"] += 1
        elif op == "end":
            if pending is None:
                with pytest.raises(capi.MadIcpError, match="no look-ahead tree build in flight"):
                    ctx.tree_build_end()
                continue
            tid, nl = ctx.tree_build_end()
            got = ctx.tree_download(tid, 2 * nl - 1)
            if pending not in reference_builds:
                reference_builds[pending] = model_of_device_build(scans[pending])[0]
            assert got.tobytes() == reference_builds[pending].tobytes(), "look-ahead build differs from the synchronous build"
            if len(trees) < 12:
                trees[tid] = dict(nodes=got, leaves=nl)
            else:
                ctx.tree_release(tid)
            pending = None
            counts["end"] += 1
        elif op == "cancel":
            ctx.tree_build_cancel()
            pending = None
            counts["cancel"] += 1
        elif op == "transform" and trees:
            tid = list(trees)[int(rng.integers(len(trees)))]
            R, t = rigid(rng, 0.2)
            ctx.tree_transform(tid, R, t)
            m = trees[tid]["nodes"]  # tree_transform's arithmetic (mad_tree.cpp:165-172): r * v in the strided order, then + t
            for field, add in (("mean", t), ("dir", None)):
                v = m[field].copy()
                x, y, z = v[:, 0].copy(), v[:, 1].copy(), v[:, 2].copy()
                for r in range(3):
                    rv = R[r, 0] * x + (R[r, 1] * y + R[r, 2] * z)
                    v[:, r] = rv + add[r] if add is not None else rv
                m[field] = v
            counts["transform"] += 1
        elif op == "search" and trees:
            tid = list(trees)[int(rng.integers(len(trees)))]
            m = trees[tid]
            got = ctx.tree_download(tid, len(m["nodes"]))
            assert got.tobytes() == m["nodes"].tobytes(), "resident tree differs from the model (step %d)" % step
            q = scans[k][:: 7] + rng.normal(size=3) * 0.05
            leaf = ctx.nn_search(tid, q, want=("leaf",))["leaf"]
            assert np.array_equal(leaf, descend(m["nodes"], q))
            counts["search"] += 1
        elif op == "register" and trees and len(inflight) < 3:
            ids = [t for t in trees if rng.random() < 0.5][:6] or [next(iter(trees))]
            moving = capi.HostTree(scans[k], B_MAX, B_MIN, 1).leaf_means()
            T = synth.path_pose(1.5 * k)
            mid = ctx.moving_upload(moving)
            want = ctx.icp_register(mid, ids, T, PARAMS, 6, moving.shape[0])
            ctx.moving_release(mid)
            # every launch route is claimed bit-identical: the streamed twin runs under randomly chosen ones
            route = {"use_graph": int(rng.integers(2)), "eager_when_busy": int(rng.integers(2)), "seq_completion": int(rng.integers(2)),
                     "host_feed_wait": int(rng.integers(2)), "wait_mode": int(rng.integers(3)), "nn_lds_top": int(rng.integers(2))}
            fused = int(rng.integers(3))  # 0 per-round launches, 1 all rounds in one launch, 2 XCD fold
            route["persistent"], route["xcd_fold"] = int(fused == 1), int(fused == 2)
            for key, v in route.items():
                ctx.set_option(key, v)
            tk = ctx.stream_submit(moving, ids, T, PARAMS, 6)
            for key, v in DEFAULT_ROUTE.items():
                ctx.set_option(key, v)
            inflight.append((tk, want, moving.shape[0]))
            counts["register"] += 1
        elif op == "collect" and inflight:
            tk, want, L = inflight.pop(0)  # tickets are collected in order
            r = ctx.stream_collect(tk, L)
            assert np.array_equal(r["X"], want["X"]) and np.array_equal(r["H"], want["H"]) and np.array_equal(r["b"], want["b"])
            assert np.array_equal(r["matched"], want["matched"])
        elif op == "release" and len(trees) > 2 and not inflight:
            tid = list(trees)[int(rng.integers(len(trees)))]
            ctx.tree_release(tid)
            del trees[tid]
            counts["release"] += 1
    # drain
    for tk, want, L in inflight:
        r = ctx.stream_collect(tk, L)
        assert np.array_equal(r["X"], want["X"]) and np.array_equal(r["H"], want["H"])
    ctx.tree_build_cancel()
    for tid, m in trees.items():
        assert ctx.tree_download(tid, len(m["nodes"])).tobytes() == m["nodes"].tobytes()
        ctx.tree_release(tid)
    assert counts["search"] > 10 and counts["register"] > 10 and counts["begin"] > 5 and counts["transform"] > 10, counts
