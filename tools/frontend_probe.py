#!/usr/bin/env python3
"""Development probe of the device front-end (tree build, deskew, ingest): runs every piece against the host builder /
the oracle and PRINTS what it finds (no asserts) — one gpurun call answers most questions.  tests/test_gpu_frontend.py
holds the asserting version.  Uses the oracle as checker only (tools/ is not product code)."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from mad_icp_amd import capi, synth  # noqa: E402

B_MAX, B_MIN = 0.2, 0.1


def keyset(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return set(map(bytes, a.view(np.uint8).reshape(a.shape[0], 24)))


def compare_tree(ctx, pts, b_max, b_min, tag):
    t0 = time.perf_counter()
    ht = capi.HostTree(pts, b_max, b_min, 3)
    t_host = time.perf_counter() - t0
    cid = ctx.cloud_upload(pts)
    ctx.synchronize()
    t0 = time.perf_counter()
    tid, nl = ctx.tree_build(cid, b_max, b_min)
    ctx.synchronize()
    t_dev = time.perf_counter() - t0
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        t2, _ = ctx.tree_build(cid, b_max, b_min)
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
        ctx.tree_release(t2)
    nn, nl2 = ctx.tree_info(tid)
    st = ctx.tree_build_stats()
    print(f"[{tag}] N={pts.shape[0]} host leaves={ht.num_leaves} dev leaves={nl} nodes={nn} | host build {t_host*1e3:.2f} ms, "
          f"dev first {t_dev*1e3:.2f} ms, dev steady {np.median(ts)*1e3:.3f} ms (min {min(ts)*1e3:.3f})")
    print(f"   levels={st['max_level']} lane_subtrees={st['lane_subtrees']} wave/level={st['wave_nodes'][:30].tolist()} "
          f"chip/level={st['chip_nodes'][:10].tolist()}")
    nodes = ctx.tree_download(tid, nn)
    # structure: re-upload through the validating path
    try:
        t3 = ctx.tree_upload(nodes, nl)
        ctx.tree_release(t3)
        print("   structure: valid DFS preorder (madicp_tree_upload accepted it)")
    except Exception as e:  # noqa: BLE001
        print("   structure: INVALID:", e)
    leaf = nodes["right"] == 0
    dm = nodes["mean"][leaf]
    hm = ht.nodes["mean"][ht.nodes["right"] == 0]
    ks_in = keyset(pts)
    ks_d, ks_h = keyset(dm), keyset(hm)
    print(f"   leaf means that are input points: dev {len(ks_d & ks_in)}/{len(ks_d)}  host {len(ks_h & ks_in)}/{len(ks_h)}")
    print(f"   leaf means shared with the host builder: {len(ks_d & ks_h)} = {100.0*len(ks_d & ks_h)/max(len(ks_h),1):.3f} % of host leaves")
    # leaf ids are 0..L-1 in preorder
    lid = nodes["leaf_id"][leaf]
    print("   leaf ids in preorder:", bool(np.array_equal(lid, np.arange(nl))), " internal leaf_id all -1:", bool((nodes["leaf_id"][~leaf] == -1).all()))
    # normals unit, bbox0 sane
    nrm = np.linalg.norm(nodes["dir"], axis=1)
    print(f"   |dir| range [{np.nanmin(nrm):.6f}, {np.nanmax(nrm):.6f}] NaN dirs: {int(np.isnan(nrm).sum())}; bbox0 max {np.nanmax(nodes['bbox0']):.3f}")
    # per-node agreement in preorder when the topology is the same
    if nn == ht.num_nodes:
        same_right = nodes["right"] == ht.nodes["right"]
        print(f"   same node count; right-offsets equal at {same_right.mean()*100:.3f} % of nodes; "
              f"max |mean diff| {np.nanmax(np.abs(nodes['mean']-ht.nodes['mean'])):.3e}, max |dir diff| (sign-free) "
              f"{np.nanmax(np.minimum(np.abs(nodes['dir']-ht.nodes['dir']), np.abs(nodes['dir']+ht.nodes['dir']))):.3e}")
    # self query of the leaf means: distance exactly 0, own ordinal
    r = ctx.nn_search(tid, dm, want=("leaf", "dist"))
    print(f"   self-query of leaf means: dist==0 for {(r['dist']==0).sum()}/{nl}, own ordinal for {(r['leaf']==np.arange(nl)).sum()}/{nl}")
    # reproducibility
    t4, _ = ctx.tree_build(cid, b_max, b_min)
    n4 = ctx.tree_download(t4, nn)
    print("   bit-reproducible:", n4.tobytes() == nodes.tobytes())
    ctx.tree_release(t4)
    ctx.cloud_release(cid)
    return tid, nl, ht


def main():
    ctx = capi.Context(0)
    # 1) small and degenerate clouds
    rng = np.random.default_rng(5)
    for tag, pts in [("1pt", np.array([[1.0, 2.0, 3.0]])),
                     ("2pt", np.array([[1.0, 2.0, 3.0], [1.5, 2.0, 3.0]])),
                     ("dup40", np.repeat(np.array([[1.0, 2.0, 3.0]]), 40, axis=0)),
                     ("line100", np.stack([np.linspace(0, 10, 100), np.zeros(100), np.zeros(100)], 1)),
                     ("gauss3000", rng.normal(size=(3000, 3)) * [5, 3, 0.05]),
                     ("gauss20000", rng.normal(size=(20000, 3)) * [20, 10, 1.0])]:
        try:
            tid, _, _ = compare_tree(ctx, pts, B_MAX, B_MIN, tag)
            ctx.tree_release(tid)
        except Exception:  # noqa: BLE001
            print(f"[{tag}] FAILED")
            traceback.print_exc()
    # 2) the bench scan
    pb = synth.make_problem(4, seed=1, n_queries=1)
    scan = pb["query_scans"][0]
    try:
        tid, nl, ht = compare_tree(ctx, scan, B_MAX, B_MIN, "scan120k")
        # registration with device-built keyframes vs host-built keyframes
        params = (B_MAX, 0.1, 0.02)
        dev_t, host_t = [], []
        for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
            c = ctx.cloud_upload(s)
            t, _ = ctx.tree_build(c, B_MAX, B_MIN)
            ctx.tree_transform(t, T[:3, :3], T[:3, 3])
            dev_t.append(t)
            ctx.cloud_release(c)
            h = capi.HostTree(s, B_MAX, B_MIN, 3)
            h.transform(T[:3, :3], T[:3, 3])
            host_t.append(ctx.upload(h))
        T0 = pb["query_guess"][0]
        tk = ctx.stream_submit_tree(tid, dev_t, T0, params, 15)
        rd = ctx.stream_collect(tk, nl)
        tk = ctx.stream_submit(ht.leaf_means(), host_t, T0, params, 15)
        rh = ctx.stream_collect(tk, ht.num_leaves)
        d = np.linalg.inv(rh["T"]) @ rd["T"]
        gt = pb["query_gt"][0]
        print(f"   registration dev-built vs host-built trees: |dt|={np.linalg.norm(d[:3,3]):.3e} m, "
              f"err vs gt dev {np.linalg.norm((np.linalg.inv(gt)@rd['T'])[:3,3]):.4f} host {np.linalg.norm((np.linalg.inv(gt)@rh['T'])[:3,3]):.4f}; "
              f"matched dev {rd['n_matched']}/{nl} host {rh['n_matched']}/{ht.num_leaves}")
        # device tree re-uploaded through the host path: identical results (device top layout == host top layout)
        nn, _ = ctx.tree_info(dev_t[0])
        re_t = []
        for t in dev_t:
            n_, l_ = ctx.tree_info(t)
            re_t.append(ctx.tree_upload(ctx.tree_download(t, n_), l_))
        tk = ctx.stream_submit_tree(tid, re_t, T0, params, 15)
        rr = ctx.stream_collect(tk, nl)
        print("   device-built trees vs the same trees re-uploaded: identical X:", bool(np.array_equal(rr["X"], rd["X"])),
              "identical H:", bool(np.array_equal(rr["H"], rd["H"])))
    except Exception:  # noqa: BLE001
        print("[scan120k] FAILED")
        traceback.print_exc()
    # 3) dense tree (b_max = 1e-5): the reference's nn_search.py property, total self-query error exactly 0
    try:
        sub = scan[::4]
        cid = ctx.cloud_upload(sub)
        t0 = time.perf_counter()
        tid2, nl2 = ctx.tree_build(cid, 1e-5, B_MIN)
        ctx.synchronize()
        print(f"[dense] N={sub.shape[0]} leaves={nl2} build {1e3*(time.perf_counter()-t0):.2f} ms, stats {ctx.tree_build_stats()['max_level']} levels")
        r = ctx.nn_search(tid2, sub, want=("dist",))
        print(f"   self-query total error {r['dist'].sum():.3e}, zero for {(r['dist']==0).sum()}/{sub.shape[0]}")
        ctx.cloud_release(cid)
    except Exception:  # noqa: BLE001
        print("[dense] FAILED")
        traceback.print_exc()
    # 4) deskew vs the oracle
    try:
        Tp = np.eye(4)
        Tn = synth_pose(0.9, 0.05, 0.02, 0.03)
        ref, vel = O.deskew(scan, Tp, Tn, 10.0)
        cid = ctx.cloud_upload(scan)
        ts = []
        for _ in range(3):
            c2 = ctx.cloud_upload(scan)
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.cloud_deskew(c2, vel, 10.0)
            ctx.synchronize()
            ts.append(time.perf_counter() - t0)
            ctx.cloud_release(c2)
        chunks = ctx.cloud_deskew(cid, vel, 10.0, want_chunks=True)
        out = ctx.cloud_download(cid)
        same = (out == ref).all(axis=1)
        print(f"[deskew] {scan.shape[0]} pts, {1e3*np.median(ts):.3f} ms; rows bit-identical to the oracle: {same.sum()}/{len(same)}; "
              f"max |diff| {np.abs(out-ref).max():.3e}; as a set: {len(keyset(out) & keyset(ref))} shared; chunks used {chunks.min()}..{chunks.max()}")
        ctx.cloud_release(cid)
    except Exception:  # noqa: BLE001
        print("[deskew] FAILED")
        traceback.print_exc()
    # 5) ingest vs the numpy restatement
    try:
        rec = np.zeros((scan.shape[0] + 5, 4), np.float32)
        rec[:-5, :3] = scan.astype(np.float32)
        rec[:-5, 3] = 0.5
        rec[-5] = [np.nan, 1, 1, 0]
        rec[-4] = [0.1, 0.1, 0.1, 0]
        rec[-3] = [500, 0, 0, 0]
        rec[-2] = [0, 0, 5, 0]   # on the z axis: the KITTI rotation axis has zero length
        rec[-1] = [3, 4, 0, 0]
        for kitti in (0, 1):
            ref = O.ingest_f32(rec, 0.7, 120.0, kitti)
            t0 = time.perf_counter()
            cid, kept = ctx.cloud_ingest_f32(rec, 0.7, 120.0, kitti)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            out = ctx.cloud_download(cid)
            ok = out.shape == ref.shape and bool(((out == ref) | (np.isnan(out) & np.isnan(ref))).all())
            print(f"[ingest kitti={kitti}] kept {kept}/{rec.shape[0]} (oracle {ref.shape[0]}) {dt*1e3:.2f} ms bit-identical: {ok}"
                  + ("" if ok or out.shape != ref.shape else f" max diff {np.nanmax(np.abs(out-ref)):.3e}"))
            ctx.cloud_release(cid)
    except Exception:  # noqa: BLE001
        print("[ingest] FAILED")
        traceback.print_exc()
    ctx.close()


def synth_pose(tx, ty, yaw, pitch):
    T = np.eye(4)
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    T[:3, :3] = Rz @ Ry
    T[:3, 3] = [tx, ty, 0.01]
    return T


if __name__ == "__main__":
    main()
