#!/usr/bin/env python3
"""Where the "tree ready" time of a caller WITHOUT prefetch goes (host clock, no profiler): the scan in pageable host memory ->
madicp_cloud_upload (pinned staging in pieces + H2D) -> madicp_tree_build (levels, summary, emission) -> tree resident.
A different scan every frame (four, cycled) so that nothing is answered from a cache that a real sensor stream would not have."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mad_icp_amd import capi, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
scene = synth.Scene(0)
scans = [synth.render_scan(scene, synth.path_pose(1.0 * i), 100 + i) for i in range(4)]
ctx = capi.Context(0)
for i in range(6):
    cid = ctx.cloud_upload(scans[i % 4]); t, nl = ctx.tree_build(cid, 0.2, 0.1); ctx.synchronize(); ctx.tree_release(t); ctx.cloud_release(cid)
up, bl, tot, cp = [], [], [], []
for i in range(reps):
    s = scans[i % 4]
    t0 = time.perf_counter()
    cid = ctx.cloud_upload(s)
    t1 = time.perf_counter()
    t, nl = ctx.tree_build(cid, 0.2, 0.1)
    t2 = time.perf_counter()
    ctx.synchronize()
    t3 = time.perf_counter()
    up.append(t1 - t0); bl.append(t2 - t1); tot.append(t3 - t0)
    ctx.tree_release(t); ctx.cloud_release(cid)
    d = np.empty_like(s)
    t4 = time.perf_counter(); np.copyto(d, s); cp.append(time.perf_counter() - t4)
med = lambda a: 1e3 * float(np.median(a))
def run(scans_, tag):
    up, bl, tot = [], [], []
    for i in range(reps):
        s = scans_[i % 4]
        t0 = time.perf_counter()
        cid = ctx.cloud_upload(s)
        t1 = time.perf_counter()
        t, nl_ = ctx.tree_build(cid, 0.2, 0.1)
        t2 = time.perf_counter()
        ctx.synchronize()
        t3 = time.perf_counter()
        up.append(t1 - t0); bl.append(t2 - t1); tot.append(t3 - t0)
        ctx.tree_release(t); ctx.cloud_release(cid)
    print("%s: cloud_upload call %.3f ms | tree_build call %.3f ms | scan in host memory -> tree resident %.3f ms" % (tag, med(up), med(bl), med(tot)))


print("N=%d leaves=%d | cloud_upload call %.3f ms | tree_build call %.3f ms | scan in host memory -> tree resident %.3f ms | "
      "(a plain host copy of the scan, for scale: %.3f ms)" % (scans[0].shape[0], nl, med(up), med(bl), med(tot), med(cp)))
ctx.close()
ctx = capi.Context(0)
f32 = [s.astype(np.float32).astype(np.float64) for s in scans]
run(f32, "float32-origin scans (option upload_f32 = 1: floats over PCIe, widened on the device)")
ctx.set_option("upload_f32", 0)
run(f32, "float32-origin scans, upload_f32 = 0 (doubles over PCIe)")
ctx.close()
