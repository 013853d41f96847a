"""A/B of icp_round between builds of the library (MADICP_HIP_LIB=<other build>): avg launch over a 15-round registration
from the perturbed guess, and from the converged pose (no walks after round 0), for the headline problem (16 keyframes, one
scan) and BASELINE configs[4] (64 keyframes, 8 scans in flight).  One line per configuration.
usage (GPU box): [MADICP_HIP_LIB=...] python tools/ab_round.py [label]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mad_icp_amd import capi, synth  # noqa: E402
capi = capi.measure_variant()  # (uses include/madicp_hip_measure.h's aids: the measurement build, mad_icp_amd/_measure)

PARAMS = (0.2, 0.1, 0.02)
label = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("MADICP_HIP_LIB", "in-tree")
groups = sys.argv[2:] or [""]  # option sets "key=value,key=value": one block of lines each (options stay set)
pb = synth.make_problem(64, seed=1, n_queries=1)
scans, gts, guesses = synth.make_query_streams(64, seed=1, n_streams=8)
s16, g16, gs16 = synth.make_query_streams(16, seed=1, n_streams=1)
scans16, gts16, guesses16 = synth.make_query_streams(16, seed=1, n_streams=8)
ctx = capi.Context(0)
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3)
    ht.transform(T[:3, :3], T[:3, 3])
    tids.append(ctx.upload(ht))
for group in groups:
    for kv in filter(None, group.split(",")):
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    for name, K, sc, gu in (("K16 x 1 scan ", 16, s16, gs16), ("K16 x 8 scans", 16, scans16, guesses16), ("K64 x 8 scans", 64, scans, guesses)):
        qts = [capi.HostTree(s, 0.2, 0.1, 3) for s in sc]
        mids = [ctx.moving_upload(q.leaf_means()) for q in qts]
        X0 = np.stack([capi.pose12(T) for T in gu])
        r = ctx.icp_register_batch(mids, tids[:K], X0, PARAMS, 15)
        out = []
        for rep in range(3):
            a, _, _, walked = ctx.icp_time_registration(mids, tids[:K], X0, PARAMS, 15, reps=20)
            out.append(a)
        print("%-10s %-28s %s: avg launch %.2f us   %.0f registrations/s resident   checksum %.12f"
              % (label, group or "defaults", name, min(out), len(mids) / (15 * min(out) * 1e-6 + 7e-6), float(np.abs(r["X"]).sum())), flush=True)
        for m in mids:
            ctx.moving_release(m)
