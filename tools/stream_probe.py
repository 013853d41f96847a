"""Host-side timing of the streamed registration path (development tool, GPU box): per-call wall time of
madicp_stream_submit / madicp_stream_collect at pipeline depths 0..2, own stream vs a torch stream."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mad_icp_amd import capi, synth

use_torch = "--torch" in sys.argv
K = 16
pb = synth.make_problem(K, seed=1, n_queries=1)
scans, gts, gs = synth.make_query_streams(K, seed=1, n_streams=8)
if use_torch:
    import torch
    st = torch.cuda.Stream()
    ctx = capi.Context(0, st.cuda_stream)
else:
    ctx = capi.Context(0)
if "--eager" in sys.argv:
    ctx.set_option("use_graph", 0)
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=")
        ctx.set_option(k, int(v))
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3); ht.transform(T[:3, :3], T[:3, 3]); tids.append(ctx.upload(ht))
leaves = [capi.HostTree(s, 0.2, 0.1, 3).leaf_means() for s in scans]
guess = [capi.pose12(T) for T in gs]
P = (0.2, 0.1, 0.02)
NQ = 8
for depth in (1, 0, 1):
    ts, tc, tstep = [], [], []
    pend = []
    n = 400
    t0 = time.perf_counter()
    last = t0
    for i in range(n):
        a = time.perf_counter()
        tk = ctx.stream_submit(leaves[i % NQ], tids, guess[i % NQ], P, 15)
        b = time.perf_counter()
        pend.append((tk, leaves[i % NQ].shape[0]))
        ts.append(b - a)
        while len(pend) > depth:
            c = time.perf_counter()
            ctx.stream_collect(*pend.pop(0))
            tc.append(time.perf_counter() - c)
        now = time.perf_counter()
        tstep.append(now - last)
        last = now
    while pend:
        ctx.stream_collect(*pend.pop(0))
    el = time.perf_counter() - t0
    tstep = np.array(tstep) * 1e6
    print("torch=%d depth %d: %.1f us/step  submit median %.1f us (p90 %.1f)  collect median %.1f us | first50 %.1f mid %.1f last50 %.1f" % (
        use_torch, depth, el / n * 1e6, np.median(ts) * 1e6, np.percentile(ts, 90) * 1e6, np.median(tc) * 1e6,
        tstep[:50].mean(), tstep[150:250].mean(), tstep[-50:].mean()))
    print("   submit max %.0f us at %d ; collect max %.0f us at %d" % (np.max(ts) * 1e6, int(np.argmax(ts)), np.max(tc) * 1e6, int(np.argmax(tc))))
