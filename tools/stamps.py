"""In-kernel phase time stamps (profiling build): per wave, wall_clock64 (100 MHz) at entry / after the prologue /
after the first pass's walk-or-reuse check / after all passes / after the butterfly / after the barrier / at exit."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mad_icp_amd import capi, synth
capi._load_orig = capi._load
capi._load = lambda name: C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmadicp_hip_ablate.so")) if "hip" in name else capi._load_orig(name)
pb = synth.make_problem(16, seed=1)
ctx = capi.Context(0)
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3); ht.transform(T[:3, :3], T[:3, 3]); tids.append(ctx.upload(ht))
h = capi.HostTree(pb["query_scans"][0], 0.2, 0.1, 3)
mid = ctx.moving_upload(h.leaf_means())
P = (0.2, 0.1, 0.02)
lib = capi.hip_lib()
lib.madicp_debug_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
def report(name, n_iters):
    for _ in range(3):
        ctx.icp_register(mid, tids, pb["query_guess"][0], P, n_iters, h.num_leaves)
    buf = np.zeros(768 * 4 * 8, np.uint64)
    lib.madicp_debug_fetch(ctx._h, buf.ctypes.data_as(C.c_void_p), buf.size)
    d = buf.reshape(-1, 8).astype(np.int64)
    d = d[d[:, 0] > 0]
    t0 = d[:, 0].min()
    rel = (d[:, :7] - t0) * 0.01  # us
    names = ["entry", "prologue", "pass1 walk/reuse", "passes done", "butterfly", "barrier", "exit"]
    print("== %s: %d waves; kernel span %.2f us (first entry -> last exit)" % (name, len(d), rel[:, 6].max()))
    for c, nm in enumerate(names):
        print("   %-18s  min %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (nm, rel[:, c].min(), np.median(rel[:, c]), np.percentile(rel[:, c], 90), rel[:, c].max()))
    dur = np.diff(rel, axis=1)
    print("   per-wave phase durations (median): " + ", ".join("%s %.2f" % (names[i + 1], np.median(dur[:, i])) for i in range(6)))
for q in (1, 2):
    ctx.set_option("queries_per_lane", q)
    report("qpt %d walk round (n_iters=1)" % q, 1)
    report("qpt %d converged round (last of 15)" % q, 15)
