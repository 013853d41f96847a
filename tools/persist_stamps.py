"""Phase timing inside icp_persist from in-kernel wall-clock stamps (development tool, GPU box only): like tools/stamps.py.
  0 top of round | 1 the eight folded rows of the previous round have arrived | 2 pose solved + broadcast | 5 passes done
  6 row published | 15 leader: its group's rows have arrived | 14 leader: F[x] published"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mad_icp_amd import _build  # noqa: E402

so = os.environ.get("MADICP_STAMPS_LIB", os.path.join(ROOT, "tools", "libmadicp_hip_stamps.so"))
src = os.path.join(_build.CSRC, "hip", "madicp_capi.hip")
if "--build" in sys.argv or not os.path.exists(so):
    subprocess.check_call([_build.HIPCC] + _build.HIP_FLAGS + ["-DMADICP_STAMPS", "-I" + _build.INC,
                                                            "-I" + os.path.join(_build.CSRC, "hip"), src, "-o", so, "-lrccl"])
    if "--build-only" in sys.argv:
        sys.exit(0)
os.environ["MADICP_HIP_LIB"] = so
from mad_icp_amd import capi, synth  # noqa: E402

K = 16
prob = synth.make_problem(K, seed=0)
ctx = capi.Context(0)
ctx.set_option("persistent", 1)
for kv in sys.argv[1:]:
    if kv.startswith("--"):
        continue
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
tids = []
for scan, T in zip(prob["keyframe_scans"], prob["keyframe_poses"]):
    t = capi.HostTree(scan, 0.2, 0.1, 3)
    t.transform(T[:3, :3], T[:3, 3])
    tids.append(ctx.upload(t))
qt = capi.HostTree(prob["query_scans"][0], 0.2, 0.1, 3)
mid = ctx.moving_upload(qt.leaf_means())
params = (0.2, 0.1, 0.02)
for _ in range(3):
    ctx.icp_register(mid, tids, prob["query_guess"][0], params, 15, qt.num_leaves)
buf = np.zeros(16 * 256 * 16, dtype=np.uint64)
lib = capi.hip_lib()
lib.madicp_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert lib.madicp_debug_stamps(ctx._h, buf.ctypes.data) == 0
s = buf.reshape(16, 256, 16).astype(np.int64) / 100.0  # us
t00 = s[0, :, 0].min()
print("| round | top (first..last wg) | wait rows | sum+solve+bcast | passes | reduce+publish | round (median wg) | leaders: rows arrive after own publish | fold+publish | everyone's next top - last leader publish |")
print("|---|---|---|---|---|---|---|---|---|---|")
for r in range(15):
    top = s[r, :, 0]
    w = np.median(s[r, :, 1] - s[r, :, 0]) if r > 0 else 0.0
    sol = np.median(s[r, :, 2] - (s[r, :, 1] if r > 0 else s[r, :, 0]))
    pas = np.median(s[r, :, 5] - s[r, :, 2])
    pub = np.median(s[r, :, 6] - s[r, :, 5])
    tot = np.median(s[r, :, 6] - s[r, :, 0])
    lead = np.arange(8)
    arr = np.median(s[r, lead, 15] - s[r, lead, 6])
    fold = np.median(s[r, lead, 14] - s[r, lead, 15])
    nxt = (np.median(s[r + 1, :, 1]) - s[r, lead, 14].max()) if r < 14 else 0.0
    print("| %d | %.2f..%.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |" % (
        r, top.min() - t00, top.max() - t00, w, sol, pas, pub, tot, arr, fold, nxt))
print("last publish of a round -> first top of the next: see columns; whole registration %.2f us" % (s[14, :8, 14].max() - t00))
# spread of arrival: when did each workgroup publish, relative to the slowest
for r in (0, 5, 12):
    p = s[r, :, 6]
    print("round %d: publish times relative to first: median %.2f  p90 %.2f  max %.2f us ; slowest wg %d" % (
        r, np.median(p - p.min()), np.percentile(p - p.min(), 90), (p - p.min()).max(), int(np.argmax(p))))
ORDER = [2, 7, 3, 4, 8, 9, 10, 11, 12, 5]
NAMES = ["bcast+init", "p0 reuse", "p0 walk", "p0 record", "p0 math", "p1 loads", "p1 reuse", "p1 record", "p1 math"]
print("| round | " + " | ".join(NAMES) + " |")
for r in range(15):
    d = np.diff(s[r][:, ORDER], axis=1)
    print("| %d | " % r + " | ".join("%.2f" % x for x in np.median(d, axis=0)) + " |")
