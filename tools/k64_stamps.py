"""Where a workgroup's time goes in BASELINE configs[4] (64 keyframes, 8 scans in flight), from in-kernel wall-clock stamps
(development tool, GPU box only; the -DMADICP_STAMPS copy of the library that tools/stamps.py builds): per GN round, over the 32
workgroups of scan 0 —
  the phases of the first unit's first two passes (tree-major rounds only: reuse, walk, leaf record, math), the time from the
  solved pose to the end of the workgroup's passes (median, min, max: what is left of the imbalance), and the launch as a whole."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mad_icp_amd import _build  # noqa: E402

so = os.environ.get("MADICP_STAMPS_LIB", os.path.join(ROOT, "tools", "libmadicp_hip_stamps.so"))
want = _build.hip_source_hash()
stamp = so + ".srchash"
have = open(stamp).read().strip() if os.path.exists(stamp) and os.path.exists(so) else ""
if have != want:
    subprocess.check_call([_build.HIPCC] + _build.HIP_FLAGS + ["-DMADICP_STAMPS", "-I" + _build.INC, "-I" + os.path.join(_build.CSRC, "hip"),
                           os.path.join(_build.CSRC, "hip", "madicp_capi.hip"), "-o", so, "-lrccl"])
    open(stamp, "w").write(want)
os.environ["MADICP_HIP_LIB"] = so
from mad_icp_amd import capi, synth  # noqa: E402

K, B = 64, 8
PARAMS = (0.2, 0.1, 0.02)
pb = synth.make_problem(K, seed=1, n_queries=1)
scans, gts, guesses = synth.make_query_streams(K, seed=1, n_streams=B)
ctx = capi.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, 0.2, 0.1, 3)
    ht.transform(T[:3, :3], T[:3, 3])
    tids.append(ctx.upload(ht))
mids = [ctx.moving_upload(capi.HostTree(s, 0.2, 0.1, 3).leaf_means()) for s in scans]
X0 = np.stack([capi.pose12(T) for T in guesses])
ctx.set_option("use_graph", 0)
for _ in range(3):
    ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
buf = np.zeros(16 * 256 * 16, dtype=np.uint64)
lib = capi.hip_lib()
lib.madicp_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert lib.madicp_debug_stamps(ctx._h, buf.ctypes.data) == 0
s = buf.reshape(16, 256, 16).astype(np.int64)[:, :32, :]  # (32 workgroups per scan at 8 scans in flight)
print("| round | entry -> pose solved | first pass of the first tree: reuse decided -> walks done (tree-major rounds) | "
      "pose -> passes done: median | min | max | passes done (wave 0) -> partial stored | launch (first entry -> last store) |")
print("|" + "---|" * 8)
for r in range(15):
    t = s[r]
    us = lambda a, b: np.median(t[:, b] - t[:, a]) / 100.0
    body = (t[:, 5] - t[:, 2]) / 100.0
    walk = us(3, 4)
    print("| %d | %.1f | %s | %.1f | %.1f | %.1f | %.1f | %.1f |" % (
        r, us(0, 2), ("%.2f" % walk) if 0.0 < walk < 1e4 and r < 2 else "-", np.median(body), body.min(), body.max(), us(5, 6),
        (t[:, 6].max() - t[:, 0].min()) / 100.0))
