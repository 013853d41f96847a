"""Phase timing inside icp_round from in-kernel wall-clock stamps (development tool, GPU box only).

Builds a second copy of the HIP library with -DMADICP_STAMPS (never the shipped one), runs one registration of the
bench workload and prints, per GN round, the median over workgroups of the time between stamps (100 MHz clock):
  0 entry | 1 previous round joined | 2 pose solved+broadcast | 3 first pass: reuse decided | 4 first pass: walks done
  5 all passes done | 6 partial written
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mad_icp_amd import _build  # noqa: E402

so = os.environ.get("MADICP_STAMPS_LIB", os.path.join(ROOT, "tools", "libmadicp_hip_stamps.so"))
os.makedirs(os.path.dirname(so), exist_ok=True)
src = os.path.join(_build.CSRC, "hip", "madicp_capi.hip")
# rebuilt whenever the sources changed (a stale copy lacks the current ABI's symbols and capi.hip_lib() refuses it):
# the hash of the sources it was built from sits beside it
want = _build.hip_source_hash()
stamp = so + ".srchash"
have = open(stamp).read().strip() if os.path.exists(stamp) and os.path.exists(so) else ""
if "--build" in sys.argv or "--build-only" in sys.argv or have != want:
    subprocess.check_call([_build.HIPCC] + _build.HIP_FLAGS + ["-DMADICP_STAMPS", "-I" + _build.INC,
                                                            "-I" + os.path.join(_build.CSRC, "hip"), src, "-o", so, "-lrccl"])
    with open(stamp, "w") as f:
        f.write(want)
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["MADICP_HIP_LIB"] = so
from mad_icp_amd import capi, synth  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prob = synth.make_problem(K, seed=0)
ctx = capi.Context(0)
for kv in sys.argv[2:]:
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
s = buf.reshape(16, 256, 16).astype(np.int64)
# stamp order along a workgroup's time line (wave 0): see kernels.hip.h MADICP_STAMP
ORDER = [0, 14, 15, 13, 1, 2, 7, 3, 4, 8, 9, 10, 11, 12, 5, 6]
NAMES = ["join issue", "pose issue", "scalars+prefetch", "stage1+2", "solve", "bcast+init", "p0 reuse", "p0 walk", "p0 record", "p0 math", "p1 loads", "p1 reuse", "p1 record",
         "p1 math", "reduce+store"]
print("| round | " + " | ".join(NAMES) + " | total | whole |")
print("|" + "---|" * (len(NAMES) + 3))
for r in range(15):
    t = s[r][:, ORDER]
    d = np.diff(t, axis=1) / 100.0  # us
    med = np.median(d, axis=0)
    tot = np.median(s[r, :, 6] - s[r, :, 0]) / 100.0
    whole = (s[r, :, 6].max() - s[r, :, 0].min()) / 100.0
    print("| %d | " % r + " | ".join("%.2f" % x for x in med) + " | %.2f | %.2f |" % (tot, whole))
