"""-DMADICP_EXACT_SOLVE: the build that undoes the registration's deliberate arithmetic deviations — after the gate the
default kernels use fused multiply-adds, refined reciprocals and a lane-parallel Gauss-Jordan solve;
with the flag e, J, the weights, the 27 accumulations and the 6x6 LDLT run in the reference's operation order with correctly
rounded divisions (mad_icp.cpp:59-72, 92-101, 111).  This test builds the HIP library a second time with the flag, into a
scratch directory, and runs the registration parity tests against it: the leg stays buildable and green, and the default
build is shown to differ from it only in the last bits."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_exact_solve_build_passes_the_registration_parity_tests(tmp_path):
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "pybind"), exist_ok=True)
    env = dict(os.environ, MADICP_NATIVE_DIR=d, MADICP_EXTRA_DEFINES="-DMADICP_EXACT_SOLVE",
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")] + sys.path))
    env.pop("MADICP_ORACLE_DIR", None)  # the oracle is the default one: the flag changes nothing it computes
    r = subprocess.run([sys.executable, "-c", "from mad_icp_amd import _build; _build.build_hip(); _build.build_host()"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_parity.py", "-k",
                        "linearize or register or reuse", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
    # and the two builds agree to the last bits on a registration (same decisions, arithmetic after the gate differs)
    code = r"""
import numpy as np
from mad_icp_amd import capi
from fixtures import street_problem, B_MAX, B_MIN, PARAMS
pb = street_problem(2)
ctx = capi.Context(0)
tids = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    ht = capi.HostTree(s, B_MAX, B_MIN, 2); ht.transform(T[:3, :3], T[:3, 3]); tids.append(ctx.upload(ht))
qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
mid = ctx.moving_upload(qh.leaf_means())
r = ctx.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
np.save(OUT, np.concatenate([r["X"], r["H"].reshape(-1), r["matched"].astype(np.float64)]))
"""
    outs = []
    for tag, e in (("exact", env), ("default", {k: v for k, v in env.items() if k not in ("MADICP_NATIVE_DIR", "MADICP_EXTRA_DEFINES")})):
        out = os.path.join(d, tag + ".npy")
        r = subprocess.run([sys.executable, "-c", "OUT = %r\n" % out + code], env=e, cwd=ROOT, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(out)
    import numpy as np

    a, b = np.load(outs[0]), np.load(outs[1])
    assert np.array_equal(a[48:], b[48:])  # matched flags: the decisions are the same
    assert np.abs(a[:12] - b[:12]).max() <= 1e-12  # poses agree to the last bits
    assert np.allclose(a[12:48], b[12:48], rtol=1e-10, atol=1e-10 * np.abs(b[12:48]).max())
    assert not np.array_equal(a[12:48], b[12:48])  # ... and the flag is not a no-op
