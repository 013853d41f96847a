"""The one version-dependent evaluation order on the path — Eigen's reduction of a contiguous 3-vector, packet order
(a0 b0 + a1 b1) + a2 b2 in 3.4 with SSE2 vs scalar order a0 b0 + (a1 b1 + a2 b2) — is isolated behind
-DMADICP_REDUX_SCALAR_ONLY in the oracle (oracle/linalg.h), the host classes (csrc/common/eig3.h, csrc/host/linalg.h)
and the kernels (csrc/hip/kernels.hip.h).  These tests build the WHOLE stack a second time with the flag, into a
scratch directory, and run the parity suites against it: product and oracle move in lock-step, so whichever order the
reference binary turns out to use, the switch is one define.

A second switch of the same kind, -DMADICP_XFORM_HOMOGENEOUS, covers the other likely divergence of a real Eigen build:
Isometry3d * Vector3d evaluated as the 4x4 matrix times the homogeneous 4-vector, ((r0 p0 + r1 p1) + r2 p2) + t, instead of
linear() * p + translation() (oracle/linalg.h apply(); DESIGN.md section 5, candidate (a)).  It touches the registration and
deskew, not the tree build.

Each leg runs in a subprocess whose environment points the loaders at the variant build (MADICP_NATIVE_DIR,
MADICP_ORACLE_DIR, MADICP_EXTRA_DEFINES)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-DMADICP_REDUX_SCALAR_ONLY", "-DMADICP_XFORM_HOMOGENEOUS"]


def _variant_env(tmp, flag):
    d = str(tmp)
    os.makedirs(os.path.join(d, "pybind"), exist_ok=True)
    env = dict(os.environ, MADICP_NATIVE_DIR=d, MADICP_ORACLE_DIR=d, MADICP_EXTRA_DEFINES=flag, MADICP_VARIANT_FLAG=flag,
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")] + sys.path))
    return env


def _run(env, code):
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


# The CPU legs never call into the HIP library — libmadicp_host.so only has to LINK against one — so the variant directory
# gets a copy of the default build's (no stamp beside it: build_hip() in the same directory still sees "stale" and compiles
# the real variant, which is what the GPU leg does first).  Saves two hipcc runs in the CPU suite.
BUILD_CPU = ("import os, shutil\n"
             "from mad_icp_amd import _build\n"
             "dst = os.path.join(os.environ['MADICP_NATIVE_DIR'], 'libmadicp_hip.so')\n"
             "if not os.path.exists(dst): shutil.copy(os.path.join(_build.PKG, 'libmadicp_hip.so'), dst)\n"
             "_build.build_host()\n"
             "import oracle_lib as O; O.build()\n")
BUILD_GPU = "from mad_icp_amd import _build; _build.build_hip(); _build.build_host()\n"


@pytest.fixture(scope="module", params=FLAGS, ids=["redux_scalar", "xform_homogeneous"])
def variant(request, natives, tmp_path_factory):
    env = _variant_env(tmp_path_factory.mktemp("variant"), request.param)
    _run(env, BUILD_CPU)
    return env


def test_flag_changes_the_arithmetic_in_lockstep(variant):
    """CPU leg: with the flag, the host tree builder is still bit-identical to the oracle's (both switched), and the
    trees differ from the default build's (the flag is not a no-op).  The transform switch does not touch the tree build:
    its CPU leg is the oracle's own deskew / registration against the default oracle's (different bits, same answer to
    1e-9) — the host and device sides of that switch are checked on the GPU leg."""
    if variant["MADICP_VARIANT_FLAG"] == "-DMADICP_XFORM_HOMOGENEOUS":
        code = """
import numpy as np, hashlib
import oracle_lib as O
from fixtures import street_problem, B_MAX, B_MIN, RHO_KER, B_RATIO
from mad_icp_amd import synth
pb = street_problem(2)
fixed = []
for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
    f = O.Tree(s, B_MAX, B_MIN, 2); f.transform(T[:3, :3], T[:3, 3]); fixed.append(f)
mv = O.Tree(pb["query_scans"][0], B_MAX, B_MIN, 2)
r = O.icp_register(mv, fixed, pb["query_guess"][0], 15, B_MAX, RHO_KER, B_RATIO, 2)
c, v = O.deskew(pb["query_scans"][0], synth.path_pose(3.0), synth.path_pose(4.0), 10.0)
print("DIGEST", hashlib.sha256(r["T"].tobytes() + c.tobytes()).hexdigest())
print("POSE", " ".join(repr(float(x)) for x in r["T"].ravel()))
"""
        out_v = _run(variant, code)
        env_default = dict(os.environ, PYTHONPATH=variant["PYTHONPATH"])
        for k in ("MADICP_NATIVE_DIR", "MADICP_ORACLE_DIR", "MADICP_EXTRA_DEFINES", "MADICP_VARIANT_FLAG"):
            env_default.pop(k, None)
        out_d = _run(env_default, code)
        assert out_v.split("DIGEST")[1].split()[0] != out_d.split("DIGEST")[1].split()[0]  # not a no-op
        pv = np.array([float(x) for x in out_v.split("POSE")[1].split()])
        pd = np.array([float(x) for x in out_d.split("POSE")[1].split()])
        assert np.abs(pv - pd).max() < 1e-9  # ... and nothing a pose notices (same trees on both sides here)
        return
    out = _run(variant, """
import numpy as np, hashlib
import oracle_lib as O
from mad_icp_amd import capi
from fixtures import street_problem, B_MAX, B_MIN
pb = street_problem(2)
h = hashlib.sha256()
for s in pb["keyframe_scans"]:
    ht = capi.HostTree(s, B_MAX, B_MIN, 2)
    ot = O.Tree(s, B_MAX, B_MIN, 2)
    ex = ot.export()
    assert ht.num_leaves == ot.num_leaves
    assert np.array_equal(ht.nodes["mean"], ex["mean"])
    h.update(ht.nodes.tobytes())
print("DIGEST", h.hexdigest())
""")
    digest_variant = out.split("DIGEST")[1].strip()
    env_default = dict(os.environ, PYTHONPATH=variant["PYTHONPATH"])
    for k in ("MADICP_NATIVE_DIR", "MADICP_ORACLE_DIR", "MADICP_EXTRA_DEFINES", "MADICP_VARIANT_FLAG"):
        env_default.pop(k, None)
    out = _run(env_default, """
import hashlib
from mad_icp_amd import _build, capi
from fixtures import street_problem, B_MAX, B_MIN
_build.build_host()
pb = street_problem(2)
h = hashlib.sha256()
for s in pb["keyframe_scans"]:
    h.update(capi.HostTree(s, B_MAX, B_MIN, 2).nodes.tobytes())
print("DIGEST", h.hexdigest())
""")
    assert out.split("DIGEST")[1].strip() != digest_variant


@pytest.mark.gpu
def test_gpu_parity_suite_passes_with_the_flag(variant):
    """GPU leg: the bit-exact correspondence / gate / pose parity tests against the variant oracle, variant kernels."""
    _run(variant, BUILD_GPU)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_parity.py",
                        "tests/test_gpu_frontend.py", "-k", "nn_search or linearize or register or deskew_matches_oracle",
                        "-p", "no:cacheprovider"],
                       env=variant, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
