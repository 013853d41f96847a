"""The one version-dependent evaluation order on the path — Eigen's reduction of a contiguous 3-vector, packet order
(a0 b0 + a1 b1) + a2 b2 in 3.4 with SSE2 vs scalar order a0 b0 + (a1 b1 + a2 b2) — is isolated behind
-DMADICP_REDUX_SCALAR_ONLY in the oracle (oracle/linalg.h), the host classes (csrc/common/eig3.h, csrc/host/linalg.h)
and the kernels (csrc/hip/kernels.hip.h).  These tests build the WHOLE stack a second time with the flag, into a
scratch directory, and run the parity suites against it: product and oracle move in lock-step, so whichever order the
reference binary turns out to use, the switch is one define.

Each leg runs in a subprocess whose environment points the loaders at the variant build (MADICP_NATIVE_DIR,
MADICP_ORACLE_DIR, MADICP_EXTRA_DEFINES)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAG = "-DMADICP_REDUX_SCALAR_ONLY"


def _variant_env(tmp):
    d = str(tmp)
    os.makedirs(os.path.join(d, "pybind"), exist_ok=True)
    env = dict(os.environ, MADICP_NATIVE_DIR=d, MADICP_ORACLE_DIR=d, MADICP_EXTRA_DEFINES=FLAG,
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")] + sys.path))
    return env


def _run(env, code):
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


BUILD = ("from mad_icp_amd import _build; _build.build_hip(); _build.build_host(); "
         "import oracle_lib as O; O.build()\n")


@pytest.fixture(scope="module")
def variant(tmp_path_factory):
    env = _variant_env(tmp_path_factory.mktemp("redux_scalar"))
    _run(env, BUILD)
    return env


def test_flag_changes_the_arithmetic_in_lockstep(variant):
    """CPU leg: with the flag, the host tree builder is still bit-identical to the oracle's (both switched), and the
    trees differ from the default build's (the flag is not a no-op)."""
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
    for k in ("MADICP_NATIVE_DIR", "MADICP_ORACLE_DIR", "MADICP_EXTRA_DEFINES"):
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
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_parity.py",
                        "-k", "nn_search or linearize or register", "-p", "no:cacheprovider"],
                       env=variant, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
