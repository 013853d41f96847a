"""The device tree builder's xor butterflies run over the VALU (v_permlane32_swap / v_permlane16_swap, DPP row_ror / row_shl /
row_shr / quad_perm: tree_build.hip.h, xor_fetch) instead of the LDS crossbar.  The claim is "same partners, same order, same
operands, hence the same bits": this test builds the HIP library a second time with -DMADICP_TB_BPERMUTE=1 (every butterfly
back on __shfl_xor) into a scratch directory and compares the node arrays of both builds of the same clouds byte for byte —
full-size scans (all four regimes), a cloud at the regime boundaries, degenerate ones."""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import hashlib
import numpy as np
from mad_icp_amd import _build
_build.build_hip(); _build.build_host()
from mad_icp_amd import capi, synth
ctx = capi.Context(0)
rng = np.random.default_rng(5)
pb = synth.make_problem(2, seed=4, n_queries=1)
clouds = [pb["query_scans"][0]] + list(pb["keyframe_scans"]) + [
    rng.normal(size=(513, 3)) * [5, 3, 0.05], rng.normal(size=(2049, 3)) * [5, 3, 1.0], rng.normal(size=(33, 3)),
    np.repeat(np.array([[1.0, 2.0, 3.0]]), 40, axis=0), np.stack([np.linspace(0, 10, 100), np.zeros(100), np.zeros(100)], 1)]
h = hashlib.sha256()
for c in clouds:
    cid = ctx.cloud_upload(c)
    tid, nl = ctx.tree_build(cid, 0.2, 0.1)
    nodes = ctx.tree_download(tid, 2 * nl - 1)
    h.update(np.ascontiguousarray(nodes).tobytes())
    h.update(ctx.tree_build_points(c.shape[0]).tobytes())
    ctx.tree_release(tid); ctx.cloud_release(cid)
print("DIGEST", h.hexdigest(), len(clouds))
ctx.close()
"""


def _digest(env):
    r = subprocess.run([sys.executable, "-c", CODE], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1]
    return line.split()[1]


@pytest.mark.gpu
def test_valu_butterflies_give_the_bits_of_the_lds_crossbar_ones(natives, tmp_path):
    base = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")] + sys.path))
    d = str(tmp_path)
    os.makedirs(os.path.join(d, "pybind"), exist_ok=True)
    # (both legs with the measurement aids: madicp_debug_tree_build_points, the member order the construction left)
    from mad_icp_amd import _build

    variant = dict(base, MADICP_NATIVE_DIR=d, MADICP_EXTRA_DEFINES="-DMADICP_TB_BPERMUTE=1 -DMADICP_MEASURE")
    assert _digest(dict(base, MADICP_NATIVE_DIR=_build.MEASURE_DIR, MADICP_EXTRA_DEFINES="-DMADICP_MEASURE")) == _digest(variant)
