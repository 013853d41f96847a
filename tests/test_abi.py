"""CPU tests of the drop-in boundary: the C-ABI libraries load and export every symbol include/*.h declares;
without a GPU the HIP entry points fail loudly (no CPU fallback exists)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(madicp_[a-z0-9_]+)\s*\(", src)))


@pytest.mark.parametrize("header,lib", [("madicp_hip.h", "libmadicp_hip.so"), ("madicp_hip_measure.h", "_measure/libmadicp_hip.so"),
                                        ("madicp_hip.h", "_measure/libmadicp_hip.so"), ("madicp_host.h", "libmadicp_host.so")])
def test_library_exports_every_declared_symbol(natives, header, lib):
    """(the measurement aids: exported by the measurement build in mad_icp_amd/_measure only — see the last test of this file)"""
    L = ctypes.CDLL(os.path.join(ROOT, "mad_icp_amd", lib))
    syms = declared_symbols(header)
    assert len(syms) >= 6
    for s in syms:
        assert hasattr(L, s), f"{lib} does not export {s} declared in include/{header}"


def test_node_record_is_64_bytes(natives):
    from mad_icp_amd import capi

    assert capi.NODE_DTYPE.itemsize == 64
    assert capi.NODE_DTYPE.fields["right"][1] == 48 and capi.NODE_DTYPE.fields["bbox0"][1] == 56


def test_no_cpu_fallback(natives):
    """On a box without a GPU, creating a context must raise — never silently compute on the CPU."""
    import torch

    from mad_icp_amd import capi

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked tests")
    try:
        c = capi.Context(0)
    except capi.MadIcpError:
        return  # no device: the call fails loudly, as it must
    c.close()
    pytest.skip("a HIP device is present although torch does not see one: covered by the gpu-marked tests")


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under mad_icp_amd/ (or include/) may reference it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "mad_icp_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle_lib|libmad_oracle|oracle/|mad_oracle", txt):
                    # documentation pointers to oracle/linalg.h are allowed in comments only
                    code = re.sub(r"//.*|#.*|/\*.*?\*/|\"\"\".*?\"\"\"", "", txt, flags=re.S)
                    if re.search(r"oracle_lib|libmad_oracle|mad_oracle", code):
                        bad.append(os.path.join(base, f))
    assert not bad, bad


def test_product_build_leaves_the_measurement_and_test_aids_out(natives):
    """The DEFAULT build is the product: the in-tree libmadicp_hip.so exports every symbol of the drop-in header and NONE of
    include/madicp_hip_measure.h, and the in-tree pypeline has the reference's surface without the realtime rule's test seam.
    The aids live in the MEASUREMENT build of the same sources (mad_icp_amd/_measure, -DMADICP_MEASURE: bench.py's roofline
    legs, four tests), which exports them all."""
    import subprocess

    from mad_icp_amd import _build

    def exported_by(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
        return set(re.findall(r"\b(madicp_[a-z0-9_]+)\b", out))

    product = exported_by(os.path.join(_build.PKG, "libmadicp_hip.so"))
    missing = [s for s in declared_symbols("madicp_hip.h") if s not in product]
    assert not missing, missing
    leaked = [s for s in declared_symbols("madicp_hip_measure.h") if s in product]
    assert not leaked, leaked
    assert not [s for s in product if s.startswith("madicp_debug_")]
    measure = exported_by(os.path.join(_build.MEASURE_DIR, "libmadicp_hip.so"))
    assert not [s for s in declared_symbols("madicp_hip.h") + declared_symbols("madicp_hip_measure.h") if s not in measure]
    # the bindings: the seam is registered by the measurement build's translation unit only
    import pybind11

    def preprocessed(extra):
        pp = subprocess.run(["g++", "-E", "-std=c++17"] + extra + ["-I" + os.path.join(ROOT, "include"),
                             "-I" + os.path.join(ROOT, "mad_icp_amd", "csrc", "host"), "-I" + os.path.join(ROOT, "mad_icp_amd", "csrc", "pybind"),
                             "-I" + pybind11.get_include(), "-I" + __import__("sysconfig").get_paths()["include"],
                             os.path.join(ROOT, "mad_icp_amd", "csrc", "pybind", "pypeline.cpp")], capture_output=True, text=True)
        assert pp.returncode == 0, pp.stderr[-2000:]
        return pp.stdout

    assert '"setTimingForTest"' not in preprocessed([]) and '"compute"' in preprocessed([])
    assert '"setTimingForTest"' in preprocessed(["-DMADICP_MEASURE"])


def test_measure_variant_is_a_second_binding_of_the_same_surface(natives):
    """capi.measure_variant(): this module a second time, bound to mad_icp_amd/_measure — the same C ABI plus the aids; the
    default binding refuses an aid with a message that says where it lives (no silent fallback, no AttributeError)."""
    from mad_icp_amd import _build, capi

    if os.environ.get("MADICP_NATIVE_DIR"):
        pytest.skip("a variant run: the process is already pointed at another build")
    mc = capi.measure_variant()
    assert mc is not capi and mc is capi.measure_variant() and mc.measure_variant() is mc
    assert mc._DIR_OVERRIDE == _build.MEASURE_DIR
    for aid in ("madicp_icp_time_registration", "madicp_debug_gather16", "madicp_debug_tree_build_points"):
        assert hasattr(mc.hip_lib(), aid) and not hasattr(capi.hip_lib(), aid), aid
    for sym in declared_symbols("madicp_hip.h"):
        assert hasattr(mc.hip_lib(), sym) and hasattr(capi.hip_lib(), sym), sym
    with pytest.raises(capi.MadIcpError, match="measure_variant"):
        capi._aid("madicp_icp_time_registration")
    assert mc._aid("madicp_icp_time_registration") is not None
    # host-side classes work from either binding (the tree builder has no aids: same bytes)
    import numpy as np

    pts = np.random.default_rng(0).normal(size=(500, 3))
    a, b = capi.HostTree(pts, 0.2, 0.1, 0), mc.HostTree(pts, 0.2, 0.1, 0)
    assert np.array_equal(a.nodes, b.nodes)
