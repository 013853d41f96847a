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


@pytest.mark.parametrize("header,lib", [("madicp_hip.h", "libmadicp_hip.so"), ("madicp_hip_measure.h", "libmadicp_hip.so"),
                                        ("madicp_host.h", "libmadicp_host.so")])
def test_library_exports_every_declared_symbol(natives, header, lib):
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
    with pytest.raises(capi.MadIcpError):
        capi.Context(0)


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


def test_product_build_leaves_the_measurement_and_test_aids_out(natives, tmp_path):
    """-DMADICP_NO_MEASURE (MADICP_EXTRA_DEFINES, built into a scratch directory through MADICP_NATIVE_DIR): the library still
    exports every symbol of the drop-in header and NONE of include/madicp_hip_measure.h, and pypeline is compiled without the
    realtime rule's test seam.  (The in-tree build keeps them: bench.py and the tests call them.)"""
    import subprocess
    import sys

    d = str(tmp_path)
    env = dict(os.environ, MADICP_NATIVE_DIR=d, MADICP_EXTRA_DEFINES="-DMADICP_NO_MEASURE",
               PYTHONPATH=os.pathsep.join([ROOT] + sys.path))
    r = subprocess.run([sys.executable, "-c", "from mad_icp_amd import _build; print(_build.build_hip())"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(d, "libmadicp_hip.so")], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\b(madicp_[a-z0-9_]+)\b", out))
    missing = [s for s in declared_symbols("madicp_hip.h") if s not in exported]
    assert not missing, missing
    leaked = [s for s in declared_symbols("madicp_hip_measure.h") if s in exported]
    assert not leaked, leaked
    assert not [s for s in exported if s.startswith("madicp_debug_")]
    # the binding: the preprocessed translation unit no longer registers the seam
    import pybind11

    pp = subprocess.run(["g++", "-E", "-std=c++17", "-DMADICP_NO_MEASURE", "-I" + os.path.join(ROOT, "include"),
                         "-I" + os.path.join(ROOT, "mad_icp_amd", "csrc", "host"), "-I" + os.path.join(ROOT, "mad_icp_amd", "csrc", "pybind"),
                         "-I" + pybind11.get_include(), "-I" + __import__("sysconfig").get_paths()["include"],
                         os.path.join(ROOT, "mad_icp_amd", "csrc", "pybind", "pypeline.cpp")], capture_output=True, text=True)
    assert pp.returncode == 0, pp.stderr[-2000:]
    assert '"setTimingForTest"' not in pp.stdout and '"compute"' in pp.stdout
