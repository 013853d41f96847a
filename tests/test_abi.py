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
