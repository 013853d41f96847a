import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip: the HIP path is the product
    pass


@pytest.fixture(scope="session")
def natives():
    """Build (if stale) and return the in-tree native libraries."""
    from mad_icp_amd import _build

    _build.build_all()
    if not os.environ.get("MADICP_NATIVE_DIR"):  # (a variant run builds into its own directory: nothing else to make)
        _build.build_measure()
    return True


@pytest.fixture(scope="session")
def ctx(natives):
    from mad_icp_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def mctx(natives):
    """A context of the MEASUREMENT build (mad_icp_amd/_measure: the same sources + include/madicp_hip_measure.h's aids) for
    the few tests that look inside a device tree build (madicp_debug_tree_build_points); everything else runs against the
    product library, which exports the drop-in header only."""
    from mad_icp_amd import capi

    mc = capi.measure_variant()
    if not hasattr(mc.hip_lib(), "madicp_debug_tree_build_points"):
        pytest.skip("this build of the library has no measurement aids (a variant run without -DMADICP_MEASURE)")
    c = mc.Context(0)
    yield c
    c.close()
