"""The reference's own example scripts, executed LITERALLY against the drop-in modules
(apps/utils/tools/nn_search.py:36-61 and mad_registration.py:48-69 — the only known-answer material the reference
ships: NN self-query error 0, pairwise registration -> identity).

The files are never part of this repository: they are read from /root/reference where it exists (this container) and
otherwise from oracle/_ref/tools/ — byte-for-byte copies that __graft_entry__.build() places there (oracle/ship_ref_tools.sh;
oracle/_ref/ is git-ignored and travels to the GPU box with the snapshot, like oracle/_ref/bin_runner).
`mad_icp.src.pybind.*` resolves to THIS repository's modules, `mad_icp.apps.*` to the reference's files (their package
directory is appended to `mad_icp.__path__`), `open3d` — which mad_registration.py imports at the top but only uses with
--viz — is an empty stub.  Needs a GPU: every search and registration runs on the HIP path."""
import contextlib
import io
import os
import re
import runpy
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/mad_icp"
if not os.path.isfile(os.path.join(REF, "apps", "utils", "tools", "nn_search.py")):
    REF = os.path.join(ROOT, "oracle", "_ref", "tools", "mad_icp")  # shipped by oracle/ship_ref_tools.sh
TOOLS = os.path.join(REF, "apps", "utils", "tools")

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.isfile(os.path.join(TOOLS, "nn_search.py")), reason="neither /root/reference nor oracle/_ref/tools present")]


@pytest.fixture()
def reference_apps(natives):
    from mad_icp_amd import _build

    _build.build_pybind()
    import mad_icp

    saved_path = list(mad_icp.__path__)
    saved_mods = {k: v for k, v in sys.modules.items() if k == "open3d" or k.startswith("mad_icp.apps")}
    if REF not in mad_icp.__path__:
        mad_icp.__path__.append(REF)  # after this repository's directory: mad_icp.src stays ours, mad_icp.apps is theirs
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    yield
    mad_icp.__path__[:] = saved_path
    for k in [k for k in sys.modules if k == "open3d" or k.startswith("mad_icp.apps")]:
        if k not in saved_mods:
            del sys.modules[k]


def _run(script, argv):
    out = io.StringIO()
    old_argv = sys.argv
    sys.argv = [script] + argv
    code = 0
    try:
        with contextlib.redirect_stdout(out):
            try:
                runpy.run_path(script, run_name="__main__")
            except SystemExit as e:  # mad_registration.py leaves through exit(0)
                code = e.code or 0
    finally:
        sys.argv = old_argv
    assert code == 0, out.getvalue()[-2000:]
    return out.getvalue()


def test_nn_search_script_runs_unchanged(reference_apps):
    import mad_icp.src.pybind.pymadtree as ours

    assert os.path.dirname(os.path.abspath(ours.__file__)).startswith(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    text = _run(os.path.join(TOOLS, "nn_search.py"), [])
    errs = [float(x) for x in re.findall(r"error in matching ([0-9.eE+-]+)", text)]
    assert len(errs) == 2, text[-1500:]
    assert errs[0] == 0.0 and errs[1] == 0.0  # apps/utils/tools/README.md: "the error should be 0"


def test_mad_registration_script_runs_unchanged(reference_apps):
    text = _run(os.path.join(TOOLS, "mad_registration.py"), [])
    m = re.search(r"estimate\s*\n(.*)", text, flags=re.S)
    assert m, text[-1500:]
    vals = [float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", m.group(1))][:16]
    T = np.array(vals).reshape(4, 4)
    assert np.abs(T - np.eye(4)).max() < 1e-6  # "gt T = identity"; the reference states no tolerance
