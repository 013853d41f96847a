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
    stubbed = ("open3d", "natsort", "rosbags", "rosbags.typesys", "rosbags.typesys.types")
    theirs = ("mad_icp.apps", "mad_icp.configurations")
    saved_mods = {k: v for k, v in sys.modules.items() if k in stubbed or k.startswith(theirs)}
    if REF not in mad_icp.__path__:
        mad_icp.__path__.append(REF)  # after this repository's directory: mad_icp.src stays ours, mad_icp.apps is theirs
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    if "natsort" not in sys.modules:  # (not in this image; the readers only call natsorted on file lists)
        try:
            import natsort  # noqa: F401
        except ModuleNotFoundError:
            ns = types.ModuleType("natsort")
            ns.natsorted = sorted
            sys.modules["natsort"] = ns
    if "rosbags" not in sys.modules:  # (not in this image either; apps/utils/point_cloud2.py — imported by the rosbag readers the
        try:                          # launcher imports at its top — wants the two message classes and PointField's type codes)
            import rosbags  # noqa: F401
        except ModuleNotFoundError:
            rb, ts, ty = types.ModuleType("rosbags"), types.ModuleType("rosbags.typesys"), types.ModuleType("rosbags.typesys.types")
            ty.sensor_msgs__msg__PointCloud2 = type("sensor_msgs__msg__PointCloud2", (), {})
            ty.sensor_msgs__msg__PointField = type("sensor_msgs__msg__PointField", (), dict(
                INT8=1, UINT8=2, INT16=3, UINT16=4, INT32=5, UINT32=6, FLOAT32=7, FLOAT64=8))  # sensor_msgs/PointField.msg
            rb.typesys, ts.types = ts, ty
            sys.modules.update({"rosbags": rb, "rosbags.typesys": ts, "rosbags.typesys.types": ty})
    yield
    mad_icp.__path__[:] = saved_path
    for k in [k for k in sys.modules if k in stubbed or k.startswith(theirs)]:
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


# ---- the reference's Python LAUNCHER, unchanged (apps/mad_icp.py:49,99-204 + apps/utils/kitti_reader.py:73-94) -------------------
LAUNCHER = os.path.join(REF, "apps", "mad_icp.py")


def _write_kitti_sequence(directory, n_frames, jitter=0.0, seed=0):
    """a KITTI-layout sequence of synthetic street scans: %06d.bin, (n, 4) float32 records (x, y, z, remission)"""
    from mad_icp_amd import synth

    scene = synth.Scene(4)
    rng = np.random.default_rng(seed)
    os.makedirs(directory, exist_ok=True)
    for i in range(n_frames):
        s = synth.render_scan(scene, synth.path_pose(0.8 * i), 300 + i, n_beams=32, n_azimuth=600)
        if jitter:
            s = s + rng.normal(scale=jitter, size=s.shape)
        rec = np.zeros((s.shape[0] + 2, 4), np.float32)
        rec[:-2, :3] = s.astype(np.float32)
        rec[:-2, 3] = 0.3
        rec[-2] = [0.1, 0.1, 0.1, 0]   # below min_range: the reader drops it (kitti_reader.py:86-89)
        rec[-1] = [300, 0, 0, 0]       # beyond max_range
        rec.tofile(os.path.join(directory, "%06d.bin" % i))


def _launch(data_dir, out_dir, dataset):
    """typer CLI of the reference's launcher, in process: options as a user types them"""
    text = _run(LAUNCHER, ["--data-path", str(data_dir), "--estimate-path", str(out_dir), "--dataset-config", dataset,
                           "--num-cores", "4", "--num-keyframes", "4", "--noviz"])
    return text, np.loadtxt(os.path.join(str(out_dir), "estimate.txt")).reshape(-1, 3, 4)


def _reader_clouds(data_dir, conf):
    """what the launcher's own reader hands to Pipeline.compute: (stamp, cloud) per frame — the reference's KittiReader itself"""
    from pathlib import Path

    from mad_icp.apps.utils.kitti_reader import KittiReader

    out = []
    with KittiReader(Path(str(data_dir)), conf["min_range"], conf["max_range"], topic=None, sensor_hz=conf["sensor_hz"],
                     apply_correction=conf.get("apply_correction", False)) as reader:
        for ts, pts in reader:
            out.append((ts, np.ascontiguousarray(pts, dtype=np.float64)))
    return out


@pytest.mark.skipif(not os.path.isfile(LAUNCHER), reason="the launcher was not shipped (oracle/ship_ref_tools.sh)")
def test_python_launcher_runs_unchanged_on_a_kitti_sequence(reference_apps, tmp_path):
    """`mad_icp --data-path ... --estimate-path ... --dataset-config kitti --noviz` — the reference's launcher, its KittiReader
    (float32 range filter, the KITTI correction), its configuration tables and its estimate writer, all unchanged — against this
    repository's `mad_icp.src.pybind.pypeline`: estimate.txt equals the oracle pipeline's base-frame poses on the very clouds
    the reader produced, to 1e-5 per coefficient (kitti: deskew False -> the default device front-end)."""
    import oracle_lib as O
    from mad_icp.configurations.datasets.dataset_configurations import DatasetConfiguration_lut
    from mad_icp.configurations.mad_params import MADConfiguration_lut
    import mad_icp.src.pybind.pypeline as ours

    assert os.path.abspath(ours.__file__).startswith(ROOT)
    n_frames = 8
    data = tmp_path / "velodyne"
    _write_kitti_sequence(str(data), n_frames)
    text, est = _launch(data, tmp_path / "out", "kitti")
    assert "The dataset is in kitti format" in text and "Loading frame # %d" % (n_frames - 1) in text
    assert est.shape[0] == n_frames
    conf, mp = DatasetConfiguration_lut["kitti"], MADConfiguration_lut["default"]
    assert conf["deskew"] is False and conf["apply_correction"] is True
    L2B = np.array(conf["lidar_to_base"], dtype=np.float64)
    op = O.Pipeline(conf["sensor_hz"], conf["deskew"], mp["b_max"], mp["rho_ker"], mp["p_th"], mp["b_min"], mp["b_ratio"], 4, 4, False)
    worst = 0.0
    for i, (ts, cloud) in enumerate(_reader_clouds(data, conf)):
        op.compute(ts, cloud)
        want = (L2B @ op.currentPose() @ np.linalg.inv(L2B))[:3]  # apps/utils/utils.py:31-34
        d = np.abs(est[i] - want).max()
        worst = max(worst, d)
        assert d <= 1e-5, (i, d)
    print("python launcher (kitti) vs oracle pipeline: worst coefficient difference %.2e over %d frames" % (worst, n_frames))


@pytest.mark.skipif(not os.path.isfile(LAUNCHER), reason="the launcher was not shipped (oracle/ship_ref_tools.sh)")
def test_python_launcher_runs_unchanged_on_a_deskewed_dataset(reference_apps, tmp_path):
    """The same with `--dataset-config mulran` (deskew True, no correction: configurations/datasets/dataset_configurations.py):
    the unmodified launcher deskews and builds on the device since round 6.  Held inside three times the envelope the oracle
    pipeline shows against itself on the reader's clouds, and to 1e-5 on the two frames that are not deskewed."""
    import envelope as E
    from mad_icp.configurations.datasets.dataset_configurations import DatasetConfiguration_lut

    n_frames = 10
    data = tmp_path / "velodyne"
    _write_kitti_sequence(str(data), n_frames, jitter=1e-4, seed=3)  # (no two float32 azimuths equal: a real sensor's noise)
    text, est = _launch(data, tmp_path / "out", "mulran")
    assert est.shape[0] == n_frames
    conf = DatasetConfiguration_lut["mulran"]
    assert conf["deskew"] is True
    L2B = np.array(conf["lidar_to_base"], dtype=np.float64)
    clouds = [c for _, c in _reader_clouds(data, conf)]
    base, _, dt, da = E.self_envelope(clouds, deskew=True, base_threads=4, num_keyframes=4)
    bound = E.running_bound(E.combined(dt, da))
    Linv = np.linalg.inv(L2B)
    for i in range(n_frames):
        got = Linv @ np.vstack([est[i], [0, 0, 0, 1]]) @ L2B
        d_t, d_a = E.pose_dev(base[i], got)
        assert E.combined(d_t, d_a) <= 3.0 * bound[i] + 2e-5, (i, d_t, d_a, bound[i])
        if i < 2:
            assert d_t <= 1e-5 and d_a <= 1e-5, (i, d_t, d_a)
