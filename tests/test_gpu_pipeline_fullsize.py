"""End-to-end Pipeline::compute at full scan size (119 725-point scans), product vs the CPU oracle pipeline
(pipeline.cpp:125-265): pose agreement at every frame, identical keyframe decisions, and the frame rate the product
sustains with and without the additive look-ahead (prefetch).  Plus the C++ caller of tests/cpp/runner_surface.cpp —
bin_runner's calls — compiled against csrc/host/pipeline.h and run."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import oracle_lib as O
from fixtures import B_MAX, B_MIN, B_RATIO, RHO_KER
from mad_icp_amd import _build, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES = 14
ORACLE_FRAMES = 9


@pytest.fixture(scope="module")
def drive():
    scene = synth.Scene(0)
    return [synth.render_scan(scene, synth.path_pose(1.0 * i), 100 + i) for i in range(N_FRAMES)]


@pytest.fixture(scope="module")
def pypeline(natives):
    from mad_icp.src.pybind import pypeline as m

    return m


def pose_err(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    return np.linalg.norm(d[:3, 3]), np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1))


@pytest.mark.parametrize("front_end", ["host", "default"])
def test_pipeline_fullsize_matches_oracle_and_reports_rate(pypeline, drive, capsys, front_end):
    """`host`: the host tree builder (the reference's trees bit for bit; MAD_ICP_GPU_BUILD=0 / setDeviceFrontEnd(False));
    `default`: what an unmodified caller gets since round 5 — tree construction on the device for deskew = false."""
    threads = min(os.cpu_count() or 1, 16)
    args = (10.0, False, B_MAX, RHO_KER, 0.8, B_MIN, B_RATIO, 16, threads, False)

    def make():
        p = pypeline.Pipeline(*args)
        assert p.deviceFrontEnd()
        if front_end == "host":
            p.setDeviceFrontEnd(False)
        return p

    gp = make()
    op = O.Pipeline(*args)
    t_plain = []
    for i, s in enumerate(drive):
        v = pypeline.VectorEigen3d(s)
        t = time.perf_counter()
        gp.compute(0.1 * i, v)
        t_plain.append(time.perf_counter() - t)
        if i < ORACLE_FRAMES:
            op.compute(0.1 * i, s)
            dt, da = pose_err(op.currentPose(), np.asarray(gp.currentPose()))
            assert dt <= 1e-5 and da <= 1e-5, (i, dt, da)
            assert gp.currentID() == op.currentID() and gp.keyframeID() == op.keyframeID()
            assert gp.isMapUpdated() == op.isMapUpdated()
            if i > 0:
                assert abs(gp.lastInliersRatio() - op.lastInliersRatio()) < 2e-3
    # the same drive with the look-ahead: the tree of scan i+1 is built while frame i is registered — bit-identical poses
    ga = make()
    t_ahead = []
    clouds = [pypeline.VectorEigen3d(s) for s in drive]
    ga.prefetch(clouds[0])
    for i in range(N_FRAMES):
        t = time.perf_counter()
        if i + 1 < N_FRAMES:
            ga.prefetch(clouds[i + 1])  # BEFORE compute(i): the build of scan i + 1 runs during the whole frame step of scan i
        ga.compute(0.1 * i, clouds[i])
        t_ahead.append(time.perf_counter() - t)
    assert np.array_equal(np.asarray(ga.trajectory()), np.asarray(gp.trajectory()))
    # two scans ahead: the builds of scans i + 1 and i + 2 share the builder's threads (one's serial top levels run beside the
    # other's parallel bottom)
    # (host builder only: the device front-end has ONE construction in flight, on the library's build stream)
    t_ahead2 = list(t_ahead)
    if front_end == "host":
        g2 = make()
        t_ahead2 = []
        g2.prefetch(clouds[0])
        g2.prefetch(clouds[1])
        for i in range(N_FRAMES):
            t = time.perf_counter()
            if i + 2 < N_FRAMES:
                g2.prefetch(clouds[i + 2])
            g2.compute(0.1 * i, clouds[i])
            t_ahead2.append(time.perf_counter() - t)
        assert np.array_equal(np.asarray(g2.trajectory()), np.asarray(gp.trajectory()))
    assert np.asarray(gp.currentLeaves()).shape == op.currentLeaves().shape if ORACLE_FRAMES == N_FRAMES else True
    gt = np.linalg.inv(synth.path_pose(0.0)) @ synth.path_pose(1.0 * (N_FRAMES - 1))
    assert np.linalg.norm(np.asarray(gp.currentPose())[:3, 3] - gt[:3, 3]) < 0.1
    with capsys.disabled():
        # (means: with a look-ahead the per-frame series is bimodal — a frame that waits a build's length, then quick ones)
        print("\n[pipeline, %s front-end @ %d pts/scan, %d host threads] compute: mean %.2f ms/frame = %.0f frames/s (build %.2f ms, "
              "registration %.3f ms); with prefetch(i + 1) before compute(i): %.2f ms = %.0f frames/s; two scans ahead: %.2f ms = %.0f frames/s"
              % (front_end, drive[0].shape[0], threads, 1e3 * np.mean(t_plain[2:]), 1.0 / np.mean(t_plain[2:]), gp.lastBuildMs(),
                 gp.lastIcpMs(), 1e3 * np.mean(t_ahead[2:-1]), 1.0 / np.mean(t_ahead[2:-1]), 1e3 * np.mean(t_ahead2[2:-2]),
                 1.0 / np.mean(t_ahead2[2:-2])))


def test_two_pipelines_share_the_context(pypeline, drive):
    """Two Pipeline objects in one process use the same device context (csrc/host/device.h): driven alternately they
    produce exactly the trajectories they produce alone."""
    args = (10.0, False, B_MAX, RHO_KER, 0.8, B_MIN, B_RATIO, 4, 4, False)
    solo = pypeline.Pipeline(*args)
    for i, s in enumerate(drive[:6]):
        solo.compute(0.1 * i, s)
    a, b = pypeline.Pipeline(*args), pypeline.Pipeline(*args)
    for i, s in enumerate(drive[:6]):
        a.compute(0.1 * i, s)
        b.compute(0.1 * i, s)
    assert np.array_equal(np.asarray(a.trajectory()), np.asarray(solo.trajectory()))
    assert np.array_equal(np.asarray(b.trajectory()), np.asarray(solo.trajectory()))


def test_cpp_runner_surface(natives, drive, tmp_path):
    """bin_runner's calls (apps/cpp_runners/bin_runner.cpp:106-121,174,180) compiled against csrc/host/pipeline.h."""
    pkg = os.path.join(ROOT, "mad_icp_amd")
    exe = str(tmp_path / "runner_surface")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(pkg, "csrc", "host"), os.path.join(ROOT, "tests", "cpp", "runner_surface.cpp"),
                           "-o", exe, "-L" + pkg, "-lmadicp_host", "-lmadicp_hip", "-Wl,-rpath," + pkg, "-pthread"])
    data = tmp_path / "velodyne"
    data.mkdir()
    for i, s in enumerate(drive[:5]):
        rec = np.zeros((s.shape[0], 4), np.float32)
        rec[:, :3] = s.astype(np.float32)
        rec.tofile(str(data / ("%06d.bin" % i)))
    est = str(tmp_path / "estimate.txt")
    out = subprocess.run([exe, str(data), est], check=True, capture_output=True, text=True).stdout
    assert "Loading frame # 0" in out and "Loading frame # 4" in out
    poses = np.loadtxt(est).reshape(-1, 3, 4)
    assert poses.shape[0] == 5 and np.allclose(poses[0], np.eye(4)[:3])
    # the same float32-rounded clouds through the Python surface give the same trajectory
    from mad_icp.src.pybind import pypeline as m

    gp = m.Pipeline(10.0, False, 0.2, 0.1, 0.8, 0.1, 0.02, 4, 4, False)
    for i, s in enumerate(drive[:5]):
        c = s.astype(np.float32)
        r = np.sqrt((c * c).sum(axis=1, dtype=np.float32))
        c = c[(r >= 0.7) & (r <= 120.0)].astype(np.float64)
        gp.compute(0.1 * i, c)
        assert np.allclose(np.asarray(gp.currentPose())[:3], poses[i], atol=1e-9)
    gt = np.linalg.inv(synth.path_pose(0.0)) @ synth.path_pose(4.0)
    assert np.linalg.norm(poses[-1][:, 3] - gt[:3, 3]) < 0.05


@pytest.mark.gpu
def test_deskewed_drive_with_the_azimuth_order_computed_ahead(pypeline, drive, capsys):
    """deskew = True on the host path: the tree needs the two previous poses, the azimuth order of the scan does not —
    prefetch(i + 1) before compute(i) computes it beside the frame step (csrc/host/deskew.h).  Scans with distinct azimuths
    (the synthetic ones, which share 64 points per azimuth column, with 1e-7 m of jitter: what a real sensor's noise does)
    take the parallel order, tied ones the reference's serial route; either way the trajectory is the one without look-ahead
    bit for bit."""
    rng = np.random.default_rng(4)
    threads = min(os.cpu_count() or 1, 16)
    args = (10.0, True, B_MAX, RHO_KER, 0.8, B_MIN, B_RATIO, 16, threads, False)
    for tag, scans in (("distinct azimuths", [s + rng.normal(scale=1e-7, size=s.shape) for s in drive]), ("tied azimuths", drive)):
        clouds = [pypeline.VectorEigen3d(s) for s in scans]
        plain, ahead = pypeline.Pipeline(*args), pypeline.Pipeline(*args)
        plain.setDeviceFrontEnd(False)  # (the HOST path's look-ahead; the default is the device front-end since round 6)
        ahead.setDeviceFrontEnd(False)
        t_plain, t_ahead = [], []
        for i in range(N_FRAMES):
            t = time.perf_counter()
            plain.compute(0.1 * i, clouds[i])
            t_plain.append(time.perf_counter() - t)
        for i in range(N_FRAMES):
            t = time.perf_counter()
            if i + 1 < N_FRAMES:
                ahead.prefetch(clouds[i + 1])
            ahead.compute(0.1 * i, clouds[i])
            t_ahead.append(time.perf_counter() - t)
        assert np.array_equal(np.asarray(ahead.trajectory()), np.asarray(plain.trajectory()))
        assert ahead.lookAheadHits() >= N_FRAMES - 3
        with capsys.disabled():
            print("\n[pipeline, host path, deskew on, %s] mean %.2f ms per frame; with prefetch(i + 1) before compute(i): %.2f ms"
                  % (tag, 1e3 * np.mean(t_plain[3:]), 1e3 * np.mean(t_ahead[3:-1])))
