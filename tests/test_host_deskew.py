"""The host deskew (csrc/host/deskew.h) against the oracle's Pipeline::deskew (pipeline.cpp:79-123), row for row, bit for bit:
with distinct azimuths through the PARALLEL azimuth order (any correct sort gives std::sort's permutation then), with tied
azimuths through the reference's own serial route, which is then the only one that can be right."""
import time

import numpy as np
import pytest

import oracle_lib as O
from mad_icp_amd import capi, synth


def _pose(tx, ty, yaw, pitch):
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    T = np.eye(4)
    T[:3, :3] = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]]) @ np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    T[:3, 3] = [tx, ty, 0.02]
    return T


MOTIONS = [(0.9, 0.05, 0.02, 0.03), (0.0, 0.0, 0.0, 0.0), (-1.4, 0.3, -0.2, 0.01)]


@pytest.mark.parametrize("motion", MOTIONS)
@pytest.mark.parametrize("n", [1, 2, 37, 5000, 70001])
def test_distinct_azimuths_take_the_parallel_order(motion, n):
    rng = np.random.default_rng(n)
    pts = rng.normal(size=(n, 3)) * [30.0, 30.0, 2.0]
    assert np.unique(np.arctan2(pts[:, 1], pts[:, 0])).size == n
    Tp, Tn = _pose(0.1, 0.0, 0.01, 0.0), _pose(*motion)
    ref, vel = O.deskew(pts, Tp, Tn, 10.0)
    out, v, fast = capi.host_deskew(pts, Tp, Tn, 10.0)
    assert fast
    assert np.array_equal(v, vel)
    assert np.array_equal(out, ref)
    out1, _, fast1 = capi.host_deskew(pts, Tp, Tn, 10.0, route=1)  # the serial route gives the same rows
    assert not fast1 and np.array_equal(out1, ref)


@pytest.mark.parametrize("kind", ["scan", "duplicates", "on_axis", "signed_zero"])
def test_tied_azimuths_take_the_reference_route(kind):
    rng = np.random.default_rng(3)
    if kind == "scan":  # a synthetic scan: 32 beams share every azimuth column
        pts = synth.render_scan(synth.Scene(0), synth.path_pose(0.0), 5, n_beams=32, n_azimuth=400)
    elif kind == "duplicates":
        pts = np.repeat(rng.normal(size=(500, 3)) * 20, 3, axis=0)
        rng.shuffle(pts)
    elif kind == "on_axis":  # y == 0: azimuth exactly 0 or pi for many points
        pts = rng.normal(size=(3000, 3)) * 10
        pts[::3, 1] = 0.0
    else:  # atan2(-0.0, x) = -0.0 and atan2(0.0, x) = +0.0 compare equal
        pts = rng.normal(size=(1000, 3)) * 10
        pts[:200, 1] = np.where(rng.integers(2, size=200) == 1, 0.0, -0.0)
        pts[:200, 0] = np.abs(pts[:200, 0]) + 1.0
    Tp, Tn = np.eye(4), _pose(0.8, -0.1, 0.03, 0.01)
    ref, vel = O.deskew(pts, Tp, Tn, 10.0)
    out, v, fast = capi.host_deskew(pts, Tp, Tn, 10.0)
    assert not fast
    assert np.array_equal(out, ref) and np.array_equal(v, vel)


def test_empty_and_timing(capsys):
    out, _, _ = capi.host_deskew(np.zeros((0, 3)), np.eye(4), np.eye(4), 10.0)
    assert out.shape == (0, 3)
    rng = np.random.default_rng(9)
    pts = rng.normal(size=(120000, 3)) * [30.0, 30.0, 2.0]
    Tp, Tn = np.eye(4), _pose(0.8, -0.1, 0.03, 0.01)
    best = {}
    for route in (0, 1):
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            capi.host_deskew(pts, Tp, Tn, 10.0, route=route)
            ts.append(time.perf_counter() - t)
        best[route] = min(ts)
    with capsys.disabled():
        print("\n[host deskew, 120 k points] parallel azimuth order %.2f ms, reference route (serial sort) %.2f ms"
              % (1e3 * best[0], 1e3 * best[1]))
