"""Shared seeded inputs for the tests (small enough for the oracle to finish in seconds)."""
import functools

import numpy as np

from mad_icp_amd import synth

B_MAX, B_MIN, RHO_KER, B_RATIO = 0.2, 0.1, 0.1, 0.02  # mad_icp/configurations/default.cfg:2-7
PARAMS = (B_MAX, RHO_KER, B_RATIO)


def four_walls(points_per_wall, wall_height=2.0, wall_width=4.0):
    """The reference's tool fixture (apps/utils/tools/tools_utils.py:3-21): 4 walls + floor, legacy
    np.random.uniform stream drawn x, y, z per plane, planes in the order wall1..wall4, floor.
    Caller seeds np.random (np.random.seed(42) in nn_search.py:36 / mad_registration.py:51)."""
    w, h = wall_width, wall_height
    spans = [((0, w), (0, 0), (0, h)), ((0, w), (w, w), (0, h)), ((0, 0), (0, w), (0, h)), ((w, w), (0, w), (0, h)),
             ((0, w), (0, w), (0, 0))]
    planes = []
    for span in spans:
        cols = [np.random.uniform(lo, hi, points_per_wall) for lo, hi in span]
        planes.append(np.column_stack(cols))
    return np.vstack(planes)


@functools.lru_cache(maxsize=None)
def street_problem(n_keyframes, seed=3, n_beams=32, n_azimuth=600, n_queries=1):
    """Reduced-resolution street problem (19k rays per scan) for oracle-speed parity tests."""
    return synth.make_problem(n_keyframes, seed=seed, n_beams=n_beams, n_azimuth=n_azimuth, n_queries=n_queries)


@functools.lru_cache(maxsize=8)
def _scene(seed):
    return synth.Scene(seed)


@functools.lru_cache(maxsize=320)
def full_scan(scene_seed, s, noise_seed):
    """One full-size (64 x 1875 rays) scan at arc parameter `s` of scene `scene_seed` — rendering costs 0.16 s and the long
    drives of several test files walk the same frames.  Callers must not modify the returned array."""
    a = synth.render_scan(_scene(scene_seed), synth.path_pose(s), noise_seed)
    a.setflags(write=False)
    return a
