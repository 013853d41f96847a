"""CPU tests of the oracle itself: the reference's two known-answer properties (the only pins the reference
offers — it has no tests), and independent numpy checks of the restated Eigen routines."""
import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st
from scipy.spatial.transform import Rotation

import oracle_lib as O
from fixtures import four_walls


def test_nn_self_query_total_error_is_zero():
    """apps/utils/tools/nn_search.py:36-61 + tools/README.md:9-10: with b_max=1e-5 (one leaf per point) every
    point of the cloud, queried against the tree built from the same cloud, finds itself."""
    np.random.seed(42)
    cloud = four_walls(10000)
    tree = O.Tree(cloud, 1e-5, 0.1, 2)
    assert tree.num_leaves == cloud.shape[0] and tree.num_nodes == 2 * cloud.shape[0] - 1
    leaf, depth, dist = tree.search(cloud, want_dist=True)
    assert dist.sum() == 0.0
    assert len(np.unique(leaf)) == cloud.shape[0]


def test_pairwise_registration_converges_to_identity():
    """apps/utils/tools/mad_registration.py:51-68 ("gt T = identity")."""
    np.random.seed(42)
    ref = four_walls(1000)
    T = np.eye(4)
    T[:3, :3] = Rotation.from_euler("xyz", [0.1, 0.1, 0.1]).as_matrix()
    T[:3, 3] = np.random.rand(3)
    rt = O.Tree(ref, 0.2, 0.1, 0)
    qt = O.Tree(ref.copy(), 0.2, 0.1, 0)
    r = O.icp_register(qt, [rt], T, 15, 0.2, 0.1, 0.02)
    assert np.abs(r["T"] - np.eye(4)).max() < 1e-6
    assert r["matched"].all()


def test_build_is_schedule_independent():
    """std::async recursion depth must not change the tree."""
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(5000, 3)) * [8, 8, 0.3]
    a = O.Tree(pts, 0.2, 0.1, 0).export()
    b = O.Tree(pts, 0.2, 0.1, 3).export()
    for k in a:
        assert np.array_equal(a[k], b[k], equal_nan=True), k


sym3 = st.lists(st.floats(-50, 50, allow_nan=False), min_size=6, max_size=6)


@settings(max_examples=200, deadline=None)
@given(sym3)
def test_eig3_matches_numpy(v):
    a, b, c, d, e, f = v
    A = np.array([[a, b, c], [b, d, e], [c, e, f]])
    w, V = O.eig3(A)
    w_np = np.linalg.eigvalsh(A)
    scale = max(1.0, np.abs(A).max())
    # the closed-form solver is accurate to ~1e-8 relative for nearly-degenerate spectra, ~1e-13 otherwise
    assert np.allclose(w, w_np, atol=2e-7 * scale)
    assert np.all(np.diff(w) >= -1e-9 * scale)
    gaps = np.diff(w_np)
    if gaps.min() > 1e-3 * scale:
        assert np.allclose(w, w_np, atol=1e-11 * scale)
        assert np.allclose(V.T @ V, np.eye(3), atol=1e-8)
        assert np.allclose(A @ V, V * w, atol=1e-8 * scale)


@settings(max_examples=100, deadline=None)
@given(st.integers(0, 2**31 - 1))
def test_ldlt6_matches_numpy_on_spd(seed):
    rng = np.random.default_rng(seed)
    J = rng.normal(size=(40, 6)) * rng.uniform(0.1, 10, size=6)
    H = J.T @ J
    b = rng.normal(size=6)
    x = O.ldlt6_solve(H, b)
    assert np.linalg.norm(H @ x - b) <= 1e-9 * max(1.0, np.linalg.norm(b)) * np.linalg.cond(H)
    assert np.allclose(x, np.linalg.solve(H, b), rtol=1e-7, atol=1e-9)


def test_ldlt6_zero_matrix_gives_zero():
    assert np.array_equal(O.ldlt6_solve(np.zeros((6, 6)), np.ones(6)), np.zeros(6))


def test_det_inverse6():
    rng = np.random.default_rng(3)
    J = rng.normal(size=(30, 6))
    H = J.T @ J
    assert np.isclose(O.det_inverse6(H), 1.0 / np.linalg.det(H), rtol=1e-9)


@settings(max_examples=100, deadline=None)
@given(st.lists(st.floats(-3, 3, allow_nan=False), min_size=3, max_size=3))
def test_expmap_logmap(w):
    w = np.array(w)
    R = O.expmap_so3(w)
    th = np.linalg.norm(w)
    if th * th >= 1e-8:
        assert np.allclose(R, Rotation.from_rotvec(w).as_matrix(), atol=1e-12)
        if th < 3.0:
            assert np.allclose(O.logmap_so3(R), w, atol=1e-6)
    else:
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        assert np.array_equal(R, np.eye(3) + K)  # first-order branch, lie_algebra.h:46-47


def test_single_point_and_tiny_clouds():
    """Edge cases of the build: 1 point -> NaN covariance -> leaf; 2 points far apart."""
    t1 = O.Tree(np.array([[1.0, 2.0, 3.0]]), 0.2, 0.1, 0)
    assert t1.num_nodes == 1 and t1.num_leaves == 1
    m, n, b0 = t1.leaves()
    assert np.array_equal(m[0], [1.0, 2.0, 3.0]) and np.isnan(n).all() and b0[0] == 0.0
    t2 = O.Tree(np.array([[0.0, 0, 0], [5.0, 0, 0]]), 0.2, 0.1, 0)
    assert t2.num_nodes == 3 and t2.num_leaves == 2
    leaf, depth = t2.search(np.array([[0.1, 0, 0], [4.0, 1, 1]]))
    assert list(depth) == [1, 1] and leaf[0] != leaf[1]
