"""CPU tests of the product's host tree builder: bit-identical to the oracle's restatement of
mad_tree.cpp:47-130 on the same clouds, including the degenerate ones."""
import numpy as np
import pytest

import oracle_lib as O
from fixtures import four_walls, street_problem
from mad_icp_amd import capi


def assert_same_tree(points, b_max, b_min, par):
    ht = capi.HostTree(points, b_max, b_min, par)
    ot = O.Tree(points, b_max, b_min, par)
    ex = ot.export()
    nodes = ht.nodes
    assert ht.num_nodes == ot.num_nodes and ht.num_leaves == ot.num_leaves
    leaf = ex["left"] < 0
    assert np.array_equal(nodes["right"] == 0, leaf)
    assert np.array_equal(nodes["mean"], ex["mean"])
    want_dir = np.where(leaf[:, None], ex["evecs"][:, :, 0], ex["evecs"][:, :, 2])
    assert np.array_equal(nodes["dir"], want_dir, equal_nan=True)
    assert np.array_equal(nodes["bbox0"], ex["bbox"][:, 0])
    idx = np.arange(len(nodes))
    assert np.array_equal((idx + nodes["right"])[~leaf], ex["right"][~leaf])
    assert np.array_equal((idx + 1)[~leaf], ex["left"][~leaf])
    # leaf ordinals follow getLeafs() (DFS, left first)
    assert np.array_equal(nodes["leaf_id"][leaf], np.arange(leaf.sum()))
    assert np.array_equal(ht.leaf_nodes, idx[leaf])
    m, n, b0 = ot.leaves()
    assert np.array_equal(ht.leaf_means(), m)
    assert np.array_equal(nodes["dir"][leaf], n, equal_nan=True)
    return ht, ot


@pytest.mark.parametrize("b_max,par", [(0.2, 0), (0.2, 3), (1e-5, 2), (1.0, 1)])
def test_street_scan(b_max, par):
    pb = street_problem(2)
    assert_same_tree(pb["keyframe_scans"][0], b_max, 0.1, par)


def test_four_walls_one_leaf_per_point():
    np.random.seed(42)
    cloud = four_walls(2000)
    ht, _ = assert_same_tree(cloud, 1e-5, 0.1, 2)
    assert ht.num_leaves == cloud.shape[0]


@pytest.mark.parametrize("case", ["one", "two", "three_collinear", "duplicates", "planar", "line", "huge_coords"])
def test_degenerate_clouds(case):
    rng = np.random.default_rng(11)
    pts = {
        "one": np.array([[0.5, -1.0, 2.0]]),
        "two": np.array([[0.0, 0, 0], [3.0, 1, 0]]),
        "three_collinear": np.array([[0.0, 0, 0], [1.0, 1, 1], [2.0, 2, 2]]),
        "duplicates": np.repeat(rng.normal(size=(7, 3)), 9, axis=0),
        "planar": np.column_stack([rng.uniform(-5, 5, 800), rng.uniform(-5, 5, 800), np.zeros(800)]),
        "line": np.column_stack([np.linspace(0, 50, 500), np.zeros(500), np.zeros(500)]),
        "huge_coords": rng.normal(size=(1000, 3)) + 1e6,
    }[case]
    assert_same_tree(pts, 0.2, 0.1, 0)
    assert_same_tree(pts, 1e-5, 0.1, 2)


def test_transform_matches_oracle():
    pb = street_problem(2)
    T = pb["keyframe_poses"][1]
    ht, ot = assert_same_tree(pb["keyframe_scans"][1], 0.2, 0.1, 2)
    ht.transform(T[:3, :3], T[:3, 3])
    ot.transform(T[:3, :3], T[:3, 3])
    ex = ot.export()
    leaf = ex["left"] < 0
    assert np.array_equal(ht.nodes["mean"], ex["mean"])
    assert np.array_equal(ht.nodes["dir"], np.where(leaf[:, None], ex["evecs"][:, :, 0], ex["evecs"][:, :, 2]))


def test_input_cloud_is_not_modified():
    rng = np.random.default_rng(2)
    pts = rng.normal(size=(3000, 3))
    keep = pts.copy()
    capi.HostTree(pts, 0.2, 0.1, 2)
    assert np.array_equal(pts, keep)


def test_empty_cloud_raises():
    with pytest.raises(ValueError):
        capi.HostTree(np.zeros((0, 3)), 0.2, 0.1, 0)


def test_concurrent_builds_share_the_task_pool():
    """Several host threads building different trees at once (the ctypes call releases the GIL; all of them fork onto
    the one persistent task pool, whose waiters help each other): every tree must equal the one built alone on the
    calling thread."""
    import threading

    from mad_icp_amd import synth

    scene = synth.Scene(9)
    clouds = [synth.render_scan(scene, synth.path_pose(0.7 * i), 300 + i, n_beams=16, n_azimuth=400) for i in range(6)]
    want = [capi.HostTree(c, 0.2, 0.1, 0).nodes.copy() for c in clouds]
    got = [None] * len(clouds)

    def work(i):
        for _ in range(3):
            got[i] = capi.HostTree(clouds[i], 0.2, 0.1, 4).nodes.copy()

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(clouds))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for a, b in zip(want, got):
        assert a.tobytes() == b.tobytes()


def _partition(points, mean, normal, impl):
    import ctypes as C

    pts = np.ascontiguousarray(points, dtype=np.float64).copy()
    dp = C.POINTER(C.c_double)
    m = np.ascontiguousarray(mean, dtype=np.float64)
    nv = np.ascontiguousarray(normal, dtype=np.float64)
    mid = capi.host_lib().madicp_host_debug_partition(pts.ctypes.data_as(dp), pts.shape[0], m.ctypes.data_as(dp),
                                                      nv.ctypes.data_as(dp), impl)
    return mid, pts


def test_partition_closed_form_equals_reference_loop_exhaustively(natives):
    """utils.h:37-52 `split`: the builder derives the permutation from per-point side flags in one descending sweep
    (tree_builder.cpp partition_from_flags) instead of running the reference's data-dependent swap loop.  Every
    left/right pattern of up to 12 points, tagged points so that the full permutation is compared, not just the sides."""
    mean, normal = np.zeros(3), np.array([1.0, 0.0, 0.0])
    for n in range(0, 13):
        for pattern in range(1 << n):
            x = np.array([-1.0 if (pattern >> i) & 1 else 1.0 for i in range(n)])
            pts = np.column_stack([x, np.arange(n, dtype=np.float64), np.arange(n, dtype=np.float64) * 7 + 1]) if n else np.zeros((0, 3))
            m0, p0 = _partition(pts, mean, normal, 0)
            m1, p1 = _partition(pts, mean, normal, 1)
            assert m0 == m1 == int((x < 0).sum()), (n, pattern)
            assert np.array_equal(p0, p1), (n, pattern)


def test_partition_closed_form_equals_reference_loop_on_large_random_inputs(natives):
    rng = np.random.default_rng(5)
    for n, p_left in [(1000, 0.5), (4097, 0.03), (4097, 0.97), (120000, 0.5), (50001, 0.0), (50001, 1.0)]:
        pts = rng.normal(size=(n, 3))
        pts[:, 0] = np.where(rng.random(n) < p_left, -np.abs(pts[:, 0]) - 1e-3, np.abs(pts[:, 0]))
        # on-plane and NaN points go right, like `dot < 0` says
        pts[rng.integers(0, n, 5), 0] = 0.0
        pts[rng.integers(0, n, 3), 1] = np.nan
        mean, normal = np.zeros(3), np.array([1.0, 0.0, 0.0])
        m0, p0 = _partition(pts, mean, normal, 0)
        m1, p1 = _partition(pts, mean, normal, 1)
        assert m0 == m1
        assert np.array_equal(p0, p1, equal_nan=True)


@pytest.mark.parametrize("par", [0, 3])
def test_builder_reports_the_spread_the_device_bound_needs(par):
    """rho2 = max |mean_i - mean_0|_2 over the internal nodes (what madicp_tree_upload computes while validating and
    madicp_tree_upload_trusted takes from the builder)."""
    pb = street_problem(2)
    ht = capi.HostTree(pb["keyframe_scans"][0], 0.2, 0.1, par)
    nodes = ht.nodes
    internal = nodes["right"] != 0
    want = np.sqrt(((nodes["mean"][internal] - nodes["mean"][0]) ** 2).sum(axis=1)).max()
    assert abs(ht.rho2 - want) <= 1e-12 * want
    one = capi.HostTree(np.array([[1.0, 2.0, 3.0]]), 0.2, 0.1, 0)
    assert one.rho2 == 0.0


def test_thread_budget_changes_nothing_but_time(natives):
    """madicp_host_set_threads (Pipeline's num_threads; the reference's omp_set_num_threads, pipeline.cpp:64-65): workers over
    the budget park, workers under it pick up tasks again when it is raised — and the tree is the same bits for every
    budget and every forking depth (the result is schedule-independent, like the reference's std::async recursion)."""
    pb = street_problem(2)
    scan = pb["keyframe_scans"][0]
    ref = capi.HostTree(scan, 0.2, 0.1, 0).nodes.tobytes()
    try:
        for threads in (1, 2, 5, 16, 64, 3):
            capi.host_lib().madicp_host_set_threads(threads)
            for par in (1, 4, 6):
                assert capi.HostTree(scan, 0.2, 0.1, par).nodes.tobytes() == ref, (threads, par)
    finally:
        capi.host_lib().madicp_host_set_threads(1 << 20)


def test_random_small_clouds():
    """Sixty random small clouds — blobs, sheets, lines, duplicates at random scales and thresholds, the inputs of
    tests/test_reference_structure_pin.py's random leg — through the product's builder and the oracle's: the corners of the
    leaf rule, the plane-predecessor rule and the small-leaf normal (mad_tree.cpp:64-93), bit for bit."""
    r2 = np.random.default_rng(77)
    for i in range(60):
        n = int(r2.integers(1, 400))
        kind = int(r2.integers(4))
        c = r2.normal(size=(n, 3)) * r2.choice([0.01, 0.3, 5.0], size=3)
        if kind == 1:
            c[:, 2] = 0.0
        elif kind == 2:
            c[:, 1:] = 0.0
        elif kind == 3:
            c = np.repeat(c[: max(1, n // 4)], 4, axis=0)
        c = c + r2.normal(size=3) * 10.0
        b_max, b_min = float(r2.choice([1e-5, 0.05, 0.2, 1.0])), float(r2.choice([0.01, 0.1, 0.5]))
        assert_same_tree(c, b_max, b_min, int(r2.integers(3)))
