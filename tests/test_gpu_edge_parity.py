"""GPU parity on inputs that stress the exactness arguments rather than the common case: scenes far from the
origin (the screening certificate's error bound grows with |q - o|), clouds with no planar structure, duplicated
points (zero-extent leaves, rank-deficient covariances), and the largest keyframe count the ABI accepts.
Same bar as tests/test_gpu_parity.py: correspondences and gate decisions bit-exact, pose within 1e-5 m / 1e-5 rad."""
import numpy as np
import pytest

import oracle_lib as O
from fixtures import B_MAX, B_MIN, B_RATIO, PARAMS, RHO_KER, street_problem
from mad_icp_amd import capi, synth

pytestmark = pytest.mark.gpu


def pose_err(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    ang = np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1))
    return np.linalg.norm(d[:3, 3]), ang


def _check_registration(ctx, fixed_clouds, fixed_poses, moving_cloud, T0, tol=(1e-5, 1e-5), b_max=B_MAX):
    hts, ots, tids = [], [], []
    for s, T in zip(fixed_clouds, fixed_poses):
        ht = capi.HostTree(s, b_max, B_MIN, 2)
        ot = O.Tree(s, b_max, B_MIN, 2)
        if T is not None:
            ht.transform(T[:3, :3], T[:3, 3])
            ot.transform(T[:3, :3], T[:3, 3])
        assert np.array_equal(ht.nodes["mean"], ot.export()["mean"])  # identical trees, or the bar below is meaningless
        hts.append(ht)
        ots.append(ot)
        tids.append(ctx.tree_upload(ht.nodes, ht.num_leaves))
    qh = capi.HostTree(moving_cloud, b_max, B_MIN, 2)
    qo = O.Tree(moving_cloud, b_max, B_MIN, 2)
    mid = ctx.moving_upload(qh.leaf_means())
    L = qh.num_leaves
    params = (b_max, RHO_KER, B_RATIO)
    # correspondences + gate at the initial pose, bit for bit, and the node-visit count
    g = ctx.icp_linearize(mid, tids, T0, params, L)
    visits = 0
    for k, ot in enumerate(ots):
        _, _, corr, rej, _, depth = O.icp_linearize(qo, ot, T0, b_max, RHO_KER, B_RATIO)
        assert np.array_equal(g["corr"][k] & 0x7FFFFFFF, corr), k
        assert np.array_equal((g["corr"][k] >> 31).astype(np.uint8), rej), k
        visits += depth
    assert g["visits"] == visits
    # the whole registration
    o = O.icp_register(qo, ots, T0, 15, b_max, RHO_KER, B_RATIO, num_threads=4)
    r = ctx.icp_register(mid, tids, T0, params, 15, L)
    dt, da = pose_err(o["T"], r["T"])
    assert dt <= tol[0] and da <= tol[1], (dt, da)
    assert np.array_equal(r["matched"], o["matched"])
    assert r["visits"] == o["depth_sum"]
    for t in tids:
        ctx.tree_release(t)
    ctx.moving_release(mid)
    return r, o


@pytest.mark.parametrize("offset", [1.0e3, 1.0e5])
def test_scene_far_from_the_origin(ctx, offset):
    """The same street, translated by kilometres (UTM-like map frames).  |q - o| stays small because the screening
    origin is the tree's own centroid, but every coordinate now carries 1e3..1e5 times less absolute precision, and the
    exact fallback must still decide like the reference."""
    pb = street_problem(2)
    shift = np.eye(4)
    shift[:3, 3] = [offset, -0.7 * offset, 0.01 * offset]
    poses = [shift @ T for T in pb["keyframe_poses"]]
    T0 = shift @ pb["query_guess"][0]
    r, o = _check_registration(ctx, pb["keyframe_scans"], poses, pb["query_scans"][0], T0)
    gt = shift @ pb["query_gt"][0]
    assert np.linalg.norm(r["T"][:3, 3] - gt[:3, 3]) < 0.1


def test_unstructured_cloud(ctx):
    """A Gaussian blob has no planes: leaves are small balls with arbitrary normals, most pairs fail the gate, H is
    poorly conditioned.  Nothing about the GPU path assumes structure."""
    rng = np.random.default_rng(11)
    ref = rng.normal(0.0, 3.0, (6000, 3))
    mov = ref[rng.permutation(6000)[:4000]] + rng.normal(0.0, 0.003, (4000, 3))
    T0 = synth.perturbation(5, trans=0.05, rot_deg=0.3)
    _check_registration(ctx, [ref], [None], mov, T0)


def test_duplicated_points_and_tiny_leaves(ctx):
    """Every point four times (zero-extent leaves, singular covariances), b_max small enough that most leaves hold a
    single distinct location."""
    rng = np.random.default_rng(12)
    base = np.concatenate([rng.uniform(-2, 2, (1500, 3)) * [1, 1, 0.0], rng.uniform(-2, 2, (1500, 3)) * [1, 0.0, 1]])
    ref = np.repeat(base, 4, axis=0)
    mov = base + rng.normal(0.0, 0.001, base.shape)
    T0 = synth.perturbation(6, trans=0.02, rot_deg=0.2)
    _check_registration(ctx, [ref], [None], mov, T0, b_max=0.05)


def test_maximum_number_of_keyframes(ctx):
    """MADICP_MAX_TREES keyframes in one registration (the device Job holds the descriptors by value)."""
    pb = street_problem(4, n_beams=16, n_azimuth=300)
    K = capi.MAX_TREES
    clouds = [pb["keyframe_scans"][k % 4][(k // 4)::3] for k in range(K)]  # 128 different sub-samplings
    poses = [pb["keyframe_poses"][k % 4] for k in range(K)]
    hts, ots, tids = [], [], []
    for s, T in zip(clouds, poses):
        ht = capi.HostTree(s, B_MAX, B_MIN, 0)
        ot = O.Tree(s, B_MAX, B_MIN, 0)
        ht.transform(T[:3, :3], T[:3, 3])
        ot.transform(T[:3, :3], T[:3, 3])
        hts.append(ht)
        ots.append(ot)
        tids.append(ctx.tree_upload(ht.nodes, ht.num_leaves))
    qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 0)
    qo = O.Tree(pb["query_scans"][0], B_MAX, B_MIN, 0)
    mid = ctx.moving_upload(qh.leaf_means())
    T0 = pb["query_guess"][0]
    o = O.icp_register(qo, ots, T0, 15, B_MAX, RHO_KER, B_RATIO, num_threads=8)
    r = ctx.icp_register(mid, tids, T0, PARAMS, 15, qh.num_leaves)
    dt, da = pose_err(o["T"], r["T"])
    assert dt <= 1e-5 and da <= 1e-5, (dt, da)
    assert np.array_equal(r["matched"], o["matched"]) and r["visits"] == o["depth_sum"]
    for t in tids:
        ctx.tree_release(t)
    ctx.moving_release(mid)


@pytest.mark.gpu
def test_random_small_clouds_nearest_neighbour_and_linearisation(ctx):
    """The sixty random small clouds of the structure pin and the host-builder test (blobs, sheets, lines, duplicates at random
    scales and thresholds, 1 .. 400 points) on the device: nearest neighbours and distances against the oracle's, and one
    linearisation of each cloud's own leaves against its tree — correspondences and gate bit for bit.  Trees of one, two,
    three leaves, leaves without a normal of their own, NaN-free or not: whatever the reference builds, the kernels walk."""
    r2 = np.random.default_rng(77)
    checked = 0
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
        par = int(r2.integers(3))
        ht = capi.HostTree(c, b_max, b_min, par)
        ot = O.Tree(c, b_max, b_min, par)
        tid = ctx.tree_upload(ht.nodes, ht.num_leaves)
        q = c + r2.normal(size=c.shape) * 0.01
        got = ctx.nn_search(tid, q, want=("leaf", "dist"))
        leaf, _, dist = ot.search(q, want_dist=True)
        assert np.array_equal(got["leaf"], leaf), i
        assert np.array_equal(got["dist"], dist, equal_nan=True), i
        mid = ctx.moving_upload(ht.leaf_means())
        T0 = np.eye(4)
        T0[:3, 3] = r2.normal(size=3) * 0.02
        g = ctx.icp_linearize(mid, [tid], T0, (b_max, RHO_KER, B_RATIO), ht.num_leaves)
        _, _, corr, rej, mat, depth = O.icp_linearize(ot, ot, T0, b_max, RHO_KER, B_RATIO)
        assert np.array_equal(g["corr"][0] & 0x7FFFFFFF, corr), i
        assert np.array_equal((g["corr"][0] >> 31).astype(np.uint8), rej), i
        assert g["visits"] == depth, i
        ctx.moving_release(mid)
        ctx.tree_release(tid)
        checked += 1
    assert checked == 60


@pytest.mark.gpu
def test_gate_within_ulps_of_its_threshold(ctx):
    """The gate of mad_icp.cpp:81-83 is `(ml - f.mean).norm() > min_ball + b_ratio * |p|` with a correctly rounded square
    root.  The kernel decides it from the SQUARES whenever they are more than 2^-50 apart (relative) and evaluates the root
    only in between (icp_linearize_body.inc.h).  Here the pairs sit ON that threshold: a fixed cloud in the plane x ~ 0 (x = a
    few ulps of 1.5), moving leaves straight above its leaf representatives at x = 1.5 + c ulps, b_ratio = 0 and min_ball =
    1.5 — so the distance is 1.5 + (c - a) ulps EXACTLY, within 2^-49 of the ball for every pair, on both sides of it and on
    it.  Gate decisions and correspondences against the oracle's, bit for bit; both outcomes must occur."""
    rng = np.random.default_rng(12)
    u = 2.0 ** -52
    n = 3000
    fixed = np.stack([rng.integers(0, 7, n) * u, rng.uniform(0, 1, n), rng.uniform(0, 1, n)], axis=1)
    ht = capi.HostTree(fixed, 0.05, 0.01, 2)
    ot = O.Tree(fixed, 0.05, 0.01, 2)
    assert np.array_equal(ht.nodes["mean"], ot.export()["mean"])
    reps = ht.leaf_means()
    moving = reps.copy()
    moving[:, 0] = 1.5 + rng.integers(-6, 13, reps.shape[0]) * u
    qh = capi.HostTree(moving, 1e-5, 0.01, 2)
    qo = O.Tree(moving, 1e-5, 0.01, 2)
    assert qh.num_leaves == moving.shape[0]  # (every moving point its own leaf)
    tid = ctx.tree_upload(ht.nodes, ht.num_leaves)
    mid = ctx.moving_upload(qh.leaf_means())
    params = (1.5, RHO_KER, 0.0)
    g = ctx.icp_linearize(mid, [tid], np.eye(4), params, qh.num_leaves)
    _, _, corr, rej, mat, depth = O.icp_linearize(qo, ot, np.eye(4), 1.5, RHO_KER, 0.0)
    assert np.array_equal(g["corr"][0] & 0x7FFFFFFF, corr)
    assert np.array_equal((g["corr"][0] >> 31).astype(np.uint8), rej)
    assert np.array_equal(g["matched"], mat) and g["visits"] == depth
    # the pairs really are where the squares cannot decide: distance / ball - 1 within a few ulps, all three cases present
    ml = qh.leaf_means()
    d = np.sqrt(((ml - reps[corr]) ** 2).sum(axis=1))
    near = np.abs(d / 1.5 - 1.0) <= 2.0 ** -49
    assert near.mean() > 0.5
    assert (d[near] > 1.5).any() and (d[near] < 1.5).any() and (d[near] == 1.5).any()
    assert 0.2 < rej[near].mean() < 0.9
    ctx.moving_release(mid)
    ctx.tree_release(tid)
