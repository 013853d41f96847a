"""BASELINE.json's full sizes on the GPU, checked through size-independent properties (the oracle would take minutes
here): self-query round trip, registration of a scan onto its own tree from a perturbed pose, additivity of (H, b)
over keyframe trees, transform round trip; plus the capacity / degenerate-input edges of the C ABI."""
import numpy as np
import pytest

from fixtures import B_MAX, B_MIN, PARAMS
from mad_icp_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    return synth.make_problem(4, seed=11)  # 4 x ~120k-point scans + one query


def test_self_query_at_120k_points(ctx, big):
    """config 1/2 shape: every one of 120k points, queried against the one-leaf-per-point tree of its own scan,
    returns itself (apps/utils/tools/README.md:9-10) — at the full scan size."""
    scan = big["keyframe_scans"][0]
    ht = capi.HostTree(scan, 1e-5, B_MIN, 3)
    assert ht.num_leaves == len(np.unique(scan, axis=0))
    tid = ctx.upload(ht)
    g = ctx.nn_search(tid, scan)
    assert g["dist"].sum() == 0.0
    assert np.array_equal(ht.nodes["mean"][g["node"]], scan)
    assert g["depth"].max() < 64
    ctx.tree_release(tid)


def test_registration_onto_own_tree_recovers_identity(ctx, big):
    """120k-point scan vs the MAD-tree of the same scan, started 0.3 m / 1 deg off: the estimate must return to the
    identity (known answer, no oracle needed) and every leaf must end up matched."""
    scan = big["keyframe_scans"][1]
    ht = capi.HostTree(scan, B_MAX, B_MIN, 3)
    tid = ctx.upload(ht)
    mid = ctx.moving_upload(ht.leaf_means())
    g = ctx.icp_register(mid, [tid], synth.perturbation(5), PARAMS, 15, ht.num_leaves)
    assert np.abs(g["T"] - np.eye(4)).max() < 1e-7
    assert g["matched"].all()
    # idempotence: starting from the answer stays at the answer
    g2 = ctx.icp_register(mid, [tid], g["T"], PARAMS, 15, ht.num_leaves)
    assert np.abs(g2["T"] - np.eye(4)).max() < 1e-7
    ctx.tree_release(tid)
    ctx.moving_release(mid)


def test_H_b_are_additive_over_keyframes(ctx, big):
    """The join over keyframe trees is a plain sum (mad_icp.cpp:106-109): linearising against K trees at once must
    equal the sum of K single-tree linearisations — a checksum of checksums at full size."""
    tids = []
    for s, T in zip(big["keyframe_scans"], big["keyframe_poses"]):
        ht = capi.HostTree(s, B_MAX, B_MIN, 3)
        ht.transform(T[:3, :3], T[:3, 3])
        tids.append(ctx.upload(ht))
    qh = capi.HostTree(big["query_scans"][0], B_MAX, B_MIN, 3)
    mid = ctx.moving_upload(qh.leaf_means())
    L = qh.num_leaves
    T0 = big["query_guess"][0]
    all_ = ctx.icp_linearize(mid, tids, T0, PARAMS, L, want_corr=True)
    H = np.zeros((6, 6))
    b = np.zeros(6)
    visits = 0
    matched = np.zeros(L, np.uint8)
    for k, t in enumerate(tids):
        one = ctx.icp_linearize(mid, [t], T0, PARAMS, L, want_corr=True)
        assert np.array_equal(one["corr"][0], all_["corr"][k])  # same correspondences alone or together
        H += one["H"]
        b += one["b"]
        visits += one["visits"]
        matched |= one["matched"]
    assert np.allclose(all_["H"], H, rtol=1e-11, atol=1e-11 * np.abs(H).max())
    assert np.allclose(all_["b"], b, rtol=1e-11, atol=1e-11 * np.abs(b).max())
    assert all_["visits"] == visits and np.array_equal(all_["matched"], matched)
    assert np.allclose(all_["H"], all_["H"].T)
    for t in tids:
        ctx.tree_release(t)
    ctx.moving_release(mid)


def test_transform_round_trip_preserves_searches(ctx, big):
    """applyTransform by T then by T^-1: the tree must answer queries like the untouched tree (to rounding)."""
    scan = big["keyframe_scans"][2]
    T = big["keyframe_poses"][2]
    ht = capi.HostTree(scan, B_MAX, B_MIN, 3)
    a = ctx.upload(ht)
    b = ctx.upload(ht)
    Ti = np.linalg.inv(T)
    ctx.tree_transform(b, T[:3, :3], T[:3, 3])
    ctx.tree_transform(b, Ti[:3, :3], Ti[:3, 3])
    q = big["query_scans"][0][::7]
    ga = ctx.nn_search(a, q)
    gb = ctx.nn_search(b, q)
    assert (ga["leaf"] != gb["leaf"]).mean() < 1e-4  # only queries within ~1e-13 m of a split plane may flip
    same = ga["leaf"] == gb["leaf"]
    assert np.allclose(ga["dist"][same], gb["dist"][same], atol=1e-9)
    ctx.tree_release(a)
    ctx.tree_release(b)


def test_single_leaf_tree_and_single_moving_leaf(ctx):
    one = capi.HostTree(np.array([[1.0, 2.0, 3.0]]), B_MAX, B_MIN, 0)
    assert one.num_nodes == 1
    tid = ctx.upload(one)
    g = ctx.nn_search(tid, np.array([[0.0, 0, 0], [9.0, 9, 9]]))
    assert list(g["leaf"]) == [0, 0] and list(g["depth"]) == [0, 0]
    assert np.allclose(g["dist"], [np.sqrt(14.0), np.sqrt(64 + 49 + 36)])
    mid = ctx.moving_upload(np.array([[1.0, 2.0, 3.05]]))
    lin = ctx.icp_linearize(mid, [tid], np.eye(4), PARAMS, 1)
    assert lin["visits"] == 0 and lin["matched"][0] == 1  # within the gate; the leaf's normal is NaN -> H is NaN
    assert np.isnan(lin["H"]).any()
    ctx.tree_release(tid)
    ctx.moving_release(mid)


def test_capacity_limits_are_errors(ctx):
    ht = capi.HostTree(np.random.default_rng(0).normal(size=(200, 3)), B_MAX, B_MIN, 0)
    tid = ctx.upload(ht)
    mid = ctx.moving_upload(ht.leaf_means())
    with pytest.raises(capi.MadIcpError):
        ctx.icp_register(mid, [tid] * (capi.MAX_TREES + 1), np.eye(4), PARAMS, 15, ht.num_leaves)
    with pytest.raises(capi.MadIcpError):
        ctx.icp_register_batch([mid] * (capi.MAX_BATCH + 1), [tid], np.tile(capi.pose12(np.eye(4)), (capi.MAX_BATCH + 1, 1)),
                               PARAMS, 15)
    with pytest.raises(capi.MadIcpError):
        ctx.icp_register(mid, [], np.eye(4), PARAMS, 15, ht.num_leaves)
    # the maximum itself is fine: 128 references to the same tree
    g = ctx.icp_register(mid, [tid] * capi.MAX_TREES, np.eye(4), PARAMS, 3, ht.num_leaves)
    assert np.isfinite(g["X"]).all()
    with pytest.raises(capi.MadIcpError):
        ctx.tree_upload(ht.nodes[:-1], ht.num_leaves)  # n_nodes != 2 n_leaves - 1
    ctx.tree_release(tid)
    ctx.moving_release(mid)


def test_batch_of_64_scans(ctx, big):
    """MADICP_MAX_BATCH scans in flight against 2 trees: every scan converges like it does alone."""
    tids = []
    for s, T in zip(big["keyframe_scans"][:2], big["keyframe_poses"][:2]):
        ht = capi.HostTree(s[::4], B_MAX, B_MIN, 2)
        ht.transform(T[:3, :3], T[:3, 3])
        tids.append(ctx.upload(ht))
    qh = capi.HostTree(big["keyframe_scans"][1][::4], B_MAX, B_MIN, 2)
    mids = [ctx.moving_upload(qh.leaf_means()) for _ in range(capi.MAX_BATCH)]
    T0 = big["keyframe_poses"][1] @ synth.perturbation(3, 0.1, 0.3)
    X0 = np.tile(capi.pose12(T0), (capi.MAX_BATCH, 1))
    r = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
    one = ctx.icp_register(mids[0], tids, T0, PARAMS, 15, qh.num_leaves)
    for s in range(capi.MAX_BATCH):
        assert np.allclose(r["X"][s], one["X"], atol=1e-9)
    d = np.linalg.inv(big["keyframe_poses"][1]) @ capi.pose44(r["X"][0])
    assert np.linalg.norm(d[:3, 3]) < 0.05
    for t in tids:
        ctx.tree_release(t)
    for m in mids:
        ctx.moving_release(m)
