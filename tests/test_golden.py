"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle): the oracle must
keep reproducing them on CPU, and the HIP path must reproduce them on the GPU."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O
from fixtures import B_MAX, B_MIN, B_RATIO, PARAMS, RHO_KER, four_walls, street_problem

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("K", [1, 3])
def test_oracle_reproduces_street_golden(K):
    g = np.load(os.path.join(GOLD, f"street_k{K}.npz"))
    pb = street_problem(K)
    assert digest(pb["query_scans"][0]) == str(g["scan_sha"]), "synthetic generator drifted"
    trees = []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        t = O.Tree(s, B_MAX, B_MIN, 2)
        t.transform(T[:3, :3], T[:3, 3])
        trees.append(t)
    q = O.Tree(pb["query_scans"][0], B_MAX, B_MIN, 2)
    assert q.num_leaves == int(g["n_leaves"])
    assert [t.num_leaves for t in trees] == list(g["tree_leaves"])
    T0 = pb["query_guess"][0]
    for k, t in enumerate(trees):
        _, _, corr, rej, _, d = O.icp_linearize(q, t, T0, B_MAX, RHO_KER, B_RATIO)
        assert digest(corr) == str(g["corr_sha"][k]) and digest(rej) == str(g["rej_sha"][k]) and d == int(g["depth"][k])
    r = O.icp_register(q, trees, T0, 15, B_MAX, RHO_KER, B_RATIO, num_threads=1)
    assert np.array_equal(r["X_iters"], g["X_iters"]) and np.array_equal(r["T"], g["T"])
    assert int(r["matched"].sum()) == int(g["n_matched"])


def test_oracle_reproduces_walls_golden():
    g = np.load(os.path.join(GOLD, "walls_nn.npz"))
    np.random.seed(42)
    cloud = four_walls(2000)
    assert digest(cloud) == str(g["cloud_sha"])
    t = O.Tree(cloud, 1e-5, 0.1, 2)
    rng = np.random.default_rng(9)
    q = cloud[rng.integers(0, len(cloud), 5000)] + rng.normal(0, 0.05, (5000, 3))
    leaf, depth, dist = t.search(q, want_dist=True)
    assert digest(leaf) == str(g["leaf_sha"]) and int(depth.sum()) == int(g["depth_sum"])
    assert float(dist.sum()) == float(g["dist_sum"])


@pytest.mark.gpu
@pytest.mark.parametrize("K", [1, 3])
def test_hip_reproduces_street_golden(ctx, K):
    from mad_icp_amd import capi

    g = np.load(os.path.join(GOLD, f"street_k{K}.npz"))
    pb = street_problem(K)
    tids = []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        ht = capi.HostTree(s, B_MAX, B_MIN, 2)
        ht.transform(T[:3, :3], T[:3, 3])
        tids.append(ctx.upload(ht))
    qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
    mid = ctx.moving_upload(qh.leaf_means())
    L = qh.num_leaves
    assert L == int(g["n_leaves"])
    T0 = pb["query_guess"][0]
    lin = ctx.icp_linearize(mid, tids, T0, PARAMS, L)
    for k in range(K):
        assert digest(lin["corr"][k] & 0x7FFFFFFF) == str(g["corr_sha"][k])
        assert digest((lin["corr"][k] >> 31).astype(np.uint8)) == str(g["rej_sha"][k])
    assert lin["visits"] == int(g["depth"].sum())
    r = ctx.icp_register(mid, tids, T0, PARAMS, 15, L)
    d = np.linalg.inv(g["T"]) @ r["T"]
    assert np.linalg.norm(d[:3, 3]) <= 1e-5
    assert np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1)) <= 1e-5
    assert abs(int(r["matched"].sum()) - int(g["n_matched"])) <= 2
    for t in tids:
        ctx.tree_release(t)
    ctx.moving_release(mid)


@pytest.mark.gpu
def test_hip_reproduces_walls_golden(ctx):
    from mad_icp_amd import capi

    g = np.load(os.path.join(GOLD, "walls_nn.npz"))
    np.random.seed(42)
    cloud = four_walls(2000)
    ht = capi.HostTree(cloud, 1e-5, 0.1, 2)
    tid = ctx.upload(ht)
    rng = np.random.default_rng(9)
    q = cloud[rng.integers(0, len(cloud), 5000)] + rng.normal(0, 0.05, (5000, 3))
    r = ctx.nn_search(tid, q)
    assert digest(r["leaf"]) == str(g["leaf_sha"]) and int(r["depth"].sum()) == int(g["depth_sum"])
    assert float(r["dist"].sum()) == float(g["dist_sum"])
    ctx.tree_release(tid)
