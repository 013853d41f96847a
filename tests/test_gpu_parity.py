"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bar (BASELINE.json north_star): correspondence indices bit-exact; pose within 1e-5 m / 1e-5 rad."""
import numpy as np
import pytest

import oracle_lib as O
from fixtures import B_MAX, B_MIN, B_RATIO, PARAMS, RHO_KER, four_walls, street_problem
from mad_icp_amd import capi

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-5
POSE_TOL_RAD = 1e-5


def pose_err(Ta, Tb):
    d = np.linalg.inv(Ta) @ Tb
    ang = np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1))
    return np.linalg.norm(d[:3, 3]), ang


def build_pair(points, b_max=B_MAX, T=None):
    """Same cloud -> product host tree (uploaded) and oracle tree, optionally moved to the map frame."""
    ht = capi.HostTree(points, b_max, B_MIN, 2)
    ot = O.Tree(points, b_max, B_MIN, 2)
    if T is not None:
        ht.transform(T[:3, :3], T[:3, 3])
        ot.transform(T[:3, :3], T[:3, 3])
    return ht, ot


def test_device_sqrt_and_norm_bit_exact(ctx):
    """fp64 sqrt on gfx950 must round like the CPU's: the gate compares sqrt() results."""
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(4096, 3)) * rng.uniform(1e-3, 1e3, size=(4096, 1))
    ht, ot = build_pair(pts, b_max=1e-5)
    tid = ctx.tree_upload(ht.nodes, ht.num_leaves)
    q = rng.normal(size=(20000, 3)) * 5
    g = ctx.nn_search(tid, q)
    leaf, depth, dist = ot.search(q, want_dist=True)
    assert np.array_equal(g["leaf"], leaf)
    assert np.array_equal(g["depth"], depth)
    assert np.array_equal(g["dist"], dist)  # bit-exact, includes a device sqrt
    ctx.tree_release(tid)


def test_nn_self_query_known_answer(ctx):
    """apps/utils/tools/nn_search.py: every point queried against the one-leaf-per-point tree of the same
    cloud returns itself -> total matching error exactly 0 (apps/utils/tools/README.md:9-10)."""
    np.random.seed(42)
    cloud = four_walls(10000)
    ht = capi.HostTree(cloud, 1e-5, 0.1, 2)
    tid = ctx.tree_upload(ht.nodes, ht.num_leaves)
    g = ctx.nn_search(tid, cloud)
    assert g["dist"].sum() == 0.0
    nodes = ht.nodes
    assert np.array_equal(nodes["mean"][g["node"]], cloud)
    ctx.tree_release(tid)


@pytest.mark.parametrize("b_max", [0.2, 1e-5])
def test_nn_search_matches_oracle(ctx, b_max):
    pb = street_problem(2)
    ht, ot = build_pair(pb["keyframe_scans"][0], b_max=b_max, T=pb["keyframe_poses"][0])
    tid = ctx.tree_upload(ht.nodes, ht.num_leaves)
    q = (pb["query_scans"][0] @ pb["query_gt"][0][:3, :3].T) + pb["query_gt"][0][:3, 3]
    g = ctx.nn_search(tid, q)
    leaf, depth, dist = ot.search(q, want_dist=True)
    assert np.array_equal(g["leaf"], leaf)
    assert np.array_equal(g["depth"], depth)
    assert np.array_equal(g["dist"], dist)
    ctx.tree_release(tid)


def test_nn_search_lds_top_equals_plain_kernel(ctx):
    """Option nn_lds_top = 1: batches of >= 16 k queries walk the tree's top levels from LDS (nn_descend_top) instead of
    the plain kernel (the default: the staged variant measured slower for a single launch).  Same descent: leaf, node,
    depth and distance identical, and equal to the oracle's."""
    pb = street_problem(2)
    s, T = pb["keyframe_scans"][0], pb["keyframe_poses"][0]
    ht = capi.HostTree(s, B_MAX, B_MIN, 2)
    ot = O.Tree(s, B_MAX, B_MIN, 2)
    tid = ctx.tree_upload(ht.nodes, ht.num_leaves)
    rng = np.random.default_rng(11)
    q = np.concatenate([pb["query_scans"][0], s[::3] + rng.normal(scale=0.05, size=s[::3].shape)])
    assert q.shape[0] >= 16384
    ctx.set_option("nn_lds_top", 1)
    a = ctx.nn_search(tid, q)
    ctx.set_option("nn_lds_top", 0)
    b = ctx.nn_search(tid, q)
    for k in ("leaf", "node", "depth", "dist"):
        assert np.array_equal(a[k], b[k]), k
    leaf, depth, dist = ot.search(q, want_dist=True)
    assert np.array_equal(a["leaf"], leaf) and np.array_equal(a["depth"], depth) and np.array_equal(a["dist"], dist)
    ctx.tree_release(tid)


def test_screening_fallback_is_exact(ctx):
    """Queries placed ON split planes (and a hair off them) defeat the 16-byte screening test, so the lanes
    must take the exact fp64 path — and still agree with the oracle bit for bit."""
    pb = street_problem(2)
    T = pb["keyframe_poses"][1]
    ht, ot = build_pair(pb["keyframe_scans"][1], T=T)
    tid = ctx.tree_upload(ht.nodes, ht.num_leaves)
    nodes = ht.nodes
    internal = nodes[nodes["right"] != 0]
    rng = np.random.default_rng(4)
    pick = internal[rng.integers(0, len(internal), 6000)]
    tang = np.cross(pick["dir"], rng.normal(size=(len(pick), 3)))
    q = np.concatenate([
        pick["mean"],                                             # s == 0 at that node
        pick["mean"] + 1e-9 * pick["dir"],
        pick["mean"] - 1e-9 * pick["dir"],
        pick["mean"] + tang * rng.uniform(0, 0.2, (len(pick), 1)),  # slides inside the plane: |s| ~ 1e-17
        pick["mean"] + tang + 3e-6 * pick["dir"] * rng.normal(size=(len(pick), 1)),
    ])
    g = ctx.nn_search(tid, q)
    leaf, depth, dist = ot.search(q, want_dist=True)
    assert np.array_equal(g["leaf"], leaf)
    assert np.array_equal(g["depth"], depth)
    assert np.array_equal(g["dist"], dist)
    ctx.tree_release(tid)


def test_nan_and_far_queries(ctx):
    """NaN / huge queries: every comparison on NaN is false -> the descent keeps going right, as on the CPU."""
    pb = street_problem(2)
    ht, ot = build_pair(pb["keyframe_scans"][0])
    tid = ctx.tree_upload(ht.nodes, ht.num_leaves)
    q = np.array([[np.nan, 0, 0], [0, np.nan, 1], [1e300, -1e300, 1e300], [np.inf, 0, 0], [1e9, 1e9, -1e9], [0, 0, 0]])
    g = ctx.nn_search(tid, q)
    leaf, depth = ot.search(q)
    assert np.array_equal(g["leaf"], leaf) and np.array_equal(g["depth"], depth)
    ctx.tree_release(tid)


def test_tree_transform_matches_host_and_oracle(ctx):
    pb = street_problem(2)
    T = pb["keyframe_poses"][1]
    ht = capi.HostTree(pb["keyframe_scans"][1], B_MAX, B_MIN, 2)
    tid = ctx.tree_upload(ht.nodes, ht.num_leaves)
    ctx.tree_transform(tid, T[:3, :3], T[:3, 3])
    dev = ctx.tree_download(tid, ht.num_nodes)
    ht.transform(T[:3, :3], T[:3, 3])
    host = ht.nodes
    for f in ("mean", "dir", "right", "leaf_id", "bbox0"):
        assert np.array_equal(dev[f], host[f], equal_nan=True), f
    ot = O.Tree(pb["keyframe_scans"][1], B_MAX, B_MIN, 2)
    ot.transform(T[:3, :3], T[:3, 3])
    assert np.array_equal(dev["mean"], ot.export()["mean"])
    ctx.tree_release(tid)


def _setup_registration(ctx, K, n_queries=1):
    pb = street_problem(K, n_queries=n_queries)
    hts, ots, tids = [], [], []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        ht, ot = build_pair(s, T=T)
        hts.append(ht)
        ots.append(ot)
        tids.append(ctx.tree_upload(ht.nodes, ht.num_leaves))
    qh, qo, mids = [], [], []
    for s in pb["query_scans"]:
        h = capi.HostTree(s, B_MAX, B_MIN, 2)
        qh.append(h)
        qo.append(O.Tree(s, B_MAX, B_MIN, 2))
        mids.append(ctx.moving_upload(h.leaf_means()))
    return pb, hts, ots, tids, qh, qo, mids


def _teardown(ctx, tids, mids):
    for t in tids:
        ctx.tree_release(t)
    for m in mids:
        ctx.moving_release(m)


@pytest.mark.parametrize("K", [1, 3])
def test_linearize_correspondences_bit_exact(ctx, K):
    """Per (leaf, tree): NN leaf ordinal and gate decision identical to MADicp::update; H, b agree to
    summation-order rounding."""
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, K)
    T = pb["query_guess"][0]
    L = qh[0].num_leaves
    g = ctx.icp_linearize(mids[0], tids, T, PARAMS, L)
    H = np.zeros((6, 6))
    b = np.zeros(6)
    matched = np.zeros(L, np.uint8)
    visits = 0
    for k in range(K):
        Hk, bk, corr, rej, mat, depth = O.icp_linearize(qo[0], ots[k], T, B_MAX, RHO_KER, B_RATIO)
        assert np.array_equal(g["corr"][k] & 0x7FFFFFFF, corr), f"tree {k}: NN leaf ordinals differ"
        assert np.array_equal((g["corr"][k] >> 31).astype(np.uint8), rej), f"tree {k}: gate decisions differ"
        H += Hk
        b += bk
        matched |= mat
        visits += depth
    assert np.array_equal(g["matched"], matched)
    assert g["visits"] == visits
    scale = np.abs(H).max()
    assert np.allclose(g["H"], (np.tril(H) + np.tril(H, -1).T), rtol=0, atol=1e-10 * scale)
    assert np.allclose(g["b"], b, rtol=0, atol=1e-10 * max(1.0, np.abs(b).max()))
    _teardown(ctx, tids, mids)


def test_trees_dealt_over_the_xcd_pieces_keep_the_callers_indices(ctx):
    """A Job lists the caller's trees dealt over the kernel's eight XCD pieces (fill_job; option deal_trees: 2 = rows of eight in
    alternating direction, the default since round 6; 1 = round-robin; 0 = as listed).  Nothing the caller sees may depend on it:
    the correspondence trace is indexed by the CALLER's tree index — checked against the oracle tree by tree, with a tree count
    that is not a multiple of eight — the matched flags and the visit count are identical whatever the option, H and b equal to
    summation-order rounding, and a registration ends at the same pose to 1e-12."""
    K = 11
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, K)
    T = pb["query_guess"][0]
    L = qh[0].num_leaves
    assert ctx.get_option("deal_trees") == 2
    res = {}
    for deal in (2, 1, 0):
        ctx.set_option("deal_trees", deal)
        res[deal] = (ctx.icp_linearize(mids[0], tids, T, PARAMS, L), ctx.icp_register(mids[0], tids, T, PARAMS, 6, L))
    ctx.set_option("deal_trees", 2)
    g, _ = res[2]
    for k in range(K):
        _, _, corr, rej, _, _ = O.icp_linearize(qo[0], ots[k], T, B_MAX, RHO_KER, B_RATIO)
        assert np.array_equal(g["corr"][k] & 0x7FFFFFFF, corr), f"tree {k}: the trace is not in the caller's order"
        assert np.array_equal((g["corr"][k] >> 31).astype(np.uint8), rej)
    for deal in (2, 1):
        assert np.array_equal(res[deal][0]["corr"], res[0][0]["corr"]), deal
        assert np.array_equal(res[deal][0]["matched"], res[0][0]["matched"]), deal
        assert res[deal][0]["visits"] == res[0][0]["visits"], deal
        scale = np.abs(res[0][0]["H"]).max()
        assert np.allclose(res[deal][0]["H"], res[0][0]["H"], rtol=0, atol=1e-12 * scale), deal
        assert np.allclose(res[deal][0]["b"], res[0][0]["b"], rtol=0, atol=1e-12 * max(1.0, np.abs(res[0][0]["b"]).max())), deal
        assert np.abs(res[deal][1]["X"] - res[0][1]["X"]).max() < 1e-12, deal
        assert np.array_equal(res[deal][1]["matched"], res[0][1]["matched"]), deal
    _teardown(ctx, tids, mids)


@pytest.mark.parametrize("K,n_queries", [(1, 1), (3, 1), (16, 1), (32, 8)])
def test_ranges_dealt_in_groups_of_64_leaves_same_decisions(ctx, K, n_queries):
    """Option interleave_ranges (kernels.hip.h, "Ranges"; default 2 since round 6): the ranges a scan is cut into are every RPT-th
    group of 64 leaves instead of contiguous stretches, so that every workgroup draws a sample of the whole scan and not one
    stretch of space.  Which workgroup adds which pair changes — H and b agree with the contiguous launch to summation-order
    rounding (1e-12), the pose to 1e-12 — and nothing else does: the correspondence trace (leaf indices and gate decisions, every
    tree), the matched flags, their count and the visit counter are the same bit for bit; in a one-scan launch, in a batch that
    shares the chip (DEEP launches: tree-major and leaf-major rounds), with option 1 (batches only) too.  The poses are the
    oracle's (mad_icp.cpp:74-117 under pipeline.cpp:166-193)."""
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, K, n_queries=n_queries)
    Ls = [h.num_leaves for h in qh]
    X0 = np.stack([capi.pose12(T) for T in pb["query_guess"]])
    assert ctx.get_option("interleave_ranges") == 2
    res = {}
    try:
        for mode in (2, 1, 0):
            ctx.set_option("interleave_ranges", mode)
            r = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
            r["matched"] = [ctx.icp_fetch_matched(i, L) for i, L in enumerate(Ls)]
            r["lin"] = ctx.icp_linearize(mids[0], tids, pb["query_guess"][0], PARAMS, Ls[0])
            res[mode] = r
    finally:
        ctx.set_option("interleave_ranges", 2)
    ref = res[0]
    for mode in (2, 1):
        o = res[mode]
        assert np.array_equal(o["lin"]["corr"], ref["lin"]["corr"]), mode
        assert np.array_equal(o["lin"]["matched"], ref["lin"]["matched"]) and o["lin"]["visits"] == ref["lin"]["visits"], mode
        scale = np.abs(ref["lin"]["H"]).max()
        assert np.allclose(o["lin"]["H"], ref["lin"]["H"], rtol=0, atol=1e-12 * scale), mode
        assert np.array_equal(o["n_matched"], ref["n_matched"]) and np.array_equal(o["visits"], ref["visits"]), mode
        for a, b in zip(o["matched"], ref["matched"]):
            assert np.array_equal(a, b), mode
        assert np.abs(o["X"] - ref["X"]).max() <= 1e-12, (mode, np.abs(o["X"] - ref["X"]).max())
        assert np.abs(o["H"] - ref["H"]).max() <= 1e-12 * np.abs(ref["H"]).max(), mode
    if n_queries == 1:  # (a one-scan launch is not a batch: option 1 leaves it contiguous — the same bits as option 0)
        assert np.array_equal(res[1]["X"], ref["X"]) and np.array_equal(res[1]["H"], ref["H"])
    q = n_queries - 1
    o = O.icp_register(qo[q], ots, pb["query_guess"][q], 15, B_MAX, RHO_KER, B_RATIO, num_threads=4)
    terr, rerr = pose_err(o["T"], capi.pose44(res[2]["X"][q]))
    assert terr <= POSE_TOL_M and rerr <= POSE_TOL_RAD
    assert np.array_equal(res[2]["matched"][q], o["matched"]) and res[2]["visits"][q] == o["depth_sum"]
    _teardown(ctx, tids, mids)


def test_deep_launch_with_ranges_of_less_than_a_pass(ctx):
    """From 24 keyframes on a launch is DEEP — one range of the scan and all the trees of its XCD piece per workgroup — as soon as a
    range holds 512 leaves (option deep_min_leaves; pick_geometry), i.e. less than one pass of a workgroup: three scans in flight
    here, ten workgroups per XCD piece, ranges of ~640 leaves.  Same decisions as the unit-per-workgroup launch of the same
    registration (deep_min_leaves out of reach), poses and H to summation-order rounding, and the oracle's poses."""
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, 24, n_queries=3)
    Ls = [h.num_leaves for h in qh]
    assert 512 * 10 <= min(Ls) < 1536 * 10  # (DEEP under the round-6 rule, not under the two-pass rule)
    X0 = np.stack([capi.pose12(T) for T in pb["query_guess"]])
    assert ctx.get_option("deep_min_leaves") == 512
    res = {}
    try:
        for name, v in (("deep", 512), ("units", 1 << 20)):
            ctx.set_option("deep_min_leaves", v)
            r = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
            r["matched"] = [ctx.icp_fetch_matched(i, L) for i, L in enumerate(Ls)]
            res[name] = r
    finally:
        ctx.set_option("deep_min_leaves", 512)
    a, b = res["deep"], res["units"]
    assert np.array_equal(a["n_matched"], b["n_matched"]) and np.array_equal(a["visits"], b["visits"])
    for x, y in zip(a["matched"], b["matched"]):
        assert np.array_equal(x, y)
    assert np.abs(a["X"] - b["X"]).max() <= 1e-12 and np.abs(a["H"] - b["H"]).max() <= 1e-12 * np.abs(b["H"]).max()
    for q in (0, 2):
        o = O.icp_register(qo[q], ots, pb["query_guess"][q], 15, B_MAX, RHO_KER, B_RATIO, num_threads=4)
        terr, rerr = pose_err(o["T"], capi.pose44(a["X"][q]))
        assert terr <= POSE_TOL_M and rerr <= POSE_TOL_RAD
        assert np.array_equal(a["matched"][q], o["matched"]) and a["visits"][q] == o["depth_sum"]
    _teardown(ctx, tids, mids)


@pytest.mark.parametrize("K", [1, 5])
def test_two_leaves_per_lane_same_decisions(ctx, K):
    """Option queries_per_lane = 2 (two leaves per lane and pass share their loads; the default is one): another instantiation of
    the round kernel over the same ranges — dealt or contiguous — with the same correspondence trace, matched flags and visit
    counter, H / b / pose to summation-order rounding."""
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, K)
    T = pb["query_guess"][0]
    L = qh[0].num_leaves
    res = {}
    try:
        for qpt in (1, 2):
            for inter in (2, 0):
                ctx.set_option("queries_per_lane", qpt)
                ctx.set_option("interleave_ranges", inter)
                res[qpt, inter] = (ctx.icp_linearize(mids[0], tids, T, PARAMS, L), ctx.icp_register(mids[0], tids, T, PARAMS, 15, L))
    finally:
        ctx.set_option("queries_per_lane", 1)
        ctx.set_option("interleave_ranges", 2)
    ref = res[1, 2]
    for key, (lin, reg) in res.items():
        assert np.array_equal(lin["corr"], ref[0]["corr"]), key
        assert np.array_equal(lin["matched"], ref[0]["matched"]) and lin["visits"] == ref[0]["visits"], key
        assert np.allclose(lin["H"], ref[0]["H"], rtol=0, atol=1e-12 * np.abs(ref[0]["H"]).max()), key
        assert np.array_equal(reg["matched"], ref[1]["matched"]) and reg["visits"] == ref[1]["visits"], key
        assert np.abs(reg["X"] - ref[1]["X"]).max() <= 1e-12, key
    _teardown(ctx, tids, mids)


@pytest.mark.parametrize("option,value,default", [("units_per_workgroup", 3, 1), ("grid_blocks_per_cu", 2, 1), ("publish_side", 0, 1),
                                                  ("lds_stage_min_leaves", 1 << 30, 1024)])
def test_launch_shape_options_change_no_decision(ctx, option, value, default):
    """The launch-shape knobs of include/madicp_hip.h that no other test turns: more (tree, range) units per workgroup, more workgroups
    per CU, results written by the closing kernel itself instead of the side stream, no LDS-staged top.  Synchronous, batched and
    streamed registrations of the same scans end with the same flags and visit counters, poses to summation-order rounding."""
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, 5, n_queries=3)
    Ls = [h.num_leaves for h in qh]
    X0 = np.stack([capi.pose12(T) for T in pb["query_guess"]])
    leaves = [h.leaf_means() for h in qh]
    assert ctx.get_option(option) == default
    res = {}
    try:
        for v in (default, value):
            ctx.set_option(option, v)
            one = ctx.icp_register(mids[0], tids, pb["query_guess"][0], PARAMS, 15, Ls[0])
            bat = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
            bat["matched"] = [ctx.icp_fetch_matched(i, L) for i, L in enumerate(Ls)]
            tk = [ctx.stream_submit(leaves[q], tids, pb["query_guess"][q], PARAMS, 15) for q in range(2)]
            st = [ctx.stream_collect(t, Ls[q]) for q, t in enumerate(tk)]
            res[v] = (one, bat, st)
    finally:
        ctx.set_option(option, default)
    (a1, ab, as_), (b1, bb, bs) = res[default], res[value]
    assert np.array_equal(a1["matched"], b1["matched"]) and a1["visits"] == b1["visits"] and np.abs(a1["X"] - b1["X"]).max() <= 1e-12
    assert np.array_equal(ab["n_matched"], bb["n_matched"]) and np.array_equal(ab["visits"], bb["visits"])
    assert np.abs(ab["X"] - bb["X"]).max() <= 1e-12
    for x, y in zip(ab["matched"], bb["matched"]):
        assert np.array_equal(x, y)
    for x, y in zip(as_, bs):
        assert np.array_equal(x["matched"], y["matched"]) and x["visits"] == y["visits"] and np.abs(x["X"] - y["X"]).max() <= 1e-12
    # (and the streamed registration is the synchronous one)
    assert np.array_equal(as_[0]["matched"], a1["matched"]) and np.abs(as_[0]["X"] - a1["X"]).max() <= 1e-12
    _teardown(ctx, tids, mids)


def test_work_distribution_over_random_geometries(ctx):
    """Round 6's work distribution — ranges dealt in groups of 64 leaves, trees in alternating rows, DEEP launches from 512 leaves
    per range, leaf-major rounds with their two queues — against the plain layout (contiguous ranges, trees as listed, one unit per
    workgroup, every round tree-major) over forty seeded draws of the launch geometry: 1-40 keyframes picked at random from a map,
    1-8 scans in flight with ragged sizes from a few hundred to 8 700 leaves, 2 / 5 / 15 rounds.  Every draw: the same matched
    flags, matched counts and visit counters, poses to summation-order rounding."""
    rng = np.random.default_rng(123)
    pb = street_problem(40, n_queries=8)
    all_tids = []
    for s_, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        ht, _ = build_pair(s_, T=T)
        all_tids.append(ctx.tree_upload(ht.nodes, ht.num_leaves))
    qh = [capi.HostTree(s_, B_MAX, B_MIN, 2) for s_ in pb["query_scans"]]
    plain = dict(interleave_ranges=0, deal_trees=0, deep_min_leaves=1 << 20, leaf_major=0)
    default = dict(interleave_ranges=2, deal_trees=2, deep_min_leaves=512, leaf_major=8192)
    for k_, v_ in default.items():
        assert ctx.get_option(k_) == v_, k_
    try:
        for trial in range(40):
            K, B, iters = int(rng.integers(1, 41)), int(rng.integers(1, 9)), int(rng.choice([2, 5, 15]))
            cut = int(rng.integers(200, qh[0].num_leaves))
            tids = [int(t) for t in rng.choice(all_tids, K, replace=False)]
            leaves = [h.leaf_means()[: cut + 37 * b] for b, h in enumerate(qh[:B])]
            Ls = [l.shape[0] for l in leaves]
            mids = [ctx.moving_upload(l) for l in leaves]
            X0 = np.stack([capi.pose12(T) for T in pb["query_guess"][:B]])
            res = []
            for opts in (plain, default):
                for k_, v_ in opts.items():
                    ctx.set_option(k_, v_)
                r = ctx.icp_register_batch(mids, tids, X0, PARAMS, iters)
                r["matched"] = [ctx.icp_fetch_matched(i, L) for i, L in enumerate(Ls)]
                res.append(r)
            a, b = res
            what = (trial, K, B, iters, min(Ls), max(Ls))
            assert np.array_equal(a["n_matched"], b["n_matched"]) and np.array_equal(a["visits"], b["visits"]), what
            for x, y in zip(a["matched"], b["matched"]):
                assert np.array_equal(x, y), what
            assert np.abs(a["X"] - b["X"]).max() <= 1e-11, (what, np.abs(a["X"] - b["X"]).max())
            for m in mids:
                ctx.moving_release(m)
    finally:
        for k_, v_ in default.items():
            ctx.set_option(k_, v_)
    for t_ in all_tids:
        ctx.tree_release(t_)


def test_interleaved_ranges_with_fewer_groups_than_ranges(ctx):
    """A scan of a few hundred leaves against many workgroups: most ranges of the dealt layout hold one group of 64 leaves or none
    (their virtual indices have no leaf behind them).  Same trace, flags and visit count as the contiguous layout; the visit count
    is the oracle's (mad_tree.cpp:144-152 per query)."""
    rng = np.random.default_rng(7)
    cloud = four_walls(6000)
    ht, ot = build_pair(cloud)
    tid = ctx.tree_upload(ht.nodes, ht.num_leaves)
    for n in (1, 63, 64, 65, 200, 777):
        moving = cloud[rng.choice(cloud.shape[0], n, replace=False)] + rng.normal(scale=0.01, size=(n, 3))
        mid = ctx.moving_upload(moving)
        T = np.eye(4)
        T[:3, 3] = [0.02, -0.01, 0.005]
        res = {}
        for mode in (2, 0):
            ctx.set_option("interleave_ranges", mode)
            res[mode] = (ctx.icp_linearize(mid, [tid], T, PARAMS, n), ctx.icp_register(mid, [tid], T, PARAMS, 5, n))
        ctx.set_option("interleave_ranges", 2)
        assert np.array_equal(res[2][0]["corr"], res[0][0]["corr"]), n
        assert np.array_equal(res[2][0]["matched"], res[0][0]["matched"]) and res[2][0]["visits"] == res[0][0]["visits"], n
        assert np.array_equal(res[2][1]["matched"], res[0][1]["matched"]) and res[2][1]["visits"] == res[0][1]["visits"], n
        assert np.abs(res[2][1]["X"] - res[0][1]["X"]).max() <= 1e-12, n
        leaf, depth = ot.search(moving @ T[:3, :3].T + T[:3, 3])
        assert res[2][0]["visits"] == int(depth.sum()), n
        ctx.moving_release(mid)
    ctx.tree_release(tid)


@pytest.mark.parametrize("K", [1, 4])
def test_register_pose_and_per_iteration_correspondences(ctx, K):
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, K)
    T0 = pb["query_guess"][0]
    L = qh[0].num_leaves
    o = O.icp_register(qo[0], ots, T0, 15, B_MAX, RHO_KER, B_RATIO, num_threads=2)
    g = ctx.icp_register(mids[0], tids, T0, PARAMS, 15, L)
    dt, da = pose_err(o["T"], g["T"])
    assert dt <= POSE_TOL_M and da <= POSE_TOL_RAD, (dt, da)
    # pose before every round agrees too
    for it in range(15):
        dt, da = pose_err(O.pose44(o["X_iters"][it]), capi.pose44(g["X_iters"][it]))
        assert dt <= POSE_TOL_M and da <= POSE_TOL_RAD, (it, dt, da)
    assert g["visits"] > 0
    # matched flags of the last round: identical unless a borderline pair flipped (poses differ by ~1e-15)
    assert (g["matched"] != o["matched"]).sum() <= 2
    # inject the oracle's pose of every round: correspondences must be bit-exact at that pose
    for it in (0, 7, 14):
        Tit = O.pose44(o["X_iters"][it])
        gi = ctx.icp_linearize(mids[0], tids, Tit, PARAMS, L)
        for k in range(K):
            _, _, corr, rej, _, _ = O.icp_linearize(qo[0], ots[k], Tit, B_MAX, RHO_KER, B_RATIO)
            assert np.array_equal(gi["corr"][k] & 0x7FFFFFFF, corr)
            assert np.array_equal((gi["corr"][k] >> 31).astype(np.uint8), rej)
    # converged near ground truth (sanity of the whole problem, loose)
    dt, da = pose_err(pb["query_gt"][0], g["T"])
    assert dt < 0.05 and da < 0.01
    _teardown(ctx, tids, mids)


def test_register_batch_equals_single(ctx):
    """Scans batched in flight advance like the same scans registered one by one.  The launch geometry (hence the
    order in which the per-workgroup partial sums are joined) depends on the batch size, so the two agree to
    summation-order rounding, not bit for bit; each of them is bit-reproducible on its own."""
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, 2, n_queries=3)
    X0 = np.stack([capi.pose12(T) for T in pb["query_guess"]])
    gb = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
    gb2 = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
    assert np.array_equal(gb["X"], gb2["X"]) and np.array_equal(gb["H"], gb2["H"])
    for s in range(3):
        gs = ctx.icp_register(mids[s], tids, pb["query_guess"][s], PARAMS, 15, qh[s].num_leaves)
        assert np.allclose(gb["X"][s], gs["X"], rtol=0, atol=1e-10)
        assert np.allclose(gb["H"][s], gs["H"], rtol=1e-9, atol=1e-9 * np.abs(gs["H"]).max())
        assert abs(int(gb["n_matched"][s]) - int(gs["matched"].sum())) <= 1
        assert gb["visits"][s] > 0
    _teardown(ctx, tids, mids)


@pytest.mark.parametrize("stage_min", [0, 1 << 30])
def test_more_keyframes_than_workgroups_per_scan(ctx, stage_min):
    """16 scans in flight against 20 keyframes: 16 workgroups per scan, 20 (tree, range) units — some workgroups walk
    two different trees one after the other (with threshold 0 they re-stage the LDS top between them).  Every scan
    must end where the oracle's registration of that scan ends, and like the same scan registered alone."""
    ctx.set_option("lds_stage_min_leaves", stage_min)
    try:
        pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, 20, n_queries=2)
        B = 16
        # each moving cloud may be in flight only once per batch: upload copies
        extra = [ctx.moving_upload(qh[s % 2].leaf_means()) for s in range(2, B)]
        mids16 = mids[:2] + extra
        X0 = np.stack([capi.pose12(pb["query_guess"][s % 2]) for s in range(B)])
        gb = ctx.icp_register_batch(mids16, tids, X0, PARAMS, 15)
        for s in range(2):
            o = O.icp_register(qo[s], ots, pb["query_guess"][s], 15, B_MAX, RHO_KER, B_RATIO, num_threads=4)
            gs = ctx.icp_register(mids[s], tids, pb["query_guess"][s], PARAMS, 15, qh[s].num_leaves)
            dt, da = pose_err(o["T"], gs["T"])
            assert dt < 1e-5 and da < 1e-5
            for r in range(s, B, 2):
                assert np.allclose(gb["X"][r], gs["X"], rtol=0, atol=1e-10)
                assert np.allclose(gb["H"][r], gs["H"], rtol=1e-9, atol=1e-9 * np.abs(gs["H"]).max())
        _teardown(ctx, tids, mids + extra)
    finally:
        ctx.set_option("lds_stage_min_leaves", 1024)


@pytest.mark.parametrize("stage_min", [0, 1 << 30])
def test_lds_staged_and_unstaged_descent_agree_with_oracle(ctx, stage_min):
    """The top levels of a tree walked from the LDS copy (threshold 0: always) or from global memory (never) give
    the same correspondences as the oracle, bit for bit."""
    ctx.set_option("lds_stage_min_leaves", stage_min)
    try:
        pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, 3)
        T = pb["query_guess"][0]
        L = qh[0].num_leaves
        g = ctx.icp_linearize(mids[0], tids, T, PARAMS, L)
        for k in range(3):
            _, _, corr, rej, _, _ = O.icp_linearize(qo[0], ots[k], T, B_MAX, RHO_KER, B_RATIO)
            assert np.array_equal(g["corr"][k] & 0x7FFFFFFF, corr)
            assert np.array_equal((g["corr"][k] >> 31).astype(np.uint8), rej)
        _teardown(ctx, tids, mids)
    finally:
        ctx.set_option("lds_stage_min_leaves", 1024)


def test_small_trees_fit_entirely_in_the_lds_top(ctx):
    """A tree with fewer internal nodes than the LDS top array is walked entirely in LDS (four-walls tool cloud)."""
    ctx.set_option("lds_stage_min_leaves", 0)
    try:
        np.random.seed(42)
        ref = four_walls(400)
        rt, ro = build_pair(ref)
        qt = capi.HostTree(ref + 0.01, B_MAX, B_MIN, 0)
        qo = O.Tree(ref + 0.01, B_MAX, B_MIN, 0)
        assert rt.num_leaves < 2000
        tid = ctx.tree_upload(rt.nodes, rt.num_leaves)
        mid = ctx.moving_upload(qt.leaf_means())
        g = ctx.icp_linearize(mid, [tid], np.eye(4), PARAMS, qt.num_leaves)
        _, _, corr, rej, _, depth = O.icp_linearize(qo, ro, np.eye(4), B_MAX, RHO_KER, B_RATIO)
        assert np.array_equal(g["corr"][0] & 0x7FFFFFFF, corr) and np.array_equal((g["corr"][0] >> 31).astype(np.uint8), rej)
        assert g["visits"] == depth
        _teardown(ctx, [tid], [mid])
    finally:
        ctx.set_option("lds_stage_min_leaves", 1024)


@pytest.mark.parametrize("K", [1, 4])
def test_correspondence_reuse_is_exact(ctx, K):
    """Reusing a correspondence in a later GN round when its margin proves it unchanged must not change a single
    bit: same poses before every round, same H, b, flags and the same count of visited nodes as walking every time."""
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, K)
    L = qh[0].num_leaves
    res = {}
    for on in (1, 0):
        ctx.set_option("cache_correspondences", on)
        res[on] = ctx.icp_register(mids[0], tids, pb["query_guess"][0], PARAMS, 15, L)
    ctx.set_option("cache_correspondences", 1)
    for key in ("X", "X_iters", "H", "b", "matched"):
        assert np.array_equal(res[1][key], res[0][key]), key
    assert res[1]["visits"] == res[0]["visits"]
    # ... and the gate reuse that rides on it (round 5: a pair that keeps its leaf and was rejected with more slack than it has
    # moved since is not evaluated again — no leaf record, no gate; mad_icp.cpp:81-83): the same bits without it
    assert ctx.get_option("cache_gate") == 1
    ctx.set_option("cache_gate", 0)
    res[2] = ctx.icp_register(mids[0], tids, pb["query_guess"][0], PARAMS, 15, L)
    ctx.set_option("cache_gate", 1)
    for key in ("X", "X_iters", "H", "b", "matched"):
        assert np.array_equal(res[1][key], res[2][key]), ("cache_gate", key)
    assert res[1]["visits"] == res[2]["visits"]
    # a second registration of the same scan from a different start must not see stale cache entries
    T2 = pb["query_gt"][0]
    a = ctx.icp_register(mids[0], tids, T2, PARAMS, 15, L)
    ctx.set_option("cache_correspondences", 0)
    b = ctx.icp_register(mids[0], tids, T2, PARAMS, 15, L)
    ctx.set_option("cache_correspondences", 1)
    assert np.array_equal(a["X_iters"], b["X_iters"]) and a["visits"] == b["visits"]
    _teardown(ctx, tids, mids)


def test_deep_launches_leaf_major_rounds(ctx):
    """A batch that shares the chip (32 keyframes, 8 scans in flight: 32 workgroups per scan, four trees per workgroup) is a
    DEEP launch: every workgroup gets one range of the scan and all the trees of its XCD piece, and the rounds that follow a
    round with few walkers run LEAF-MAJOR (icp_leaf_major.inc.h; option "leaf_major"): the moving leaf once per pass for all
    the workgroup's trees, the pairs that still walk queued per wavefront and walked densely.  Same DECISIONS as the tree-major
    order, bit for bit — matched flags, matched counts, the visit counter (the reference's count: mad_tree.cpp:144-152 per pair)
    — whatever the threshold (default, every round from round 2, never), with and without gate reuse, and as without any
    correspondence reuse; H, b and the pose agree to 1e-12 (another summation order, nothing else); bit-reproducible; and the
    poses are the oracle's (mad_icp.cpp:74-117 under pipeline.cpp:166-193)."""
    pb = street_problem(32, n_queries=8)
    tids, ots = [], []
    for s_, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        ht, ot = build_pair(s_, T=T)
        ots.append(ot)
        tids.append(ctx.tree_upload(ht.nodes, ht.num_leaves))
    qh = [capi.HostTree(s_, B_MAX, B_MIN, 2) for s_ in pb["query_scans"]]
    mids = [ctx.moving_upload(h.leaf_means()) for h in qh]
    X0 = np.stack([capi.pose12(T) for T in pb["query_guess"]])
    assert min(h.num_leaves for h in qh) >= 2 * 4 * 768  # (ranges of at least two passes with four workgroups per XCD piece: a DEEP launch)
    assert ctx.get_option("leaf_major") == 8192
    res = {}
    for name, opts in (("default", dict()), ("default again", dict()), ("every round", dict(leaf_major=1 << 20)),
                       ("every round, no gate reuse", dict(leaf_major=1 << 20, cache_gate=0)), ("never", dict(leaf_major=0)),
                       ("never, no gate reuse", dict(leaf_major=0, cache_gate=0)), ("no reuse", dict(cache_correspondences=0))):
        for k_, v_ in opts.items():
            ctx.set_option(k_, v_)
        r = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
        r["matched"] = [ctx.icp_fetch_matched(i, h.num_leaves) for i, h in enumerate(qh)]
        res[name] = r
        ctx.set_option("leaf_major", 8192)
        ctx.set_option("cache_correspondences", 1)
        ctx.set_option("cache_gate", 1)
    ref = res["default"]
    for key in ("X", "H", "b", "n_matched", "visits"):
        assert np.array_equal(ref[key], res["default again"][key]), key  # deterministic
        assert np.array_equal(res["never"][key], res["never, no gate reuse"][key]), key  # gate reuse: the same bits
    # (in leaf-major rounds the walkers are added last: with gate reuse a rejected pair whose slack has worn off walks again — one
    # threshold per pair since round 6 — so the ORDER of the sums differs from the run without it, nothing else)
    a, b = res["every round"], res["every round, no gate reuse"]
    assert np.array_equal(a["n_matched"], b["n_matched"]) and np.array_equal(a["visits"], b["visits"])
    assert np.abs(a["X"] - b["X"]).max() <= 1e-12 and np.abs(a["H"] - b["H"]).max() <= 1e-12 * np.abs(a["H"]).max()
    for other in ("every round", "every round, no gate reuse", "never", "never, no gate reuse", "no reuse"):
        o = res[other]
        assert np.array_equal(ref["n_matched"], o["n_matched"]) and np.array_equal(ref["visits"], o["visits"]), other
        for a, b in zip(ref["matched"], o["matched"]):
            assert np.array_equal(a, b), other
        assert np.abs(ref["X"] - o["X"]).max() <= 1e-12, (other, np.abs(ref["X"] - o["X"]).max())
        assert np.abs(ref["H"] - o["H"]).max() <= 1e-12 * np.abs(ref["H"]).max(), other
        assert np.abs(ref["b"] - o["b"]).max() <= 1e-10 * max(1.0, np.abs(ref["b"]).max()), other
    for q in (0, 5):
        o = O.icp_register(O.Tree(pb["query_scans"][q], B_MAX, B_MIN, 2), ots, pb["query_guess"][q], 15, B_MAX, RHO_KER, B_RATIO,
                           num_threads=4)
        d = np.linalg.inv(o["T"]) @ capi.pose44(ref["X"][q])
        assert np.linalg.norm(d[:3, 3]) <= 1e-5 and np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1)) <= 1e-5
        assert np.array_equal(ref["matched"][q], o["matched"])
        assert ref["visits"][q] == o["depth_sum"]
    _teardown(ctx, tids, mids)


def test_register_is_deterministic_and_graph_equals_eager(ctx):
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, 2)
    L = qh[0].num_leaves
    a = ctx.icp_register(mids[0], tids, pb["query_guess"][0], PARAMS, 15, L)
    b = ctx.icp_register(mids[0], tids, pb["query_guess"][0], PARAMS, 15, L)
    assert np.array_equal(a["X"], b["X"]) and np.array_equal(a["H"], b["H"])
    ctx.set_option("use_graph", 0)
    c = ctx.icp_register(mids[0], tids, pb["query_guess"][0], PARAMS, 15, L)
    ctx.set_option("use_graph", 1)
    assert np.array_equal(a["X"], c["X"]) and np.array_equal(a["H"], c["H"])
    _teardown(ctx, tids, mids)


@pytest.mark.parametrize("option", ["persistent", "xcd_fold"])
@pytest.mark.parametrize("K", [1, 3])
def test_persistent_rounds_are_bit_identical(ctx, K, option):
    """Option "persistent": all GN rounds of a registration as ONE launch (icp_persist — the workgroups exchange their
    adders inside the launch, in the two levels of the canonical summation order) must give the bits of the per-round
    launches: pose before every round, final pose, H, b, matched flags, visit counter — single scan, a batch, streamed —
    and agree with the oracle like them (mad_icp.cpp:105-117).  Speed is another matter (profiles/r3_b_persist_negative.md).
    Option "xcd_fold" — per-round launches whose group leaders fold their XCD's rows at the end of the launch — likewise
    (profiles/r3_j_xcd_fold_negative.md)."""
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, K, n_queries=3)
    L = qh[0].num_leaves
    T0 = pb["query_guess"][0]
    X0 = np.stack([capi.pose12(T) for T in pb["query_guess"]])
    res = {}
    for persist in (0, 1):
        ctx.set_option(option, persist)
        one = ctx.icp_register(mids[0], tids, T0, PARAMS, 15, L)
        two = ctx.icp_register(mids[0], tids, T0, PARAMS, 2, L)
        bat = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
        tk = ctx.stream_submit(qh[0].leaf_means(), tids, T0, PARAMS, 15)
        stm = ctx.stream_collect(tk, L)
        res[persist] = (one, two, bat, stm)
    ctx.set_option(option, 0)
    for a, b in zip(res[0], res[1]):
        for key in a:
            if isinstance(a[key], np.ndarray):
                assert np.array_equal(a[key], b[key]), key
            else:
                assert a[key] == b[key], key
    o = O.icp_register(qo[0], ots, T0, 15, B_MAX, RHO_KER, B_RATIO, num_threads=1)
    d = np.linalg.inv(o["T"]) @ res[1][0]["T"]
    assert np.linalg.norm(d[:3, 3]) <= 1e-5 and np.abs(d[:3, :3] - np.eye(3)).max() <= 1e-5
    assert np.array_equal(res[1][0]["matched"], o["matched"])
    _teardown(ctx, tids, mids)


def test_trusted_upload_equals_validated_upload(ctx):
    """madicp_tree_upload_trusted (no host validation pass; rho2 from the builder) must give the tree madicp_tree_upload
    gives: same node array back, same registration bit for bit."""
    pb = street_problem(2)
    L = None
    res = []
    for trusted in (False, True):
        tids = []
        for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
            ht = capi.HostTree(s, B_MAX, B_MIN, 2)
            ht.transform(T[:3, :3], T[:3, 3])
            tids.append(ctx.upload(ht, trusted=trusted))
        qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
        L = qh.num_leaves
        mid = ctx.moving_upload(qh.leaf_means())
        res.append((ctx.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, L), ctx.tree_download(tids[0], ht.num_nodes if False else capi.HostTree(pb["keyframe_scans"][0], B_MAX, B_MIN, 2).num_nodes)))
        _teardown(ctx, tids, [mid])
    for key in ("X", "H", "b", "matched", "X_iters"):
        assert np.array_equal(res[0][0][key], res[1][0][key]), key
    assert res[0][0]["visits"] == res[1][0]["visits"]
    assert res[0][1].tobytes() == res[1][1].tobytes()
    with pytest.raises(capi.MadIcpError):
        ctx.tree_upload_trusted(capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 0).nodes, 3, 1.0)  # n_nodes != 2 n_leaves - 1


def test_match_all_rounds_is_the_or_over_the_rounds(ctx):
    """Option "match_all_rounds" (what Pipeline uses when its realtime budget cuts the loop short): the reference resets
    matched_ only in iteration MAX_ICP_ITS - 1 (pipeline.cpp:172-176), so after an early break a leaf counts as matched if
    ANY round that ran matched it.  Checked against the oracle's per-round flags at the poses the device reports."""
    pb, hts, ots, tids, qh, qo, mids = _setup_registration(ctx, 2)
    L = qh[0].num_leaves
    T0 = pb["query_guess"][0]
    last_only = ctx.icp_register(mids[0], tids, T0, PARAMS, 5, L)
    ctx.set_option("match_all_rounds", 1)
    try:
        g = ctx.icp_register(mids[0], tids, T0, PARAMS, 5, L)
        tk = ctx.stream_submit(qh[0].leaf_means(), tids, T0, PARAMS, 5)
        st = ctx.stream_collect(tk, L)
    finally:
        ctx.set_option("match_all_rounds", 0)
    assert np.array_equal(g["X"], last_only["X"]) and np.array_equal(g["H"], last_only["H"])  # only the flags differ
    want = np.zeros(L, np.uint8)
    for r in range(5):
        for k in range(len(ots)):
            _, _, _, _, mat, _ = O.icp_linearize(qo[0], ots[k], capi.pose44(g["X_iters"][r]), B_MAX, RHO_KER, B_RATIO)
            want |= mat
    assert np.array_equal(g["matched"], want)
    assert np.array_equal(st["matched"], want) and st["n_matched"] == int(want.sum())
    assert (want != last_only["matched"]).any()  # (the first rounds, half a metre off, match other leaves than the last)
    _teardown(ctx, tids, mids)


def test_pairwise_registration_known_answer(ctx):
    """apps/utils/tools/mad_registration.py:51-68 — query = copy of reference, guess = euler-xyz(0.1,0.1,0.1)
    + rand(3) translation drawn after the cloud; ground truth is the identity."""
    from scipy.spatial.transform import Rotation

    np.random.seed(42)
    ref = four_walls(1000)
    T = np.eye(4)
    T[:3, :3] = Rotation.from_euler("xyz", [0.1, 0.1, 0.1]).as_matrix()
    T[:3, 3] = np.random.rand(3)
    rt = capi.HostTree(ref, 0.2, 0.1, 0)
    qt = capi.HostTree(ref.copy(), 0.2, 0.1, 0)
    tid = ctx.tree_upload(rt.nodes, rt.num_leaves)
    mid = ctx.moving_upload(qt.leaf_means())
    g = ctx.icp_register(mid, [tid], T, (0.2, 0.1, 0.02), 15, qt.num_leaves)
    assert np.abs(g["T"] - np.eye(4)).max() < 1e-6  # the reference states no tolerance ("gt T = identity")
    assert g["matched"].all()
    _teardown(ctx, [tid], [mid])


def test_abi_errors_are_loud(ctx):
    with pytest.raises(capi.MadIcpError):
        ctx.tree_release(123456)
    with pytest.raises(capi.MadIcpError):
        ctx.nn_search(98765, np.zeros((4, 3)))
    with pytest.raises(capi.MadIcpError):
        ctx.icp_register(4242, [1], np.eye(4), PARAMS, 15, 10)
