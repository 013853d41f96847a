"""Generates tests/golden/*.npz from the oracle (the reference itself cannot be built or imported in this
environment — no Eigen — so these vectors pin the ORACLE and the HIP path against regressions, not the
reference binary; the header of oracle/mad_oracle.h says "parity unpinned" for that reason).

  python tests/golden/make_golden.py        # rewrites the .npz files

Inputs are regenerated from seeds by tests/fixtures.py; only small outputs are stored: poses before every GN
round, final H/b, matched count, sha256 of the per-tree correspondence arrays at the initial guess."""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as O  # noqa: E402
from fixtures import B_MAX, B_MIN, B_RATIO, RHO_KER, four_walls, street_problem  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def street(K):
    pb = street_problem(K)
    trees = []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        t = O.Tree(s, B_MAX, B_MIN, 2)
        t.transform(T[:3, :3], T[:3, 3])
        trees.append(t)
    q = O.Tree(pb["query_scans"][0], B_MAX, B_MIN, 2)
    T0 = pb["query_guess"][0]
    corr_sha, rej_sha, depth = [], [], []
    for t in trees:
        _, _, corr, rej, _, d = O.icp_linearize(q, t, T0, B_MAX, RHO_KER, B_RATIO)
        corr_sha.append(digest(corr))
        rej_sha.append(digest(rej))
        depth.append(d)
    r = O.icp_register(q, trees, T0, 15, B_MAX, RHO_KER, B_RATIO, num_threads=1)
    np.savez(os.path.join(HERE, f"street_k{K}.npz"), scan_sha=digest(pb["query_scans"][0]),
             n_leaves=q.num_leaves, tree_leaves=np.array([t.num_leaves for t in trees]), corr_sha=np.array(corr_sha),
             rej_sha=np.array(rej_sha), depth=np.array(depth), X_iters=r["X_iters"], T=r["T"], H=r["H"], b=r["b"],
             n_matched=int(r["matched"].sum()))


def baseline(K, seed, n_queries, name):
    """BASELINE.json's full-size configurations (119 725-point scans): what tests/test_gpu_baseline_configs.py checks
    the HIP path against.  Per tree: sha-256 of scan 0's correspondence / gate arrays at the initial guess; per scan:
    the pose before every round, the final pose and the matched-leaf count."""
    from mad_icp_amd import synth

    pb = synth.make_problem(K, seed=seed, n_queries=n_queries)
    trees = []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        t = O.Tree(s, B_MAX, B_MIN, 3)
        t.transform(T[:3, :3], T[:3, 3])
        trees.append(t)
    qs = [O.Tree(s, B_MAX, B_MIN, 3) for s in pb["query_scans"]]
    corr_sha, rej_sha, depth = [], [], []
    for t in trees:
        _, _, corr, rej, _, d = O.icp_linearize(qs[0], t, pb["query_guess"][0], B_MAX, RHO_KER, B_RATIO)
        corr_sha.append(digest(corr))
        rej_sha.append(digest(rej))
        depth.append(d)
    rs = [O.icp_register(q, trees, T0, 15, B_MAX, RHO_KER, B_RATIO, num_threads=8) for q, T0 in zip(qs, pb["query_guess"])]
    np.savez(os.path.join(HERE, name), scan_sha=digest(pb["query_scans"][0]), n_leaves=np.array([q.num_leaves for q in qs]),
             tree_leaves=np.array([t.num_leaves for t in trees]), corr_sha=np.array(corr_sha), rej_sha=np.array(rej_sha),
             depth=np.array(depth), X_iters=np.stack([r["X_iters"] for r in rs]), T=np.stack([r["T"] for r in rs]),
             n_matched=np.array([int(r["matched"].sum()) for r in rs]), depth_sum=np.array([r["depth_sum"] for r in rs]))


def walls():
    np.random.seed(42)
    cloud = four_walls(2000)
    t = O.Tree(cloud, 1e-5, 0.1, 2)
    rng = np.random.default_rng(9)
    q = cloud[rng.integers(0, len(cloud), 5000)] + rng.normal(0, 0.05, (5000, 3))
    leaf, depth, dist = t.search(q, want_dist=True)
    np.savez(os.path.join(HERE, "walls_nn.npz"), cloud_sha=digest(cloud), leaf_sha=digest(leaf), depth_sum=int(depth.sum()),
             dist_sum=float(dist.sum()), leaf_head=leaf[:64])


if __name__ == "__main__":
    street(1)
    street(3)
    walls()
    baseline(1, 1, 1, "baseline_k1.npz")
    baseline(16, 1, 1, "baseline_k16.npz")
    baseline(64, 2, 8, "baseline_k64_b8.npz")
    print("golden vectors written to", HERE)
