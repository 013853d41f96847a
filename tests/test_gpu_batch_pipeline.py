"""Batches in flight: madicp_moving_update_async (the next batch's scans uploaded on the copy stream beside the batch the
compute stream is working on) + madicp_icp_publish_enqueue / _collect (a batch's results carried to a pinned host block by one
kernel behind it, collected by ticket after the NEXT batch has been enqueued) against the synchronous sequence
madicp_moving_update -> madicp_icp_register_batch_enqueue -> madicp_icp_fetch: the same registrations, bit for bit, in whatever
order the host interleaves them (BASELINE configs[4]'s loop in bench.py: 8 new scans in, 8 results out per step)."""
import numpy as np
import pytest

from fixtures import B_MAX, B_MIN, PARAMS, street_problem
from mad_icp_amd import capi

pytestmark = pytest.mark.gpu
N_ITERS = 15


def test_pipelined_batches_are_the_synchronous_batches(ctx):
    B, n_steps = 3, 7
    pb = street_problem(4, n_queries=4)
    tids = []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        ht = capi.HostTree(s, B_MAX, B_MIN, 2)
        ht.transform(T[:3, :3], T[:3, 3])
        tids.append(ctx.upload(ht))
    leaves = [capi.HostTree(s, B_MAX, B_MIN, 2).leaf_means() for s in pb["query_scans"]]
    guesses = np.stack([capi.pose12(T) for T in pb["query_guess"]])
    nq = len(leaves)

    def scans_of(i):
        return [(i + s) % nq for s in range(B)]

    # synchronous reference: one set of moving ids, update -> enqueue -> fetch (+ matched flags) per step
    mids = [ctx.moving_upload(leaves[0]) for _ in range(B)]
    ref = []
    for i in range(n_steps):
        q = scans_of(i)
        for s, k in enumerate(q):
            ctx.moving_update(mids[s], leaves[k])
        ctx.icp_register_batch_enqueue(mids, tids, guesses[q], PARAMS, N_ITERS)
        r = ctx.icp_fetch(B)
        r["matched"] = [ctx.icp_fetch_matched(s, leaves[k].shape[0]) for s, k in enumerate(q)]
        ref.append(r)
    # pipelined: two sets used alternately; batch i is collected after batch i + 1 has been uploaded and enqueued
    sets = [mids, [ctx.moving_upload(leaves[0]) for _ in range(B)]]
    got, prev = [], None
    for i in range(n_steps):
        q = scans_of(i)
        cur = sets[i % 2]
        for s, k in enumerate(q):
            ctx.moving_update_async(cur[s], leaves[k])
        ctx.icp_register_batch_enqueue(cur, tids, guesses[q], PARAMS, N_ITERS)
        tk = ctx.icp_publish_enqueue(B)
        if prev is not None:
            got.append(ctx.icp_publish_collect(prev, B))
        prev = tk
    got.append(ctx.icp_publish_collect(prev, B))
    with pytest.raises(capi.MadIcpError):
        ctx.icp_publish_collect(prev, B)  # (a ticket is collected once)
    # the ring holds four outstanding tickets: a fifth is refused until the oldest has been collected
    held = [ctx.icp_publish_enqueue(B) for _ in range(4)]
    with pytest.raises(capi.MadIcpError, match="ring full"):
        ctx.icp_publish_enqueue(B)
    for tk in held:
        assert np.array_equal(ctx.icp_publish_collect(tk, B)["X"], got[-1]["X"])  # (the last batch's results, four times)
    assert len(got) == n_steps
    for i in range(n_steps):
        for k in ("X", "H", "b", "n_matched", "visits"):
            assert np.array_equal(got[i][k], ref[i][k]), (i, k)
    # the last batch's matched flags are still those of the synchronous run
    q = scans_of(n_steps - 1)
    for s, k in enumerate(q):
        assert np.array_equal(ctx.icp_fetch_matched(s, leaves[k].shape[0]), ref[-1]["matched"][s])
    # an update of a set that a batch in flight still reads is ordered behind it by the library: same results
    ctx.icp_register_batch_enqueue(sets[0], tids, guesses[scans_of(5)], PARAMS, N_ITERS)  # (sets[0] holds step 6's scans)
    tk_a = ctx.icp_publish_enqueue(B)
    for s, k in enumerate(scans_of(2)):
        ctx.moving_update_async(sets[0][s], leaves[k])  # rewritten while the batch above may still be running
    ctx.icp_register_batch_enqueue(sets[0], tids, guesses[scans_of(2)], PARAMS, N_ITERS)
    tk_b = ctx.icp_publish_enqueue(B)
    a, b = ctx.icp_publish_collect(tk_a, B), ctx.icp_publish_collect(tk_b, B)
    assert np.array_equal(b["X"], ref[2]["X"]) and np.array_equal(b["n_matched"], ref[2]["n_matched"])
    # (batch a: step 6's scans from step 5's guesses — only that it completed with finite poses)
    assert np.isfinite(a["X"]).all()
    for t in tids:
        ctx.tree_release(t)
    for m in sets[0] + sets[1]:
        ctx.moving_release(m)
