"""The sharded launch sequence on ONE GPU with a one-rank RCCL communicator (development tool, GPU box): what the pieces of
the multi-GPU round cost before any xGMI latency is added.  Same loop for every line — B new scans in (moving_update),
register, B results out (icp_fetch) — so the lines compare like with like:

  fused            no communicator: icp_round x 15 + icp_final (what one GPU does on its own)
  shard reduce     communicator, options shard_tail = 0, shard_split = 0: icp_round, icp_reduce, all-reduce per round (round 3)
  shard tail       shard_tail = 1: the round kernel leaves the rank's adders itself, the all-reduce follows directly
  shard tail+split B >= 2: two halves on two streams, one half's all-reduce under the other half's round

usage: python tools/shard_probe.py [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mad_icp_amd import capi, synth  # noqa: E402

PARAMS = (0.2, 0.1, 0.02)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    K = 16
    pb = synth.make_problem(K, seed=1, n_queries=1)
    scans, gts, gs = synth.make_query_streams(K, seed=1, n_streams=8)
    ctx = capi.Context(0)
    tids = []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        ht = capi.HostTree(s, 0.2, 0.1, 3)
        ht.transform(T[:3, :3], T[:3, 3])
        tids.append(ctx.upload(ht))
    leaves = [capi.HostTree(s, 0.2, 0.1, 3).leaf_means() for s in scans]
    guess = [capi.pose12(T) for T in gs]

    def run(B):
        mids = [ctx.moving_upload(leaves[s]) for s in range(B)]

        def one(i):
            qs = [(i * B + s) % 8 for s in range(B)]
            for s, q in enumerate(qs):
                ctx.moving_update(mids[s], leaves[q])
            ctx.icp_register_batch_enqueue(mids, tids, np.stack([guess[q] for q in qs]), PARAMS, 15)
            return qs, ctx.icp_fetch(B)

        for i in range(10):
            one(i)
        ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            last = one(i)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        for m in mids:
            ctx.moving_release(m)
        return steps * B / dt, last

    only = os.environ.get("SHARD_PROBE_B")
    sizes = tuple(int(x) for x in only.split(",")) if only else (1, 2, 8)
    ref = {}
    for B in sizes:
        v, last = run(B)
        ref[B] = last
        print("fused            B=%d  %8.1f registrations/s  %7.1f us per registration" % (B, v, 1e6 / v), flush=True)
    ctx.comm_init(capi.Context.comm_unique_id(), 1, 0)
    for name, tail, split, Bs in (("shard reduce    ", 0, 0, (1, 2, 8)), ("shard tail      ", 1, 0, (1, 2, 8)), ("shard tail+split", 1, 1, (2, 8))):
        ctx.set_option("shard_tail", tail)
        ctx.set_option("shard_split", 2 * split)
        for B in Bs:
            if B not in sizes:
                continue
            v, last = run(B)
            same = all(np.array_equal(last[1][k], ref[B][1][k]) for k in ("X", "H", "b"))
            print("%s B=%d  %8.1f registrations/s  %7.1f us per registration   results == fused: %s" % (name, B, v, 1e6 / v, same), flush=True)
    # a one-rank all-reduce launches nothing: a delay kernel of 15 us stands in for the collective's latency across xGMI
    # (option "debug_collective_us") — what the split is for
    ctx.set_option("debug_collective_us", int(os.environ.get("SHARD_PROBE_DELAY_US", "15")))
    for name, tail, split, Bs in (("+15us reduce    ", 0, 0, (1, 2, 8)), ("+15us tail      ", 1, 0, (1, 2, 8)), ("+15us tail+split", 1, 1, (2, 8))):
        ctx.set_option("shard_tail", tail)
        ctx.set_option("shard_split", 2 * split)
        for B in Bs:
            if B not in sizes:
                continue
            v, last = run(B)
            print("%s B=%d  %8.1f registrations/s  %7.1f us per registration" % (name, B, v, 1e6 / v), flush=True)
    ctx.set_option("debug_collective_us", 0)
    ctx.comm_destroy()
    ctx.close()


if __name__ == "__main__":
    main()
