"""Multi-rank path on CPU: world_size-2 gloo processes run the staged keyframe-sharded GN loop with the oracle as
the per-rank lineariser (tests may use the oracle; the product never does) and must reproduce the single-process
oracle registration.  Also: shard plan properties, host gn_update vs the oracle's updateState."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from fixtures import B_MAX, B_MIN, B_RATIO, RHO_KER, street_problem
from mad_icp_amd import capi, sharded


def test_shard_plan_partitions_keyframes():
    for K in (1, 2, 7, 16, 64):
        for W in (1, 2, 4, 8):
            parts = [sharded.shard_keyframes(K, W, r) for r in range(W)]
            flat = sorted(k for p in parts for k in p)
            assert flat == list(range(K))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        sharded.shard_keyframes(4, 2, 2)
    # rows of `world` ids in alternating direction: BASELINE configs[3] pairs the newest keyframe with the oldest ...
    assert [sharded.shard_keyframes(16, 8, r) for r in range(8)] == [[r, 15 - r] for r in range(8)]
    # ... and a keyframe never changes hands while the window slides (ownership is a function of the id alone)
    for W in (2, 4, 8):
        for k in range(100):
            assert k in sharded.shard_keyframes(k + 1, W, sharded.keyframe_owner(k, W))
            assert k in sharded.shard_keyframes(k + 37, W, sharded.keyframe_owner(k, W))


def test_host_gn_update_matches_oracle_update_state(natives):
    pb = street_problem(2)
    trees = []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        t = O.Tree(s, B_MAX, B_MIN, 2)
        t.transform(T[:3, :3], T[:3, 3])
        trees.append(t)
    q = O.Tree(pb["query_scans"][0], B_MAX, B_MIN, 2)
    T0 = pb["query_guess"][0]
    H = np.zeros((6, 6))
    b = np.zeros(6)
    for t in trees:
        Hk, bk, *_ = O.icp_linearize(q, t, T0, B_MAX, RHO_KER, B_RATIO)
        H += Hk
        b += bk
    X1 = capi.gn_update(H, b, capi.pose12(T0))
    r = O.icp_register(q, trees, T0, 2, B_MAX, RHO_KER, B_RATIO, num_threads=1)  # X_iters[1] = pose after round 0
    assert np.allclose(X1, r["X_iters"][1], rtol=0, atol=1e-12)


def _worker(rank, world, port, K, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pb = street_problem(K)
        mine = sharded.shard_keyframes(K, world, rank)
        trees = []
        for k in mine:
            T = pb["keyframe_poses"][k]
            t = O.Tree(pb["keyframe_scans"][k], B_MAX, B_MIN, 2)
            t.transform(T[:3, :3], T[:3, 3])
            trees.append(t)
        q = O.Tree(pb["query_scans"][0], B_MAX, B_MIN, 2)

        def linearize(X):
            H = np.zeros((6, 6))
            b = np.zeros(6)
            m = np.zeros(q.num_leaves, np.uint8)
            for t in trees:
                Hk, bk, _, _, mk, _ = O.icp_linearize(q, t, capi.pose44(X), B_MAX, RHO_KER, B_RATIO)
                H += Hk
                b += bk
                m |= mk
            return H, b, m

        reg = sharded.StagedShardedRegistration(linearize, q.num_leaves)
        r = reg.register(pb["query_guess"][0], 15)
        np.savez(out % rank, X=r["X"], H=r["H"], matched=r["matched"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("K", [3])
def test_staged_sharded_registration_world2_gloo(natives, tmp_path, K):
    world = 2
    out = str(tmp_path / "rank%d.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, K, out), nprocs=world, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    # every rank ends with the same pose, bit for bit (same all-reduced H,b, same solve)
    assert np.array_equal(r0["X"], r1["X"]) and np.array_equal(r0["matched"], r1["matched"])
    # and it is the single-process registration (summation order differs: tolerance, not bits)
    pb = street_problem(K)
    trees = []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        t = O.Tree(s, B_MAX, B_MIN, 2)
        t.transform(T[:3, :3], T[:3, 3])
        trees.append(t)
    q = O.Tree(pb["query_scans"][0], B_MAX, B_MIN, 2)
    ref = O.icp_register(q, trees, pb["query_guess"][0], 15, B_MAX, RHO_KER, B_RATIO, num_threads=1)
    d = np.linalg.inv(ref["T"]) @ capi.pose44(r0["X"])
    assert np.linalg.norm(d[:3, 3]) <= 1e-9 and np.abs(d[:3, :3] - np.eye(3)).max() <= 1e-9
    assert (r0["matched"] != ref["matched"]).sum() <= 1
    assert np.allclose(r0["H"], ref["H"], rtol=1e-9, atol=1e-9 * np.abs(ref["H"]).max())


class _FakeMailboxCtx:
    """stands in for capi.Context in the CPU test of the mailbox hand-shake: records what attach_peer_mailboxes does with it"""

    def __init__(self, rank, fail):
        self.rank, self.fail, self.calls, self.attached = rank, fail, [], None

    def set_option(self, key, value):
        self.calls.append(("set_option", key, value))

    def p2p_detach(self):
        self.calls.append(("detach",))

    def p2p_export(self):
        self.calls.append(("export",))
        if self.fail:
            raise capi.MadIcpError("madicp error -2: mailbox: no exportable fine-grained device memory on this runtime")
        return bytes([self.rank]) * 64

    def p2p_attach(self, handles, world, rank):
        self.calls.append(("attach",))
        self.attached = (list(handles), world, rank)


def _mailbox_worker(rank, world, port, failing_rank, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fake = _FakeMailboxCtx(rank, rank == failing_rank)
        res = {"raised": "", "order": "", "handles_ok": False}
        try:
            sharded.attach_peer_mailboxes(fake, allow_coarse=(failing_rank < 0))
            hs, w, r = fake.attached
            res["handles_ok"] = (w == world and r == rank and hs == [bytes([q]) * 64 for q in range(world)])
        except capi.MadIcpError as e:
            res["raised"] = str(e)
        res["order"] = ",".join(c[0] for c in fake.calls)
        np.savez(out % rank, **res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("failing_rank", [-1, 1])
def test_mailbox_handshake_world3_gloo(tmp_path, failing_rank):
    """sharded.attach_peer_mailboxes with three gloo ranks on the CPU (a stand-in context): every rank detaches, exports, gathers
    and attaches the handles in RANK order; when one rank's export fails, EVERY rank raises — with that rank's message — and
    nobody is left waiting in the gather or attaches half a session."""
    world = 3
    out = str(tmp_path / "rank%d.npz")
    port = 25500 + ((os.getpid() * 3 + failing_rank) % 3000)
    mp.spawn(_mailbox_worker, args=(world, port, failing_rank, out), nprocs=world, join=True)
    R = [np.load(out % r) for r in range(world)]
    for r, z in enumerate(R):
        if failing_rank < 0:
            assert bool(z["handles_ok"]) and str(z["raised"]) == "", (r, str(z["raised"]))
            assert str(z["order"]) == "set_option,detach,export,attach", str(z["order"])
        else:
            assert "rank %d" % failing_rank in str(z["raised"]) and "fine-grained" in str(z["raised"]), (r, str(z["raised"]))
            assert str(z["order"]) == "detach,export" and not bool(z["handles_ok"]), str(z["order"])


@pytest.mark.gpu
def test_native_rccl_path_single_rank(ctx):
    """World size 1 on the one GPU of the test box: the reduce -> ncclAllReduce -> update launch sequence must give
    exactly what the fused single-GPU sequence gives."""
    from fixtures import PARAMS

    pb = street_problem(2)
    tids = []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        ht = capi.HostTree(s, B_MAX, B_MIN, 2)
        ht.transform(T[:3, :3], T[:3, 3])
        tids.append(ctx.upload(ht))
    qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
    mid = ctx.moving_upload(qh.leaf_means())
    L = qh.num_leaves
    a = ctx.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, L)
    c2 = capi.Context(0)
    try:
        t2 = []
        for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
            ht = capi.HostTree(s, B_MAX, B_MIN, 2)
            ht.transform(T[:3, :3], T[:3, 3])
            t2.append(c2.upload(ht))
        m2 = c2.moving_upload(qh.leaf_means())
        # three scans in flight (the same leaves from three guesses), fused: as one batch, as a pair, one alone
        g3 = []
        for i in range(3):
            T = pb["query_guess"][0].copy()
            T[:3, 3] += [0.02 * i, -0.01 * i, 0.0]
            g3.append(capi.pose12(T))
        g3 = np.stack(g3)
        m3 = [c2.moving_upload(qh.leaf_means()) for _ in range(3)]
        f3 = c2.icp_register_batch(m3, t2, g3, PARAMS, 15)
        f_pair = c2.icp_register_batch(m3[1:], t2, g3[1:], PARAMS, 15)
        f_one = c2.icp_register_batch(m3[:1], t2, g3[:1], PARAMS, 15)
        c2.comm_init(capi.Context.comm_unique_id(), 1, 0)
        b = c2.icp_register(m2, t2, pb["query_guess"][0], PARAMS, 15, L)
        # ... and with the older sequence (a separate icp_reduce launch in front of every all-reduce)
        # ... and with the variant whose round kernel leaves the rank's adders itself (option "shard_tail": the workgroup that
        # draws the last ticket folds the rows; measured slower, kept bit-identical)
        c2.set_option("shard_tail", 1)
        b_red = c2.icp_register(m2, t2, pb["query_guess"][0], PARAMS, 15, L)
        c2.set_option("shard_tail", 0)
        # a sharded batch: unsplit it is the fused batch bit for bit; split in two halves on two streams every
        # half is a launch shape of its own — scan 0 alone in its half, scans 1-2 the pair
        c2.set_option("shard_split", 0)
        s3 = c2.icp_register_batch(m3, t2, g3, PARAMS, 15)
        c2.set_option("shard_split", 2)  # (2: split from two scans on; the default, 1, starts at four)
        h3 = c2.icp_register_batch(m3, t2, g3, PARAMS, 15)
        h3_matched = [c2.icp_fetch_matched(i, L) for i in range(3)]
        c2.comm_destroy()
        f_pair_again = c2.icp_register_batch(m3[1:], t2, g3[1:], PARAMS, 15)
        pair_matched = [c2.icp_fetch_matched(i, L) for i in range(2)]
    finally:
        c2.close()
    assert np.array_equal(a["X"], b["X"]) and np.array_equal(a["H"], b["H"]) and np.array_equal(a["matched"], b["matched"])
    assert np.array_equal(a["X"], b_red["X"]) and np.array_equal(a["H"], b_red["H"]) and np.array_equal(a["matched"], b_red["matched"])
    for k in ("X", "H", "b", "n_matched"):
        assert np.array_equal(s3[k], f3[k]), k
        assert np.array_equal(h3[k][0], f_one[k][0]), k
        assert np.array_equal(h3[k][1:], f_pair[k]), k
        assert np.array_equal(f_pair_again[k], f_pair[k]), k
    assert np.array_equal(h3_matched[1], pair_matched[0]) and np.array_equal(h3_matched[2], pair_matched[1])
    for t in tids:
        ctx.tree_release(t)
    ctx.moving_release(mid)


# ---- the native RCCL path with more than one rank (runs whenever >= 2 GPUs are visible; skips on a 1-GPU box) ---------
_NATIVE_WORKER = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ["MADICP_ROOT"]); sys.path.insert(0, os.path.join(os.environ["MADICP_ROOT"], "tests"))
from mad_icp_amd import capi
from fixtures import street_problem, B_MAX, B_MIN, PARAMS
rank, world, K, tmp, graph = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
pb = street_problem(max(K, 1))
ctx = capi.Context(rank)
idf = os.path.join(tmp, "uid.bin")
if rank == 0:
    uid = capi.Context.comm_unique_id()
    with open(idf + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idf + ".tmp", idf)
else:
    t0 = time.time()
    while not os.path.exists(idf):
        if time.time() - t0 > 120:
            sys.exit(7)
        time.sleep(0.05)
    uid = open(idf, "rb").read()
ctx.comm_init(uid, world, rank)
if graph:
    ctx.set_option("comm_graph", 1)
tids = []
for k in range(K):
    if k % world != rank:
        continue
    ht = capi.HostTree(pb["keyframe_scans"][k], B_MAX, B_MIN, 2)
    T = pb["keyframe_poses"][k]
    ht.transform(T[:3, :3], T[:3, 3])
    tids.append(ctx.upload(ht))
qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
mid = ctx.moving_upload(qh.leaf_means())
r = ctx.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
np.savez(os.path.join(tmp, "rank%d.npz" % rank), X=r["X"], H=r["H"], matched=r["matched"], n_local=len(tids))
ctx.comm_destroy()
ctx.close()
"""


@pytest.mark.gpu
@pytest.mark.parametrize("K,graph", [(4, 0), (1, 0), (4, 1)])
def test_native_rccl_path_two_ranks(ctx, tmp_path, K, graph):
    """Two processes, one GPU each, through libmadicp_hip.so's own communicator: keyframe trees sharded k % 2, one
    ncclAllReduce of [H b n] per round, matched flags OR-ed once (replaces mad_icp.cpp:106-109 across GPUs).  Both ranks
    must hold the same pose bit for bit and agree with the single-GPU registration to rounding (the sum order differs).
    K = 1: rank 1 owns no tree and still has to join every collective.  graph = 1: the RCCL calls captured in the
    registration's hipGraph."""
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the native multi-rank path is covered on one GPU by the 1-rank test above)")
    from fixtures import PARAMS

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_NATIVE_WORKER)
    env = dict(os.environ, MADICP_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(K), str(tmp_path), str(graph)], env=env)
             for r in range(2)]
    try:
        rcs = [p.wait(timeout=300) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs == [0, 0], rcs
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(a["X"], b["X"]) and np.array_equal(a["H"], b["H"]) and np.array_equal(a["matched"], b["matched"])
    assert int(a["n_local"]) + int(b["n_local"]) == K
    # against the single-GPU path on this process's context
    pb = street_problem(max(K, 1))
    tids = []
    for k in range(K):
        ht = capi.HostTree(pb["keyframe_scans"][k], B_MAX, B_MIN, 2)
        T = pb["keyframe_poses"][k]
        ht.transform(T[:3, :3], T[:3, 3])
        tids.append(ctx.upload(ht))
    qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
    mid = ctx.moving_upload(qh.leaf_means())
    ref = ctx.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
    d = np.linalg.inv(ref["T"]) @ capi.pose44(a["X"])
    assert np.linalg.norm(d[:3, 3]) <= 1e-9 and np.abs(d[:3, :3] - np.eye(3)).max() <= 1e-9
    assert (a["matched"] != ref["matched"]).sum() <= 1
    for t in tids:
        ctx.tree_release(t)
    ctx.moving_release(mid)


# ---- the PRODUCT's sharded path with two ranks on ONE GPU (gloo processes sharing device 0) -------------------------------
def _product_worker(rank, world, port, K, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fixtures import PARAMS

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = capi.Context(0)
    try:
        pb = street_problem(max(K, 1))
        mine = sharded.shard_keyframes(K, world, rank)
        tids = []
        for k in mine:
            T = pb["keyframe_poses"][k]
            ht = capi.HostTree(pb["keyframe_scans"][k], B_MAX, B_MIN, 2)
            ht.transform(T[:3, :3], T[:3, 3])
            tids.append(ctx.upload(ht))
        qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
        lm = qh.leaf_means()
        L = qh.num_leaves
        mid = ctx.moving_upload(lm)
        guess = pb["query_guess"][0]

        # (i) the staged driver with THE PRODUCT as this rank's lineariser (a rank without trees contributes zeros)
        def linearize(X):
            if not tids:
                return np.zeros((6, 6)), np.zeros(6), np.zeros(L, np.uint8)
            r = ctx.icp_linearize(mid, tids, capi.pose44(X), PARAMS, L, want_corr=False)
            return r["H"], r["b"], r["matched"]

        st = sharded.StagedShardedRegistration(linearize, L).register(guess, 15)

        # (ii) the library's own sharded launch sequence — icp_reduce, all-reduce, icp_round / icp_final reading the
        # reduced totals, K = 0 on a rank without trees — over the host-staged transport (RCCL refuses two ranks on one GPU)
        sharded.init_host_comm(ctx)
        nat = ctx.icp_register(mid, tids, guess, PARAMS, 15, L)
        tk = ctx.stream_submit(lm, tids, guess, PARAMS, 15)
        sm = ctx.stream_collect(tk, L)
        mid2 = ctx.moving_upload(lm[: L // 2])
        X0 = np.stack([capi.pose12(guess), capi.pose12(guess)])
        # (a batch of two as two halves on two streams — option "shard_split" = 2 — so that the split launch sequence, its second
        # stream and the one order of collectives on every rank are executed with two ranks)
        ctx.set_option("shard_split", 2)
        bt = ctx.icp_register_batch([mid, mid2], tids, X0, PARAMS, 15)
        ctx.set_option("shard_split", 0)
        bt_whole = ctx.icp_register_batch([mid, mid2], tids, X0, PARAMS, 15)
        assert np.abs(bt_whole["X"] - bt["X"]).max() <= 1e-9
        ctx.comm_destroy()
        np.savez(out % rank, st_X=st["X"], st_H=st["H"], st_matched=st["matched"], nat_X=nat["X"], nat_H=nat["H"],
                 nat_matched=nat["matched"], nat_Xi=nat["X_iters"], sm_X=sm["X"], sm_H=sm["H"], sm_matched=sm["matched"],
                 sm_n=sm["n_matched"], bt_X=bt["X"], bt_n=bt["n_matched"], n_local=len(tids))
    finally:
        ctx.close()
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("K", [4, 1])
def test_product_sharded_two_ranks_one_gpu(ctx, tmp_path, K):
    """BASELINE configs[3]'s split executed by the product with TWO ranks: keyframe trees sharded k % 2 between two
    processes (each its own capi.Context on device 0), (H, b) joined over gloo every round, matched flags OR-ed once.
    Both the staged driver (ctx.icp_linearize per round) and the library's own launch sequence (host-staged transport:
    the RCCL path's kernels and ordering, only the all-reduce itself differs) must give every rank the same pose bit for
    bit, agree with the single-context registration to 1e-9 and with the oracle to 1e-5 (mad_icp.cpp:106-109,
    pipeline.cpp:180-183).  K = 1: rank 1 owns no tree and still joins every collective."""
    from fixtures import PARAMS

    world = 2
    out = str(tmp_path / "rank%d.npz")
    port = 29500 + ((os.getpid() + 7 * K) % 2000)
    mp.spawn(_product_worker, args=(world, port, K, out), nprocs=world, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert int(r0["n_local"]) + int(r1["n_local"]) == K
    for key in ("st_X", "st_H", "st_matched", "nat_X", "nat_H", "nat_matched", "nat_Xi", "sm_X", "sm_H", "sm_matched", "sm_n",
                "bt_X", "bt_n"):
        assert np.array_equal(r0[key], r1[key]), key  # every rank holds the same state, bit for bit
    # the streamed submission is the same registration
    assert np.array_equal(r0["sm_X"], r0["nat_X"]) and np.array_equal(r0["sm_matched"], r0["nat_matched"])
    assert int(r0["sm_n"]) == int(r0["nat_matched"].sum())
    # scan 0 of a batch of two: same registration, other launch geometry (half the workgroups per scan: another summation tree)
    d0 = np.linalg.inv(capi.pose44(r0["nat_X"])) @ capi.pose44(r0["bt_X"][0])
    assert np.linalg.norm(d0[:3, 3]) <= 1e-9 and np.abs(d0[:3, :3] - np.eye(3)).max() <= 1e-9

    # the single-context registration (all K trees on one rank, fused join)
    pb = street_problem(max(K, 1))
    tids, otrees = [], []
    for k in range(K):
        T = pb["keyframe_poses"][k]
        ht = capi.HostTree(pb["keyframe_scans"][k], B_MAX, B_MIN, 2)
        ht.transform(T[:3, :3], T[:3, 3])
        tids.append(ctx.upload(ht))
        ot = O.Tree(pb["keyframe_scans"][k], B_MAX, B_MIN, 2)
        ot.transform(T[:3, :3], T[:3, 3])
        otrees.append(ot)
    qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
    qo = O.Tree(pb["query_scans"][0], B_MAX, B_MIN, 2)
    mid = ctx.moving_upload(qh.leaf_means())
    one = ctx.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
    orc = O.icp_register(qo, otrees, pb["query_guess"][0], 15, B_MAX, RHO_KER, B_RATIO, num_threads=1)

    def close(Xa, Tb, tol):
        d = np.linalg.inv(Tb) @ capi.pose44(Xa)
        return np.linalg.norm(d[:3, 3]) <= tol and np.abs(d[:3, :3] - np.eye(3)).max() <= tol

    for name in ("st_X", "nat_X"):
        assert close(r0[name], one["T"], 1e-9), name
        assert close(r0[name], orc["T"], 1e-5), name
    for name in ("st_matched", "nat_matched"):
        assert (r0[name] != one["matched"]).sum() <= 1, name
        assert (r0[name] != orc["matched"]).sum() <= 1, name
    assert np.allclose(r0["nat_H"], one["H"], rtol=1e-9, atol=1e-9 * np.abs(one["H"]).max())
    # the pose before every round, too (the reduced totals fed every solve)
    for it in range(15):
        assert close(r0["nat_Xi"][it], capi.pose44(one["X_iters"][it]), 1e-9), it
    for t in tids:
        ctx.tree_release(t)
    ctx.moving_release(mid)


def _p2p_worker(rank, world, port, K, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # two ranks on ONE GPU: each gets half of the CUs for its compute stream, so that both ranks' round kernels are resident
    # at the same time while they poll each other's mailboxes (on real hardware every rank has a GPU of its own)
    os.environ["MADICP_CU_MASK"] = "lo" if rank == 0 else "hi"
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fixtures import PARAMS

    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = capi.Context(0)
    try:
        pb = street_problem(max(K, 1), n_queries=2)
        mine = sharded.shard_keyframes(K, world, rank)
        tids = []
        for k in mine:
            T = pb["keyframe_poses"][k]
            ht = capi.HostTree(pb["keyframe_scans"][k], B_MAX, B_MIN, 2)
            ht.transform(T[:3, :3], T[:3, 3])
            tids.append(ctx.upload(ht))
        qh = [capi.HostTree(sc, B_MAX, B_MIN, 2) for sc in pb["query_scans"]]
        L = qh[0].num_leaves
        mids = [ctx.moving_upload(q.leaf_means()) for q in qh]
        guess = pb["query_guess"][0]
        sharded.init_host_comm(ctx)                 # (the mailboxes need a communicator; moving sets beyond 131 072 leaves OR their flags through it)
        ctx.set_option("comm_timeout_ms", 20000)
        ref = ctx.icp_register(mids[0], tids, guess, PARAMS, 15, L)  # icp_reduce + host all-reduce per round
        sharded.attach_peer_mailboxes(ctx, allow_coarse=True)  # (the two ranks share ONE device)
        ctx.set_option("shard_p2p", 1)
        res = []
        for rep in range(3):                        # (several registrations: the slots' registration parity alternates)
            res.append(ctx.icp_register(mids[0], tids, guess, PARAMS, 15, L))
        tk = ctx.stream_submit(qh[0].leaf_means(), tids, guess, PARAMS, 15)
        sm = ctx.stream_collect(tk, L)
        X0 = np.stack([capi.pose12(guess), capi.pose12(pb["query_guess"][1])])
        bt = ctx.icp_register_batch(mids, tids, X0, PARAMS, 15)
        odd = ctx.icp_register(mids[0], tids, guess, PARAMS, 4, L)  # (another round count: tags and slots do not assume 15)
        ctx.set_option("shard_p2p", 0)
        back = ctx.icp_register(mids[0], tids, guess, PARAMS, 15, L)
        odd_ref = ctx.icp_register(mids[0], tids, guess, PARAMS, 4, L)
        ctx.comm_destroy()
        np.savez(out % rank, ref_X=ref["X"], ref_H=ref["H"], ref_matched=ref["matched"], ref_Xi=ref["X_iters"],
                 p_X=np.stack([r["X"] for r in res]), p_H=res[0]["H"], p_matched=res[0]["matched"], p_Xi=res[0]["X_iters"],
                 sm_X=sm["X"], sm_matched=sm["matched"], bt_X=bt["X"], bt_n=bt["n_matched"], odd_X=odd["X"], odd_ref_X=odd_ref["X"],
                 back_X=back["X"], n_local=len(tids))
    finally:
        ctx.close()
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("K", [4, 1])
def test_sharded_over_peer_mailboxes_two_ranks_one_gpu(ctx, tmp_path, K):
    """Option "shard_p2p": the per-round join of the ranks' adders (mad_icp.cpp:106-109 across ranks) inside the round
    kernel's prologue, over hipIpc-mapped mailboxes — rank r's workgroup 0 stores its 30 sums as tagged granules into the
    other rank's mailbox, every workgroup polls the own mailbox and adds the rows in rank order.  Two processes, device 0,
    half of the CUs each.  Every rank must hold the same bits; the result must be the sharded registration's over the
    transport (icp_reduce + all-reduce per round: the same two-level sum, then the same rank-order sum — bit for bit), agree
    with the single-context registration to 1e-9 and with the oracle to 1e-5; streamed, batched, three registrations in a row
    and a 4-round registration; K = 1: rank 1 owns no tree and still sends its (zero) rows.  NOTHING here went over xGMI."""
    from fixtures import PARAMS

    world = 2
    out = str(tmp_path / "rank%d.npz")
    port = 31500 + ((os.getpid() + 11 * K) % 2000)
    mp.spawn(_p2p_worker, args=(world, port, K, out), nprocs=world, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert int(r0["n_local"]) + int(r1["n_local"]) == K
    for key in ("p_X", "p_H", "p_matched", "p_Xi", "sm_X", "sm_matched", "bt_X", "bt_n", "odd_X", "back_X"):
        assert np.array_equal(r0[key], r1[key]), key  # every rank the same bits
    # the mailbox join adds the ranks' totals in rank order; so does a two-rank all-reduce (a + b): the same bits
    assert np.array_equal(r0["p_X"][0], r0["ref_X"]) and np.array_equal(r0["p_H"], r0["ref_H"])
    assert np.array_equal(r0["p_Xi"], r0["ref_Xi"]) and np.array_equal(r0["p_matched"], r0["ref_matched"])
    assert np.array_equal(r0["p_X"][1], r0["p_X"][0]) and np.array_equal(r0["p_X"][2], r0["p_X"][0])
    assert np.array_equal(r0["sm_X"], r0["p_X"][0]) and np.array_equal(r0["sm_matched"], r0["p_matched"])
    assert np.array_equal(r0["odd_X"], r0["odd_ref_X"]) and np.array_equal(r0["back_X"], r0["ref_X"])
    d0 = np.linalg.inv(capi.pose44(r0["p_X"][0])) @ capi.pose44(r0["bt_X"][0])
    assert np.linalg.norm(d0[:3, 3]) <= 1e-9 and np.abs(d0[:3, :3] - np.eye(3)).max() <= 1e-9
    # the single-context registration and the oracle
    pb = street_problem(max(K, 1), n_queries=2)
    tids, otrees = [], []
    for k in range(K):
        T = pb["keyframe_poses"][k]
        ht = capi.HostTree(pb["keyframe_scans"][k], B_MAX, B_MIN, 2)
        ht.transform(T[:3, :3], T[:3, 3])
        tids.append(ctx.upload(ht))
        ot = O.Tree(pb["keyframe_scans"][k], B_MAX, B_MIN, 2)
        ot.transform(T[:3, :3], T[:3, 3])
        otrees.append(ot)
    qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
    mid = ctx.moving_upload(qh.leaf_means())
    one = ctx.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
    orc = O.icp_register(O.Tree(pb["query_scans"][0], B_MAX, B_MIN, 2), otrees, pb["query_guess"][0], 15, B_MAX, RHO_KER, B_RATIO,
                         num_threads=1)
    for T_, tol in ((one["T"], 1e-9), (orc["T"], 1e-5)):
        d = np.linalg.inv(T_) @ capi.pose44(r0["p_X"][0])
        assert np.linalg.norm(d[:3, 3]) <= tol and np.abs(d[:3, :3] - np.eye(3)).max() <= tol
    assert (r0["p_matched"] != orc["matched"]).sum() <= 1
    for t in tids:
        ctx.tree_release(t)
    ctx.moving_release(mid)


@pytest.mark.gpu
def test_peer_mailboxes_with_one_rank_and_a_missing_peer(natives):
    """A world of one over the mailboxes is the fused registration bit for bit (nothing to wait for); and a rank whose peer
    never sends runs into comm_timeout_ms: MADICP_ERR_COMM (-3), context usable afterwards."""
    from fixtures import PARAMS

    pb = street_problem(2)
    c = capi.Context(0)
    try:
        tids = []
        for s_, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
            ht = capi.HostTree(s_, B_MAX, B_MIN, 2)
            ht.transform(T[:3, :3], T[:3, 3])
            tids.append(c.upload(ht))
        qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
        mid = c.moving_upload(qh.leaf_means())
        ref = c.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
        c.comm_init_host(1, 0, lambda arr, kind: None)
        c.set_option("p2p_allow_coarse", 1)
        h = c.p2p_export()
        assert len(h) == 64
        c.p2p_attach([h], 1, 0)
        with pytest.raises(capi.MadIcpError, match="detach first"):
            c.p2p_export()  # (peers of a running session still write the mailbox)
        c.set_option("shard_p2p", 1)
        solo = c.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
        assert np.array_equal(solo["X"], ref["X"]) and np.array_equal(solo["H"], ref["H"])
        assert np.array_equal(solo["matched"], ref["matched"]) and np.array_equal(solo["X_iters"], ref["X_iters"])
        c.comm_destroy()
        with pytest.raises(capi.MadIcpError, match="communicator"):
            c.p2p_attach([h], 1, 0)  # (needs a communicator)
        # rank 0 of a pretended world of two whose rank 1 never shows up: its own mailbox stands in for the peer's (a second
        # mapping of an allocation in the process that made it is refused by hipIpc), so the stores succeed and the polls
        # of rank 1's row run out
        c.comm_init_host(2, 0, lambda arr, kind: None)
        c.set_option("comm_timeout_ms", 300)
        with pytest.raises(capi.MadIcpError, match="export again"):
            c.p2p_attach([h, h], 2, 0)  # (a session starts with a fresh — zeroed — export)
        h = c.p2p_export()
        try:
            c.p2p_attach([h, h], 2, 0)
            attached = True
        except capi.MadIcpError:
            attached = False  # (this runtime refuses to open the process's own handle: nothing to test here)
        if attached:
            with pytest.raises(capi.MadIcpError, match="error -3"):
                c.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
            with pytest.raises(capi.MadIcpError, match="lost a peer"):  # the session is over: refused before anything is launched
                c.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
        c.set_option("shard_p2p", 0)
        c.set_option("comm_timeout_ms", 60000)
        c.comm_destroy()
        c.synchronize()
        again = c.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
        assert np.array_equal(again["X"], ref["X"])
    finally:
        c.close()


@pytest.mark.gpu
def test_host_transport_failure_is_comm_error(natives):
    """A transport that fails must surface as MADICP_ERR_COMM (-3), not hang or crash, and leave the context usable."""
    from fixtures import PARAMS

    pb = street_problem(2)
    c = capi.Context(0)
    try:
        tids = []
        for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
            ht = capi.HostTree(s, B_MAX, B_MIN, 2)
            ht.transform(T[:3, :3], T[:3, 3])
            tids.append(c.upload(ht))
        qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
        mid = c.moving_upload(qh.leaf_means())
        ref = c.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
        calls = []

        def broken(arr, kind):
            calls.append(kind)
            if len(calls) == 3:
                raise RuntimeError("link down")

        c.comm_init_host(1, 0, broken)
        with pytest.raises(capi.MadIcpError, match="error -3"):
            c.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
        assert len(calls) == 3
        c.comm_destroy()
        c.synchronize()
        again = c.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
        assert np.array_equal(again["X"], ref["X"])
        # a world of one over an identity transport is the fused path, bit for bit
        c.comm_init_host(1, 0, lambda arr, kind: None)
        solo = c.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, qh.num_leaves)
        c.comm_destroy()
        assert np.array_equal(solo["X"], ref["X"]) and np.array_equal(solo["H"], ref["H"])
        assert np.array_equal(solo["matched"], ref["matched"])
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_stream_collect_wait_modes(natives, mode):
    """Option "wait_mode": spinning, yielding and sleeping collections return the same registration; a bounded wait
    that runs out says MADICP_ERR_TIMEOUT (-5) and the ticket can be collected again."""
    from fixtures import PARAMS

    pb = street_problem(2)
    c = capi.Context(0)
    try:
        tids = []
        for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
            ht = capi.HostTree(s, B_MAX, B_MIN, 2)
            ht.transform(T[:3, :3], T[:3, 3])
            tids.append(c.upload(ht))
        qh = capi.HostTree(pb["query_scans"][0], B_MAX, B_MIN, 2)
        lm, L = qh.leaf_means(), qh.num_leaves
        mid = c.moving_upload(lm)
        ref = c.icp_register(mid, tids, pb["query_guess"][0], PARAMS, 15, L)
        c.set_option("wait_mode", mode)
        c.set_option("wait_timeout_ms", 5000)
        for _ in range(3):
            tk = c.stream_submit(lm, tids, pb["query_guess"][0], PARAMS, 15)
            r = c.stream_collect(tk, L)
            assert np.array_equal(r["X"], ref["X"]) and np.array_equal(r["matched"], ref["matched"])
        with pytest.raises(capi.MadIcpError):
            c.set_option("wait_mode", 3)
    finally:
        c.close()
