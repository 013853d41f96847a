"""Keyframe-sharded multi-GPU registration (BASELINE configs[3]).

The contribution of every keyframe tree to (H, b) is independent (the reference already runs them on different
OpenMP threads, pipeline.cpp:180-183) and the join is a plain sum (mad_icp.cpp:106-109), so keyframe trees are
sharded across ranks — round-robin by keyframe index — the moving leaves and the pose are replicated, and each GN
round ends with ONE small all-reduce of [H, b]; the matched flags are OR-ed (MAX) once, after the last round.
Every rank then solves the 6x6 redundantly and holds the same pose.

Three transports (and, on top of native or host, the peer MAILBOXES of `attach_peer_mailboxes`: option "shard_p2p" moves the
per-round join into the round kernel itself — stores into the peers' hipIpc-mapped mailboxes, polls of the own one — so that
only the matched flags still go through the transport, once per registration):
  * native  — `init_native_comm(ctx)`: the all-reduces are RCCL calls enqueued by libmadicp_hip.so on its own HIP
              stream between the kernels of a round (no host round trip); torch.distributed only carries the
              128-byte ncclUniqueId once.  This is what bench.py --gpus N uses.
  * host    — `init_host_comm(ctx)`: the SAME launch sequence inside the library (icp_reduce -> all-reduce -> icp_round
              reading the reduced totals, a rank without trees joining with zeros), but the all-reduce is a callback
              into torch.distributed on host memory (madicp_comm_init_host).  Works with any backend and with ranks
              that share one GPU — which RCCL refuses — so it is how the multi-rank product path is tested on a
              1-GPU box, and a fallback where no xGMI/RDMA path connects the ranks.
  * staged  — `StagedShardedRegistration`: one madicp_icp_linearize per round, (H,b) all-reduced through
              torch.distributed (any backend: nccl on GPUs, gloo in the CPU tests), host-side updateState.  Slower
              (host round trip per round) but backend-agnostic; the CPU tests drive it with an injected linearise
              function to check the sharding and the collective logic with world_size 2.
"""
import numpy as np

from . import capi


def keyframe_owner(k, world_size):
    """The rank that owns keyframe id `k` (promotion order, pipeline.cpp:253): the ids are dealt in rows of `world_size`,
    alternate rows in opposite directions — a function of the id alone, so a keyframe never changes hands while the window
    slides.  Along a trajectory the newest keyframes are the dear ones (at BASELINE configs[4] the accepted pairs grow by a
    quarter every four keyframes): plain round-robin hands the last rank a tree world_size - 1 places newer than the first
    rank's in EVERY row and every round then waits for that rank; alternating rows pair the newest of one row with the oldest
    of the next (the same deal the library makes over its eight XCD pieces: option deal_trees)."""
    row, col = divmod(int(k), world_size)
    return col if row % 2 == 0 else world_size - 1 - col


def shard_keyframes(n_keyframes, world_size, rank):
    """Keyframe indices owned by `rank` (keyframe_owner)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return [k for k in range(n_keyframes) if keyframe_owner(k, world_size) == rank]


def init_native_comm(ctx, group=None, device=None):
    """Create the RCCL communicator inside `ctx`: rank 0 makes the unique id, torch.distributed broadcasts it."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    uid = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(capi.Context.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ctx.comm_init(bytes(uid.cpu().numpy().tobytes()), world, rank)
    return rank, world


def init_host_comm(ctx, group=None):
    """Give `ctx` a host-staged transport over `group` (any torch.distributed backend): see madicp_comm_init_host."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    on_gpu = dist.get_backend(group) == "nccl"

    def all_reduce(arr, kind):
        t = torch.from_numpy(arr)  # shares the library's pinned staging buffer
        op = dist.ReduceOp.SUM if kind == capi.REDUCE_SUM_F64 else dist.ReduceOp.MAX
        if on_gpu:
            g = t.cuda()
            dist.all_reduce(g, op=op, group=group)
            t.copy_(g.cpu())
        else:
            dist.all_reduce(t, op=op, group=group)

    ctx.comm_init_host(world, rank, all_reduce)
    return rank, world


def attach_peer_mailboxes(ctx, group=None, allow_coarse=False):
    """Option "shard_p2p": every rank exports its (freshly zeroed) mailbox, torch.distributed gathers the 64-byte handles — the
    point every rank passes between zeroing its mailbox and the first registration of the session — every rank maps the others'
    (madicp_p2p_export / madicp_p2p_attach).  Needs a communicator in `ctx` already (init_native_comm / init_host_comm); ranks
    on one node (hipIpc).  After this, `ctx.set_option("shard_p2p", 1)` makes the per-round join of the ranks' adders AND the
    OR of the matched flags happen inside the registration's own kernels — no collective at all.  Call it again to start a
    new session (after a MADICP_ERR_COMM, or with other ranks).  A rank whose export fails does not leave its peers waiting:
    the ranks agree and every one raises.  `allow_coarse`: accept a coarse-grained mailbox (ranks that share ONE device: the
    one-GPU tests and bench.py under MADICP_BENCH_BACKEND=gloo)."""
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    err = None
    mine = b""
    try:
        if allow_coarse:
            ctx.set_option("p2p_allow_coarse", 1)
        ctx.p2p_detach()  # (a no-op without a running session)
        mine = ctx.p2p_export()
    except capi.MadIcpError as e:
        err = "rank %d: %s" % (rank, e)
    got = [None] * world
    dist.all_gather_object(got, (err, mine), group=group)
    errs = [e for e, _ in got if e]
    if errs:
        raise capi.MadIcpError("peer mailboxes: " + "; ".join(errs))
    ctx.p2p_attach([h for _, h in got], world, rank)
    return rank, world


class StagedShardedRegistration:
    """GN loop with the (H,b) join done by torch.distributed.

    linearize(X12) -> (H (6,6), b (6,), matched (L,) uint8) must return THIS rank's contribution at pose X12 —
    normally `lambda X: ctx.icp_linearize(mid, local_tree_ids, X, params, L)` unpacked; the tests inject a CPU one.
    """

    def __init__(self, linearize, n_moving, group=None, device="cpu"):
        self.linearize = linearize
        self.L = n_moving
        self.group = group
        self.device = device

    def register(self, X0, n_iters):
        import torch
        import torch.distributed as dist

        X = capi.pose12(X0)
        H = np.zeros((6, 6))
        b = np.zeros(6)
        matched = np.zeros(self.L, np.uint8)
        for it in range(n_iters):
            Hl, bl, ml = self.linearize(X)
            buf = torch.from_numpy(np.concatenate([np.asarray(Hl, np.float64).reshape(-1), np.asarray(bl, np.float64)])).to(self.device)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            hb = buf.cpu().numpy()
            H, b = hb[:36].reshape(6, 6).copy(), hb[36:].copy()
            if it == n_iters - 1:  # flags of the last round only (pipeline.cpp:172-176), OR over ranks
                m = torch.from_numpy(np.ascontiguousarray(ml, np.uint8)).to(self.device)
                dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
                matched = m.cpu().numpy()
            X = capi.gn_update(H, b, X)
        return dict(X=X, T=capi.pose44(X), H=H, b=b, matched=matched)
