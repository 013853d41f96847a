#!/usr/bin/env python3
"""bench.py — scan registrations/s of the MAD-ICP hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2]): 120k-point KITTI-shaped synthetic scans registered against a 16-keyframe local
map, 15 Gauss-Newton rounds (tools/constants.h:34 of the reference), default parameters
(configurations/default.cfg:2-7).

A *step* is one registration as SURVEY §8(d)(i) defines it: a NEW scan's moving leaves in (host memory) -> 15 GN
rounds against the K resident keyframe trees -> X, H, b, matched flags and their count out (host memory).  A different
scan every step (8 distinct scans, cycled).  The library streams: the leaves of scan i+1 are fed on its copy stream
while scan i registers, results are written by the last kernel into pinned host memory (madicp_stream_submit /
madicp_stream_collect), so upload and read-back are INSIDE the timed region and overlap the device work.  The keyframe
trees are resident before the timed region starts (they are the local map).  The device-resident loop of round 1
(same pre-uploaded scan re-registered, nothing read back) is kept as the secondary key `resident_loop`.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1 (one rank per GPU) — `value` is BASELINE configs[3], the north-star's multi-GPU configuration:
  shard     (value) the 16 keyframe trees are dealt over the ranks (16/N per GPU, sharded.keyframe_owner); every rank holds the
            moving leaves of the N scans in flight (one per GPU: per-GPU work fixed as N grows, "scaling": "weak"),
            linearises them against ITS trees, and every GN round ends with ONE RCCL all-reduce of [H(21) b(6) n v w]
            per scan over xGMI, enqueued by the library between its kernels; the matched flags are OR-ed once.  Every
            step uploads N new scans and reads N results back.
  second keys, measured in the same run: `shard_one_scan` (ONE scan in flight, total work fixed: strong scaling — a
            registration is ~15 us of work per round, so this is latency-bound by the all-reduce) and `replica`
            (every rank holds the whole map and streams its own scans, no collective).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (icp_round: one Gauss-Newton round); see DESIGN.md
section 6 for every figure in it.  `cpu_baseline` is the CPU restatement of the reference's OpenMP path (oracle/, kind
"port") timed on this box's host cores on a bounded sample.
"""
import argparse
import csv
import gc
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (Rounds 4-5 set GPU_MAX_HW_QUEUES=8 here because the look-ahead frame seemed to depend on the runtime's queue map.  It did
# not: the slow frames were the bindings' by-value copy of the cloud on the HOST — profiles/r5_u_lookahead_canary.md — and with
# that gone the runtime's default is the better setting for the look-ahead, 0.65-0.69 against 0.86-0.88 ms.  This process now
# runs with whatever the caller's environment says, like any embedder; pipeline_end_to_end.lookahead_in_a_plain_process still
# reports both settings.)
# The MULTI-RANK branch is different: a process that also holds an RCCL communicator (its streams, torch's) has more streams than
# the runtime's default four hardware queues, and the streamed loop's upload / rounds / read-back then share queues — measured on
# one GPU with a world of one (tools/shard_world1.sh, two repetitions, same box): replica 4 027 -> 4 440, shard 3 487 -> 3 750,
# shard_p2p 3 928 -> 4 290 registrations/s with eight queues; the one-process bench does not care (4 684 / 4 623).  Read by the
# HIP runtime when it initialises, so it is set before anything imports torch; an explicit setting in the environment wins.
# Ranks that SHARE a device (MADICP_BENCH_BACKEND=gloo on a box with fewer GPUs than ranks) are the opposite case: every process's
# queues compete for the device's hardware queue slots, and once the processes together hold more than the device maps at a time
# its scheduler rotates them with a quantum of milliseconds — fatal for kernels that poll a peer's mailbox (measured: eight ranks
# x eight queues, 345 ms per sharded registration; four ranks, 9 ms).  Few queues per process there.
_world = int(os.environ.get("WORLD_SIZE", "1") or 1)
if _world > 1 and os.environ.get("MADICP_BENCH_BACKEND", "nccl") != "nccl":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(2, min(8, 24 // _world))))
elif _world > 1 or os.environ.get("MADICP_BENCH_FORCE_MULTI") == "1":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

B_MAX, B_MIN, RHO_KER, B_RATIO, N_ITERS = 0.2, 0.1, 0.1, 0.02, 15
PARAMS = (B_MAX, RHO_KER, B_RATIO)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
L2_PEAK_GBS = 34500.0  # aggregate L2 bandwidth, same guide
N_DISTINCT = 8         # distinct query scans cycled through the timed region
MIN_WARMUP = 40        # untimed steps really run before the timed region (whatever --warmup says, never fewer): "warmup_run"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--keyframes", type=int, default=16)
    ap.add_argument("--scans", type=int, default=0, help="scans in flight per step (0: 1 on one GPU, N in shard mode)")
    ap.add_argument("--mode", choices=["auto", "shard", "replica"], default="auto",
                    help="N > 1 only; auto = shard (BASELINE configs[3]) with the replica figure as a second key")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-baseline", choices=["auto", "off"], default="auto")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget of the baseline sample")
    ap.add_argument("--pmc", choices=["auto", "off"], default="auto",
                    help="measure HBM traffic of icp_round in this run (rocprofv3 --pmc sub-runs of this script)")
    ap.add_argument("--no-rebuild", action="store_true", help="do not force-rebuild the HIP library first")
    ap.add_argument("--option", action="append", default=[], help="library option key=value (tuning)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--build-child", default=None, help=argparse.SUPPRESS)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
def upload_map(ctx, capi, pb, keyframes):
    tids, n_nodes = [], 0
    for k in keyframes:
        T = pb["keyframe_poses"][k]
        ht = capi.HostTree(pb["keyframe_scans"][k], B_MAX, B_MIN, 3)
        ht.transform(T[:3, :3], T[:3, 3])
        tids.append(ctx.upload(ht))
        n_nodes += ht.num_nodes
    return tids, n_nodes


def pose_error(Tgt, T):
    err = np.linalg.inv(Tgt) @ T
    return float(np.linalg.norm(err[:3, 3]))


# the gather calibration of the PMC sub-runs (madicp_debug_gather16): region, gathers per launch, seed, launches
GATHER_REGION, GATHER_N, GATHER_SEED, GATHER_REPS = 2 << 30, 1 << 22, 7, 3


def gather_lines(region_bytes=GATHER_REGION, n=GATHER_N, seed=GATHER_SEED):
    """distinct 64-byte and 128-byte lines ONE launch of the gather probe touches (the library's hash, restated)"""
    g = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = g * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
    idx = h % np.uint64(region_bytes // 16)
    return int(np.unique(idx >> np.uint64(2)).size), int(np.unique(idx >> np.uint64(3)).size)


def streamed_loop(ctx, capi, leaves, guesses, tids, n, results=None, stamps=None):
    """n registrations, each a new scan in / results out, one submission ahead of the collection."""
    nq = len(leaves)
    prev = None
    for i in range(n):
        q = i % nq
        tk = ctx.stream_submit(leaves[q], tids, guesses[q], PARAMS, N_ITERS)
        if prev is not None:
            r = ctx.stream_collect(prev[0], leaves[prev[1]].shape[0])
            if results is not None:
                results.append((prev[1], r))
        prev = (tk, q)
        if stamps is not None:
            stamps.append(time.perf_counter())
    if prev is not None:
        r = ctx.stream_collect(prev[0], leaves[prev[1]].shape[0])
        if results is not None:
            results.append((prev[1], r))


# ---------------------------------------------------------------------------------------------------------
def pmc_child(path):
    """Sub-run under rocprofv3 --pmc: a stream copy of known size (calibration) and a few registrations of the bench
    workload, loaded from the arrays the parent saved."""
    from mad_icp_amd import capi

    z = np.load(path, allow_pickle=False)
    K = int(z["K"])
    ctx = capi.Context(0)
    ctx.stream_copy_gbs(int(z["copy_bytes"]), 3)
    # icp_round's own access pattern with a known line count: random 16-byte gathers over 2 GiB (beyond the Infinity Cache)
    ctx.gather16_us(GATHER_REGION, GATHER_N, seed=GATHER_SEED, reps=GATHER_REPS)
    tids = []
    for k in range(K):
        tids.append(ctx.tree_upload(z["nodes%d" % k].view(capi.NODE_DTYPE).reshape(-1), int(z["leaves%d" % k])))
    mid = ctx.moving_upload(z["moving"])
    X0 = z["X0"].reshape(1, 12)
    for _ in range(int(z["regs"])):
        ctx.icp_register_batch_enqueue([mid], tids, X0, PARAMS, N_ITERS)
    ctx.synchronize()
    if "nn_queries" in z.files:  # nn_descend (the pymadtree path) against the last keyframe tree
        ctx.nn_time_descend(tids[K - 1], z["nn_queries"], 6)
    if "K2" in z.files:  # the stress configuration in the same profiler pass: its launches FOLLOW the headline's
        K2, B2 = int(z["K2"]), int(z["B2"])
        for k in range(K, K2):
            tids.append(ctx.tree_upload(z["nodes%d" % k].view(capi.NODE_DTYPE).reshape(-1), int(z["leaves%d" % k])))
        mids2 = [ctx.moving_upload(z["moving2_%d" % s]) for s in range(B2)]
        X2 = z["X2"].reshape(B2, 12)
        for _ in range(int(z["regs2"])):
            ctx.icp_register_batch_enqueue(mids2, tids, X2, PARAMS, N_ITERS)
        ctx.synchronize()
    ctx.close()


def measure_traffic(pb, tids_trees, moving, X0, copy_bytes=1 << 30, regs=6, stress=None, regs2=3, nn_queries=None):
    """HBM-side traffic of icp_round per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE, TCC requests: one
    pass each, --kernel-trace only).  Each pass also runs a device-to-device copy of `copy_bytes`: the known byte
    count FETCH_SIZE / WRITE_SIZE are calibrated on (the guide: FETCH_SIZE reads half of a wide streaming read)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="madicp_pmc_", dir="/tmp")
    try:
        arrays = dict(K=len(tids_trees), moving=moving, X0=X0, copy_bytes=copy_bytes, regs=regs)
        if nn_queries is not None:
            arrays["nn_queries"] = np.ascontiguousarray(nn_queries, dtype=np.float64)
        all_trees = list(tids_trees)
        if stress is not None:  # (trees 0..K-1 are the headline's map: the stress map extends it)
            all_trees = stress["trees"]
            arrays.update(K2=len(all_trees), B2=len(stress["moving"]), X2=stress["X0"], regs2=regs2)
            for si, lm in enumerate(stress["moving"]):
                arrays["moving2_%d" % si] = lm
        for k, ht in enumerate(all_trees):
            arrays["nodes%d" % k] = np.frombuffer(ht.nodes.tobytes(), dtype=np.uint8)
            arrays["leaves%d" % k] = ht.num_leaves
        npz = os.path.join(tmp, "problem.npz")
        np.savez(npz, **arrays)
        from mad_icp_amd import _build as _b

        env = dict(os.environ, TMPDIR="/tmp", MADICP_NATIVE_DIR=_b.MEASURE_DIR)  # (the calibration probes are measurement aids)
        out = {}
        for name, counters in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"),
                               ("tcc", "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum")):
            d = os.path.join(tmp, name)
            cmd = [rocprof, "--kernel-trace", "--pmc"] + counters.split() + [
                "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", npz]
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240,
                               check=True)
            except Exception as e:  # noqa: BLE001 — any failure of the profiler leaves traffic unmeasured, never the bench
                return {"error": "rocprofv3 pass %s failed: %s" % (name, str(e)[:200])}
            acc = {}
            n_head = regs * N_ITERS  # icp_round launches of the headline problem: the first ones of the process
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    rows = sorted(csv.DictReader(fh), key=lambda r: int(r.get("Dispatch_Id", "0") or 0))
                seen_rounds = {}
                for row in rows:
                    kn = row.get("Kernel_Name", "")
                    key = ("copy" if "stream_copy" in kn else "gather" if "gather16_probe" in kn else
                           "nn" if "nn_descend" in kn else "round" if "icp_round" in kn else None)
                    if key == "round":
                        c = seen_rounds.get(row["Counter_Name"], 0)
                        seen_rounds[row["Counter_Name"]] = c + 1
                        if c >= n_head:
                            key = "round2"
                    if key:
                        acc.setdefault((key, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            for (key, cname), vals in acc.items():
                if key in ("copy", "gather"):
                    vals = vals[1:] or vals  # first launch touches cold pages
                out[(key, cname)] = float(np.mean(vals))
                out[(key, cname, "n")] = len(vals)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------
BUILD_REPS = 6


def build_child(path):
    """Sub-run under rocprofv3 (--kernel-trace; with --pmc for the counter passes): a few device MAD-tree builds of the bench scan
    and — where the measurement build is loaded (MADICP_NATIVE_DIR), for the counters' calibration — a 1 GiB copy in front."""
    from mad_icp_amd import capi

    scan = np.load(path)["scan"]
    ctx = capi.Context(0)
    if hasattr(capi.hip_lib(), "madicp_debug_stream_copy"):
        ctx.stream_copy_gbs(1 << 30, 3)
    cid = ctx.cloud_upload(scan)
    for _ in range(BUILD_REPS):
        t_, _nl = ctx.tree_build(cid, B_MAX, B_MIN)
        ctx.synchronize()
        ctx.tree_release(t_)
    ctx.close()


TB_FAMILIES = (("tb_init", "init"), ("tb_chip_stats", "chip_stats"), ("tb_chip_scatter", "chip_scatter"), ("tb_level", "level"),
               ("tb_finish", "finish"), ("tb_emit", "emit"))


def builder_roofline(scan, n_leaves, max_level):
    """Per kernel family of ONE device tree build: launches, traced time (rocprofv3 --kernel-trace of a sub-run that builds
    the bench scan's tree and nothing else; the last build of the trace), algorithmic bytes, and their rate against the HBM
    peak.  Algorithmic bytes, an UPPER bound — every point is counted alive on every level, although a point
    whose leaf is finished drops out: chip levels 0-5: a point is read by the statistics pass (24 B), read and written by the
    scatter (48 B), its rank-table entry written and read (8 B); wave / quad steps: read + written once (48 B); a split reads
    its 240-byte record and writes two; init: the cloud once + the leaf marks cleared; finish: marks twice, scan once, every
    record once; emission: every record once, 80 B out per node (+ 64 B per leaf)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="madicp_tb_", dir="/tmp")
    try:
        npz = os.path.join(tmp, "scan.npz")
        np.savez(npz, scan=scan)
        d = os.path.join(tmp, "trace")
        cmd = [rocprof, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable, os.path.abspath(__file__),
               "--build-child", npz]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=240, check=True)
        except Exception as e:  # noqa: BLE001
            return {"error": "rocprofv3 kernel trace of the builder failed: %s" % str(e)[:160]}
        files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if not files:
            return {"error": "no kernel trace written"}
        with open(files[0]) as fh:
            rows = sorted(csv.DictReader(fh), key=lambda r: int(r["Start_Timestamp"]))
        starts = [i for i, r in enumerate(rows) if "tb_init" in r["Kernel_Name"]]
        if not starts:
            return {"error": "no tb_init in the trace"}
        seg = rows[starts[-1]:]
        end = max((i for i, r in enumerate(seg) if "tb_emit" in r["Kernel_Name"]), default=len(seg) - 1)
        seg = seg[: end + 1]
        N, L = int(scan.shape[0]), int(n_leaves)
        n_nodes, n_int = 2 * L - 1, L - 1
        chip_levels, steps = 6, max(0, int(max_level) + 1 - 6)
        alg = {"init": N * 24 + N * 4,
               "chip_stats": chip_levels * N * (24 + 4),
               "chip_scatter": chip_levels * N * (24 + 24 + 4),
               "level": steps * N * 48 + n_int * 3 * 240,
               "finish": N * 4 * 3 + n_nodes * 240,
               "emit": n_nodes * (240 + 80) + L * 64}
        fam = {}
        for r in seg:
            for pat, key in TB_FAMILIES:
                if pat in r["Kernel_Name"]:
                    f = fam.setdefault(key, {"launches": 0, "us": 0.0})
                    f["launches"] += 1
                    f["us"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                    break
        span_us = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
        out = {"bound": "latency", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traced_build_us": round(span_us, 1), "kernels": {}}
        for key, f in fam.items():
            gbs = alg[key] / (f["us"] * 1e-6) / 1e9 if f["us"] > 0 else 0.0
            out["kernels"][key] = {"launches": f["launches"], "us": round(f["us"], 1), "algorithmic_bytes": int(alg[key]),
                                   "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        # counter traffic per family (FETCH_SIZE / WRITE_SIZE, one pass each, calibrated on the 1 GiB copy of the same pass)
        try:
            from mad_icp_amd import _build as _b

            env = dict(os.environ, TMPDIR="/tmp", MADICP_NATIVE_DIR=_b.MEASURE_DIR)
            raw = {}
            for cname in ("FETCH_SIZE", "WRITE_SIZE"):
                dd = os.path.join(tmp, cname)
                subprocess.run([rocprof, "--kernel-trace", "--pmc", cname, "--output-format", "csv", "-d", dd, "-o", "p", "--",
                                sys.executable, os.path.abspath(__file__), "--build-child", npz], cwd="/tmp", env=env,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
                rows_c = []
                for f in glob.glob(os.path.join(dd, "**", "*counter_collection.csv"), recursive=True):
                    with open(f) as fh:
                        rows_c += [r for r in csv.DictReader(fh) if r.get("Counter_Name") == cname]
                rows_c.sort(key=lambda r: int(r.get("Dispatch_Id", "0") or 0))
                copies = [float(r["Counter_Value"]) for r in rows_c if "stream_copy" in r["Kernel_Name"]]
                last_init = max(i for i, r in enumerate(rows_c) if "tb_init" in r["Kernel_Name"])
                per = {}
                for r in rows_c[last_init:]:
                    for pat, key in TB_FAMILIES:
                        if pat in r["Kernel_Name"]:
                            per[key] = per.get(key, 0.0) + float(r["Counter_Value"]) * 1024.0
                            break
                cal = (float(1 << 30) / (np.mean(copies[1:] or copies) * 1024.0)) if copies else 1.0
                raw[cname] = (per, cal)
            for key in out["kernels"]:
                f_raw, w_raw = raw["FETCH_SIZE"][0].get(key, 0.0), raw["WRITE_SIZE"][0].get(key, 0.0)
                tr = f_raw * raw["FETCH_SIZE"][1] + w_raw * raw["WRITE_SIZE"][1]
                k_ = out["kernels"][key]
                k_["traffic"] = int(tr)
                k_["traffic_frac_of_hbm_peak"] = round(tr / (k_["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if k_["us"] > 0 else None
            out["traffic"] = int(sum(k_.get("traffic", 0) for k_ in out["kernels"].values()))
            out["traffic_calibration"] = {"fetch": round(raw["FETCH_SIZE"][1], 3), "write": round(raw["WRITE_SIZE"][1], 3),
                                          "on": "a 1 GiB device-to-device copy in the same rocprofv3 pass (the guide's streaming "
                                                "correction; counts Infinity-Cache hits: an upper bound of HBM traffic)"}
        except Exception as e:  # noqa: BLE001 — the counters are a secondary figure
            out["traffic"] = None
            out["traffic_error"] = str(e)[:160]
        tot = sum(alg[k] for k in fam)
        out["algorithmic_bytes"] = int(tot)
        out["achieved"] = round(tot / (span_us * 1e-6) / 1e9, 1)
        out["frac"] = round(tot / (span_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        out["note"] = ("one madicp_tree_build of the bench scan, the last of %d in a rocprofv3 --kernel-trace sub-run; algorithmic bytes are "
                       "an upper bound (every point counted alive on every level); the build is a chain of ~%d dependent launches whose "
                       "working set (2 x 2.9 MB of points, ~10 MB of records) lives in L2 / Infinity Cache: what bounds it is the "
                       "dependent-latency chain of a level (record -> sums -> eigen-solve -> extents -> partition), not HBM — "
                       "profiles/r6_tree_build_pmc_summary.md has the wait fractions" % (BUILD_REPS, len(seg)))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    if args.pmc_child:
        pmc_child(args.pmc_child)
        return
    if args.build_child:
        build_child(args.build_child)
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible — the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # MADICP_BENCH_BACKEND=gloo lets the N>1 code path be EXECUTED on a box with fewer GPUs than ranks: the ranks share devices,
    # every rank's compute stream gets its own slice of the CU mask (MADICP_CU_MASK=rank/world, so that all ranks' round kernels
    # are resident side by side), the `shard` key runs over the library's host-staged transport (RCCL refuses two ranks on one
    # GPU: same kernels and ordering, the all-reduce itself over gloo) and `shard_p2p` over the peer mailboxes.  The line then
    # says "N ranks on one GPU: functional, not a scaling number".
    backend = os.environ.get("MADICP_BENCH_BACKEND", "nccl")
    if backend != "nccl" and world > 1:
        os.environ.setdefault("MADICP_CU_MASK", "%d/%d" % (rank, world))
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    small = torch.device("cuda", device_index) if backend == "nccl" else torch.device("cpu")
    # MADICP_BENCH_FORCE_MULTI=1 (development): run the N > 1 code path with a world of one — the only way to exercise the
    # shard branch's glue (communicator hand-over, batched loop, collectives with one rank) on a 1-GPU box
    force_multi = world == 1 and os.environ.get("MADICP_BENCH_FORCE_MULTI") == "1" and "RANK" in os.environ
    if world > 1 or force_multi:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    from mad_icp_amd import _build, capi, synth

    # the libraries this run measures are built from the sources in this tree, in this run
    if rank == 0:
        _build.build_hip(force=not args.no_rebuild)
        _build.build_host()
        if world == 1:  # (the measurement build of the same sources: the roofline's launch times and the counter sub-runs)
            _build.build_measure(force=not args.no_rebuild)
    if world > 1:
        dist.barrier()

    def fence():
        if world > 1 or force_multi:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1 or force_multi:
            t = torch.tensor([x], dtype=torch.float64, device=small)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    K = args.keyframes
    stream = torch.cuda.Stream()
    # (ranks sharing a device: the library creates the compute stream itself — the one the CU-mask slice applies to)
    ctx = capi.Context(device_index) if (backend != "nccl" and world > 1) else capi.Context(device_index, stream.cuda_stream)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))

    # every rank renders the same seeded scene: 16 keyframe scans + 8 distinct query scans of the same difficulty (scan 0
    # is THE configs[2] query; the others are rendered 5 cm apart with their own noise and their own 0.3 m / 1 deg guess)
    pb = synth.make_problem(K, seed=args.seed, n_queries=1)
    pb["query_scans"], pb["query_gt"], pb["query_guess"] = synth.make_query_streams(K, seed=args.seed, n_streams=N_DISTINCT)
    t_build = time.perf_counter()
    q_trees = [capi.HostTree(s, B_MAX, B_MIN, 3) for s in pb["query_scans"]]
    t_build = (time.perf_counter() - t_build) / len(q_trees)
    leaves = [qt.leaf_means() for qt in q_trees]
    Ls = [qt.num_leaves for qt in q_trees]
    guesses = [capi.pose12(T) for T in pb["query_guess"]]

    out = {}
    if world == 1 and not force_multi:
        out = single_gpu(args, ctx, capi, synth, pb, leaves, Ls, guesses, q_trees, fence, t_build)
    else:
        out = multi_gpu(args, ctx, capi, synth, pb, leaves, Ls, guesses, fence, max_over_ranks, dist, torch, rank, world,
                        small, backend)
    if rank == 0:
        out["built_in_this_run"] = {"forced_rebuild": not args.no_rebuild, "hip_source_sha256": _build.hip_source_hash()}
        print(json.dumps(out), flush=True)
    if world > 1 or force_multi:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def base_line(args, world, value, elapsed, workload, extra_config, warmup_run=None):
    return {
        "metric": "scan registrations/sec (120k pts vs 16 keyframes)",
        "value": round(value, 2),
        "unit": "registrations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "warmup_run": args.warmup if warmup_run is None else warmup_run,  # the untimed steps that really ran
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": dict({"workload": workload, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default")}, **extra_config),
    }


# ---------------------------------------------------------------------------------------------------------
STRESS_K, STRESS_B = 64, 8


def stress_problem(args, ctx, capi, synth, pb, kf_trees, tids):
    """BASELINE configs[4]: the local map extended to 64 keyframes (~300 MB of records: beyond the 256 MB Infinity Cache),
    8 query scans batched in flight.  Keyframes 0..15 are the headline's (make_problem renders keyframe k from k and the
    seed alone); 16..63 and the 8 queries near keyframe 63 are rendered here."""
    scene = synth.Scene(args.seed)
    trees, ids = list(kf_trees), list(tids)
    n_nodes = sum(t.num_nodes for t in trees)
    for k in range(len(trees), STRESS_K):
        T = synth.path_pose(k * 3.0)
        ht = capi.HostTree(synth.render_scan(scene, T, args.seed * 1000 + k), B_MAX, B_MIN, 3)
        ht.transform(T[:3, :3], T[:3, 3])
        trees.append(ht)
        ids.append(ctx.upload(ht))
        n_nodes += ht.num_nodes
    scans, gts, guesses = synth.make_query_streams(STRESS_K, seed=args.seed, n_streams=STRESS_B)
    qts = [capi.HostTree(sc, B_MAX, B_MIN, 3) for sc in scans]
    return dict(trees=trees, tids=ids, n_nodes=n_nodes, moving=[q.leaf_means() for q in qts], Ls=[q.num_leaves for q in qts],
                gts=gts, X0=np.stack([capi.pose12(T) for T in guesses]))


def stress_figures(args, ctx, capi, st, fence, hbm_copy, mctx=None):
    """registrations/s (8 NEW scans in -> 8 results out per step, and the resident loop), icp_round time and its
    bytes for the 64-keyframe / 8-in-flight configuration."""
    B = STRESS_B
    mids = [ctx.moving_upload(lm) for lm in st["moving"]]
    X0 = st["X0"]

    def step(i):
        for s_ in range(B):
            ctx.moving_update(mids[s_], st["moving"][(i + s_) % B])
        order = [(i + s_) % B for s_ in range(B)]
        ctx.icp_register_batch_enqueue(mids, st["tids"], X0[order], PARAMS, N_ITERS)
        return order, ctx.icp_fetch(B)

    for i in range(3):
        step(i)
    fence()
    n, reps = 12, 5  # (every figure below: the median of `reps` repetitions of n steps — a step is ~2 ms, one hiccup is a tenth of a repetition)
    rates = []
    for _ in range(reps):
        t = time.perf_counter()
        for i in range(n):
            order, last = step(i)
        fence()
        rates.append(B * n / (time.perf_counter() - t))
    streamed_sync, sync_spread = float(np.median(rates)), [round(min(rates), 1), round(max(rates), 1)]
    # The same step — 8 NEW scans in, 8 results out — with one batch ahead of the collection, like the headline's loop: the next
    # batch's scans go up on the copy stream (a second set of moving buffers: madicp_moving_update_async) and are enqueued before
    # the previous batch's results — carried out by one kernel behind it, madicp_icp_publish_enqueue — are collected.
    sets = [mids, [ctx.moving_upload(lm) for lm in st["moving"]]]

    def submit(i):
        cur = sets[i % 2]
        order_ = [(i + s_) % B for s_ in range(B)]
        for s_ in range(B):
            ctx.moving_update_async(cur[s_], st["moving"][order_[s_]])
        ctx.icp_register_batch_enqueue(cur, st["tids"], X0[order_], PARAMS, N_ITERS)
        return order_, ctx.icp_publish_enqueue(B)

    def pipelined(count):
        prev, out_ = None, None
        for i in range(count):
            cur = submit(i)
            if prev is not None:
                out_ = (prev[0], ctx.icp_publish_collect(prev[1], B))
            prev = cur
        return prev[0], ctx.icp_publish_collect(prev[1], B)

    pipelined(4)
    fence()
    rates = []
    for _ in range(reps):
        t = time.perf_counter()
        order, last = pipelined(n)
        fence()
        rates.append(B * n / (time.perf_counter() - t))
    streamed, streamed_spread = float(np.median(rates)), [round(min(rates), 1), round(max(rates), 1)]
    for m_ in sets[1]:
        ctx.moving_release(m_)
    terr = max(pose_error(st["gts"][q], capi.pose44(last["X"][s_])) for s_, q in enumerate(order))
    for s_ in range(B):
        ctx.moving_update(mids[s_], st["moving"][s_])
    for _ in range(3):
        ctx.icp_register_batch_enqueue(mids, st["tids"], X0, PARAMS, N_ITERS)
    fence()
    rates = []
    for _ in range(reps):
        t = time.perf_counter()
        for _ in range(n):
            ctx.icp_register_batch_enqueue(mids, st["tids"], X0, PARAMS, N_ITERS)
        fence()
        rates.append(B * n / (time.perf_counter() - t))
    resident, resident_spread = float(np.median(rates)), [round(min(rates), 1), round(max(rates), 1)]
    # launch times: the measurement build's identical kernels (the product library exports no timing aid), its own copy of the map
    m_tids = [mctx.upload(ht) for ht in st["trees"]]
    m_mids = [mctx.moving_upload(lm) for lm in st["moving"]]
    avg_us, final_us, visits, walked = mctx.icp_time_registration(m_mids, m_tids, X0, PARAMS, N_ITERS, reps=10)
    for m_ in m_mids:
        mctx.moving_release(m_)
    for t_ in m_tids:
        mctx.tree_release(t_)
    pairs = float(sum(st["Ls"])) * STRESS_K
    layout = pairs * (32 + 8 + 64) + 16.0 * float(walked.sum()) + 240.0 * 256
    survey = pairs * (24 + 64 + 1) + 64.0 * float(visits.sum()) + 216.0 * B
    out = {
        "workload": "BASELINE configs[4]: %d keyframe MAD-trees (%d nodes, %d MB of node + screening + leaf records: beyond the "
                    "256 MB Infinity Cache), %d query scans batched in flight, 15 GN rounds"
                    % (STRESS_K, st["n_nodes"], (st["n_nodes"] * (64 + 16) + st["n_nodes"] // 2 * 64) >> 20, B),
        "registrations_per_s_new_scans_in_results_out": round(streamed, 1),
        "registrations_per_s_new_scans_in_results_out_synchronous": round(streamed_sync, 1),
        "streamed_note": "a step = 8 NEW scans' leaves in from host memory -> one batched registration -> 8 results out to host memory; "
                         "the first figure keeps one batch ahead of the collection (uploads on the copy stream beside the batch in "
                         "flight, results carried out by a kernel behind it: the headline's loop for batches), the second uploads, "
                         "registers and fetches strictly one after the other (rounds 3-5 reported that one)",
        "registrations_per_s_resident": round(resident, 1),
        "spread": {"repetitions": reps, "steps_each": n, "what": "each registrations/s figure is the median repetition; [min, max] here",
                   "new_scans_in_results_out": streamed_spread, "synchronous": sync_spread, "resident": resident_spread},
        "work_distribution": {"deal_trees": int(ctx.get_option("deal_trees")), "interleave_ranges": int(ctx.get_option("interleave_ranges")),
                              "note": "round 6: the scan's leaves go to the workgroups in groups of 64 dealt over the ranges, the keyframe "
                                      "trees to the XCD pieces in alternating rows — no workgroup draws only the busy stretch of the scan "
                                      "or only the newest keyframes (profiles/r6_range_balance.md)"},
        "max_translation_error_m": round(terr, 5),
        "icp_round_avg_launch_us": round(avg_us, 2), "icp_final_launch_us": round(final_us, 2),
        "pairs_per_launch": int(pairs), "nodes_walked_per_launch": int(walked.sum()),
        "mean_descent_depth": round(float(visits.sum()) / pairs, 3),
        "roofline": {"bound": "latency", "roof": "hbm", "kernel": "icp_round", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                     "achieved": None, "frac": None, "traffic": None,
                     "what_achieved_counts": "HBM bytes per launch from the memory counters / avg launch time (filled in by the "
                                             "PMC sub-runs); layout_bytes and survey_8d_contract are cache-served, not HBM rates",
                     "layout_bytes": {"bytes_per_launch": int(layout), "gbs": round(layout / (avg_us * 1e-6) / 1e9, 1),
                                      "x_hbm_peak": round(layout / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                      "x_measured_copy_rate": round(layout / (avg_us * 1e-6) / 1e9 / max(hbm_copy, 1.0), 4),
                                      "note": "above the box's own copy rate: these bytes are served by L2 / Infinity Cache"},
                     "survey_8d_contract": {"bytes_per_launch": int(survey),
                                            "x_hbm_peak": round(survey / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 3)}},
    }
    for m in mids:
        ctx.moving_release(m)
    return out, avg_us, layout


def traffic_from_counters(m, key, layout_bytes, avg_us):
    """FETCH_SIZE / WRITE_SIZE of the launches filed under `key`, calibrated on the 1 GiB copy of the same pass."""
    copy_bytes = float(1 << 30)
    f_raw, w_raw = m.get((key, "FETCH_SIZE"), 0.0) * 1024, m.get((key, "WRITE_SIZE"), 0.0) * 1024
    cf, cw = m.get(("copy", "FETCH_SIZE"), 0.0) * 1024, m.get(("copy", "WRITE_SIZE"), 0.0) * 1024
    kf = copy_bytes / cf if cf > 0 else 1.0   # the guide says 2.0 for wide streaming reads on gfx950
    kw = copy_bytes / cw if cw > 0 else 1.0
    traffic_stream = f_raw * kf + w_raw * kw
    traffic = traffic_stream
    gather = None
    g_raw = m.get(("gather", "FETCH_SIZE"), 0.0) * 1024
    if g_raw > 0:
        # FETCH_SIZE calibrated in THIS kernel's access pattern: one 16-byte gather per lane, every lane another line, over a
        # region the Infinity Cache cannot hold — the probe's launches touch a known number of distinct lines
        d64, d128 = gather_lines()
        kg64, kg128 = 64.0 * d64 / g_raw, 128.0 * d128 / g_raw
        gather = {"probe": "%d random 16-byte gathers over %d MiB per launch (madicp_debug_gather16)" % (GATHER_N, GATHER_REGION >> 20),
                  "distinct_64B_lines": d64, "distinct_128B_lines": d128, "fetch_size_bytes_raw": int(g_raw),
                  "raw_bytes_per_distinct_64B_line": round(g_raw / d64, 2),
                  "fetch_calibration_64B_lines": round(kg64, 3), "fetch_calibration_128B_lines": round(kg128, 3)}
        traffic = f_raw * kg64 + w_raw * kw
    detail = {
        "fetch_size_bytes_raw": int(f_raw), "write_size_bytes_raw": int(w_raw),
        "fetch_calibration": round(kf, 3), "write_calibration": round(kw, 3),
        "calibration_used": ("gather64: FETCH_SIZE x (64 B x distinct lines / FETCH_SIZE) of the 16-byte gather probe in the same "
                             "pass — this kernel's access pattern; WRITE_SIZE x the streaming copy's factor" if gather else
                             "streaming copy (the gather probe did not run)"),
        "gather_probe": gather,
        "traffic_with_streaming_calibration": int(traffic_stream),
        "traffic_raw_counters": int(f_raw + w_raw),
        "infinity_cache": "FETCH_SIZE counts requests on the L2's memory side: hits in the 256 MB Infinity Cache are included, so "
                          "this is an UPPER bound of the HBM traffic wherever the working set fits that cache",
        "calibrated_on": "a 1 GiB device-to-device copy in the same rocprofv3 pass (known 1 GiB read + 1 GiB written)",
        "launches_averaged": int(m.get((key, "FETCH_SIZE", "n"), 0)),
        "traffic_frac_of_hbm_peak": round(traffic / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
        "traffic_over_layout_bytes": round(traffic / layout_bytes, 4)}
    l2 = None
    req = m.get((key, "TCC_REQ_sum"))
    if req:
        hit, miss = m.get((key, "TCC_HIT_sum"), 0.0), m.get((key, "TCC_MISS_sum"), 0.0)
        l2 = {"requests_per_launch": int(req), "hit_rate": round(hit / max(hit + miss, 1.0), 4),
              "requested_bytes_per_launch_at_128B": int(req * 128),
              "frac_of_l2_peak": round(req * 128 / (avg_us * 1e-6) / 1e9 / L2_PEAK_GBS, 4)}
    return traffic, detail, l2


def set_counter_roofline(roofline, traffic, detail, l2, avg_us, hbm_copy):
    """achieved / frac of a roofline object := the HBM bytes the counters saw per launch / the launch's duration"""
    gbs = traffic / (avg_us * 1e-6) / 1e9
    roofline["traffic"] = int(traffic)
    roofline["achieved"] = round(gbs, 1)
    roofline["frac"] = round(gbs / HBM_PEAK_GBS, 4)
    roofline["frac_of_measured_copy_rate"] = round(gbs / max(hbm_copy, 1.0), 4)
    roofline["traffic_detail"] = detail
    if l2:
        roofline["l2"] = l2


# ---------------------------------------------------------------------------------------------------------
def single_gpu(args, ctx, capi, synth, pb, leaves, Ls, guesses, q_trees, fence, t_build):
    K = args.keyframes
    B = max(1, args.scans)
    kf_trees = []
    tids, n_nodes = [], 0
    for k in range(K):
        T = pb["keyframe_poses"][k]
        ht = capi.HostTree(pb["keyframe_scans"][k], B_MAX, B_MIN, 3)
        ht.transform(T[:3, :3], T[:3, 3])
        kf_trees.append(ht)
        tids.append(ctx.upload(ht))
        n_nodes += ht.num_nodes

    # ---- secondary: device-resident loops (round 1's definition), 1 and 8 scans in flight -------------------
    mids = [ctx.moving_upload(lm) for lm in leaves]
    X0 = np.stack(guesses)

    def resident(nb, steps):
        for _ in range(5):
            ctx.icp_register_batch_enqueue(mids[:nb], tids, X0[:nb], PARAMS, N_ITERS)
        fence()
        t = time.perf_counter()
        for _ in range(steps):
            ctx.icp_register_batch_enqueue(mids[:nb], tids, X0[:nb], PARAMS, N_ITERS)
        fence()
        return nb * steps / (time.perf_counter() - t)

    resident1 = resident(1, min(args.steps, 200))
    resident8 = resident(min(8, len(mids)), max(10, min(args.steps, 200) // 4))
    batch_value = None
    if B > 1:
        batch_value = resident(min(B, len(mids)), max(10, args.steps // B))

    # ---- roofline of the dominant kernel (icp_round), timed live with HIP events on the library's stream ---------
    # The product library exports include/madicp_hip.h and nothing else; the timing aids (n launches of icp_round / a whole
    # registration between two HIP events on the stream they run on) live in the MEASUREMENT build of the same sources
    # (mad_icp_amd/_measure, -DMADICP_MEASURE: identical kernels, same flags), loaded beside it — its own context, its own
    # copy of the map.  `value` and every registrations/s figure of this line come from the product library.
    mc = capi.measure_variant()
    mctx = mc.Context(ctx.device if hasattr(ctx, "device") else 0)
    m_tids = [mctx.upload(ht) for ht in kf_trees]
    m_mids = [mctx.moving_upload(leaves[0])]
    first_us, visits0 = mctx.icp_time_linearize(m_mids, m_tids, X0[:1], PARAMS, 60)
    avg_us, final_us, visits, walked = mctx.icp_time_registration(m_mids, m_tids, X0[:1], PARAMS, N_ITERS, reps=40)
    pairs = Ls[0] * K
    visits_pl, walked_pl = float(visits.sum()), float(walked.sum())
    # bytes one launch must move with THIS data layout: per (leaf, tree) pair the moving leaf (x,y,z,|p|: 32 B), its
    # cached correspondence (8 B) and the matched leaf's record (64 B); per node really walked one 16-byte screening
    # record; one 240-byte partial per workgroup.  
    layout_bytes = pairs * (32 + 8 + 64) + 16.0 * walked_pl + 240.0 * 256
    achieved = layout_bytes / (avg_us * 1e-6) / 1e9
    survey_bytes = pairs * (24 + 64 + 1) + 64.0 * visits_pl + 216.0
    # fixed cost of a round: the same registration on the first 1024 leaves only (16 trees x 16 ranges = 256 workgroups
    # of ONE wave of work each), started at the converged pose so that nothing is walked after round 0
    conv = ctx.icp_register(mids[0], tids, pb["query_guess"][0], PARAMS, N_ITERS, Ls[0])["X"]
    small = mctx.moving_upload(leaves[0][:1024])
    fixed_us, _, _, _ = mctx.icp_time_registration([small], m_tids, conv.reshape(1, 12), PARAMS, N_ITERS, reps=40)
    conv_us, _, _, conv_walked = mctx.icp_time_registration(m_mids, m_tids, conv.reshape(1, 12), PARAMS, N_ITERS, reps=40)
    mctx.moving_release(small)
    hbm_copy = mctx.stream_copy_gbs(1 << 30, 10)

    # `achieved` / `frac` are what the memory counters saw (filled in below, once the PMC sub-runs have run): the only
    # figure of this kernel that is physically an HBM rate.  The bytes the data layout must move and SURVEY 8(d)'s contract
    # bytes are kept as named secondary keys — both are served by LDS / L1 / L2 / Infinity Cache and say nothing about HBM.
    roofline = {
        "bound": "latency", "roof": "hbm", "kernel": "icp_round",
        "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
        "traffic": None,
        "what_achieved_counts": "bytes per launch on the L2's memory side from the counters (FETCH_SIZE calibrated on random 16-byte "
                                "gathers with a known line count — this kernel's access pattern —, WRITE_SIZE on a 1 GiB copy; "
                                "rocprofv3 --pmc sub-runs of this script) / avg launch time; Infinity-Cache hits are counted, so "
                                "an upper bound of the HBM rate.  The kernel is not bandwidth-bound: see latency_budget",
        "limiter": "latency, not bandwidth: a chain of dependent steps per round at 3 waves/SIMD — see latency_budget",
        "avg_launch_us": round(avg_us, 2), "first_round_launch_us": round(first_us, 2), "final_launch_us": round(final_us, 2),
        "launch_times_from": "HIP events around captured launch sequences on the library's own stream, through the timing aids of "
                             "the MEASUREMENT build of the same sources (mad_icp_amd/_measure, -DMADICP_MEASURE: identical kernels and "
                             "flags, built in this run) — the product library exports include/madicp_hip.h only; `value` and every "
                             "registrations/s figure are the product library's",
        "rounds": N_ITERS, "pairs_per_launch": pairs,
        "nodes_walked_per_launch": int(walked_pl), "nodes_visited_per_launch_reference_count": int(visits_pl),
        "mean_descent_depth": round(visits_pl / pairs, 3),
        "layout_bytes": {"bytes_per_launch": int(layout_bytes), "gbs": round(achieved, 1),
                         "x_hbm_peak": round(achieved / HBM_PEAK_GBS, 4),
                         "note": "bytes one launch must move with this data layout (32+8+64 B per (leaf,tree) pair, 16 B per "
                                 "node really walked, 240 B per workgroup partial) / avg launch time; served by LDS, L1, L2 and "
                                 "the Infinity Cache (the 16-keyframe map is %d MB), NOT an HBM rate"
                                 % ((n_nodes * (64 + 16) + n_nodes // 2 * 64) >> 20)},
        "survey_8d_contract": {"bytes_per_launch": int(survey_bytes),
                               "gbs": round(survey_bytes / (avg_us * 1e-6) / 1e9, 1),
                               "x_hbm_peak": round(survey_bytes / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 3),
                               "note": "SURVEY 8(d)'s contract figure 24+64d+64+1 per pair prices every visit of the REFERENCE's "
                                       "descent at a 64-byte node from HBM; this kernel reads 16-byte records, mostly from "
                                       "LDS/L2, and provably skips unchanged descents, so it is not a rate this kernel moves "
                                       "(x_hbm_peak > 1 says exactly that)"},
        "latency_budget": {
            "fixed_us_per_round": round(fixed_us, 2),
            "work_us_per_round": round(avg_us - fixed_us, 2),
            "converged_round_us": round(conv_us, 2),
            "how": "fixed = avg icp_round launch of the same registration on 1024 leaves (256 workgroups x one wave) from the "
                   "converged pose: dispatch + join of 256 partials + 6x6 LDLT/expSO3 + pose broadcast + reduction; work = "
                   "avg launch - fixed; converged = avg launch of the full scan from the converged pose (no descent after round 0)"},
        "measured_hbm_copy_gbs": round(hbm_copy, 1),
    }

    # ---- BASELINE configs[4] (64 keyframes, 8 scans in flight): the configuration whose map exceeds the Infinity Cache ----
    stress = None
    st = None
    if K == 16 and B == 1 and os.environ.get("MADICP_BENCH_STRESS", "1") != "0":
        try:
            st = stress_problem(args, ctx, capi, synth, pb, kf_trees, tids)
            stress, stress_us, stress_layout = stress_figures(args, ctx, capi, st, fence, hbm_copy, mctx)
        except Exception as e:  # noqa: BLE001 — a secondary figure never takes the bench line down
            stress, st = {"error": str(e)[:200]}, None

    pmc_m = None
    if args.pmc == "auto":
        q_map_pmc = (pb["query_scans"][0] @ pb["query_gt"][0][:3, :3].T) + pb["query_gt"][0][:3, 3]
        m = measure_traffic(pb, kf_trees, leaves[0], X0[0], stress=st, nn_queries=q_map_pmc)
        pmc_m = m
        if "error" in m:
            roofline["traffic_error"] = m["error"]
        else:
            traffic, detail, l2 = traffic_from_counters(m, "round", layout_bytes, avg_us)
            set_counter_roofline(roofline, traffic, detail, l2, avg_us, hbm_copy)
            if st is not None and ("round2", "FETCH_SIZE") in m:
                traffic2, detail2, l2b = traffic_from_counters(m, "round2", stress_layout, stress_us)
                set_counter_roofline(stress["roofline"], traffic2, detail2, l2b, stress_us, hbm_copy)
    if st is not None:
        for t_ in st["tids"][K:]:
            ctx.tree_release(t_)
        st = None

    # ---- nn_descend (the pymadtree path: mad_tree_wrapper.h:48-67), 120k queries ------------------------------
    q_map = (pb["query_scans"][0] @ pb["query_gt"][0][:3, :3].T) + pb["query_gt"][0][:3, 3]
    us_a, depth_a = mctx.nn_time_descend(m_tids[-1], q_map, 30)
    dense = capi.HostTree(pb["keyframe_scans"][-1], 1e-5, B_MIN, 3)
    Tk = pb["keyframe_poses"][-1]
    dense.transform(Tk[:3, :3], Tk[:3, 3])
    dense_id = mctx.upload(dense)
    us_b, depth_b = mctx.nn_time_descend(dense_id, q_map, 30)
    nq = q_map.shape[0]
    nn = {"queries": nq,
          "keyframe_tree_b_max_0.2": {"leaves": kf_trees[-1].num_leaves, "us_per_launch": round(us_a, 2),
                                      "mqueries_per_s": round(nq / us_a, 1), "mean_depth": round(depth_a / nq, 2)},
          "dense_tree_b_max_1e-5": {"leaves": dense.num_leaves, "us_per_launch": round(us_b, 2),
                                    "mqueries_per_s": round(nq / us_b, 1), "mean_depth": round(depth_b / nq, 2),
                                    "layout_gbs": round(nq * (24 + 16.0 * depth_b / nq + 64 + 16) / (us_b * 1e-6) / 1e9, 1)},
          "note": "queries and outputs resident; one launch = all queries against one tree (pymadtree searchCloud)"}
    if pmc_m and ("nn", "FETCH_SIZE") in pmc_m:
        # counter traffic of nn_descend against the keyframe tree (the PMC sub-runs launch it behind the registrations)
        lay = nq * (24 + 16.0 * depth_a / nq + 64 + 16)  # query + a 16-byte record per level + the leaf record + outputs
        tr, det, l2n = traffic_from_counters(pmc_m, "nn", lay, us_a)
        nn["keyframe_tree_b_max_0.2"]["roofline"] = {
            "bound": "latency", "roof": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": int(tr),
            "achieved": round(tr / (us_a * 1e-6) / 1e9, 1), "frac": round(tr / (us_a * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
            "layout_bytes": int(lay), "traffic_detail": det, "l2": l2n}
    mctx.tree_release(dense_id)
    mctx.close()

    # ---- the device front-end: the query scan's MAD-tree built on the device ----------------
    front = None
    try:
        scan0 = pb["query_scans"][0]
        cid = ctx.cloud_upload(scan0)
        for _ in range(3):
            t_, _nl = ctx.tree_build(cid, B_MAX, B_MIN)
            ctx.tree_release(t_)
        ctx.synchronize()
        tb, tub = [], []
        for _ in range(15):
            t1 = time.perf_counter()
            t_, dev_leaves = ctx.tree_build(cid, B_MAX, B_MIN)
            ctx.synchronize()
            tb.append(time.perf_counter() - t1)
            ctx.tree_release(t_)
        ctx.cloud_release(cid)
        for _ in range(8):
            t1 = time.perf_counter()
            c2 = ctx.cloud_upload(scan0)
            t_, _nl = ctx.tree_build(c2, B_MAX, B_MIN)
            ctx.synchronize()
            tub.append(time.perf_counter() - t1)
            ctx.tree_release(t_)
            ctx.cloud_release(c2)
        st = ctx.tree_build_stats()
        front = {"device_tree_build_ms_per_scan": round(float(np.median(tb)) * 1e3, 3),
                 "device_upload_plus_build_ms_per_scan": round(float(np.median(tub)) * 1e3, 3),
                 "host_tree_build_ms_per_scan": round(t_build * 1e3, 2),
                 "points": int(scan0.shape[0]), "device_leaves": int(dev_leaves), "host_leaves": int(Ls[0]),
                 "levels": int(st["max_level"]),
                 "roofline": builder_roofline(scan0, dev_leaves, st["max_level"]) if args.pmc == "auto" else None,
                 "note": "madicp_tree_build on a resident cloud (wall time incl. its one host synchronisation); upload = "
                         "pageable host memory -> pinned staging -> HBM; the host figure is the product's CPU builder on this "
                         "box's cores (bit-identical to the oracle's), which the streamed headline does NOT include either"}
    except Exception as e:  # noqa: BLE001 — a secondary figure never takes the bench line down
        front = {"error": str(e)[:200]}

    # ---- end to end: Pipeline.compute on a short synthetic drive ("an end-to-end Pipeline::compute figure")
    pipe = None
    try:
        from mad_icp_amd import _build as _b

        _b.build_pybind()
        from mad_icp.src.pybind import pypeline as pm

        scene = synth.Scene(0)
        drive = [synth.render_scan(scene, synth.path_pose(1.0 * i), 100 + i) for i in range(16)]
        threads = min(os.cpu_count() or 1, 16)
        pipe = {"frames": len(drive), "points_per_scan": int(drive[0].shape[0]), "host_threads": threads}
        # deskewed datasets (most of the reference's configurations): a sensor's noise leaves no two azimuths equal; the
        # synthetic scans share 64 points per azimuth column, so the deskew keys run on a copy with 1e-7 m of jitter
        jitter = np.random.default_rng(0)
        drive_j = [sc + jitter.normal(scale=1e-7, size=sc.shape) for sc in drive]
        # ("default" is not run first: the first drive of a process also pays for the pool's and the builder's first allocations)
        for key, dev, ahead, dsk in (("host_path", False, 0, False), ("default", None, 0, False), ("host_path_lookahead", False, 2, False),
                                     ("device_front_end", True, 0, False), ("device_front_end_lookahead", True, 1, False),
                                     ("host_path_deskew", False, 1, True), ("device_front_end_deskew", True, 0, True),
                                     ("default_deskew", None, 0, True), ("default_f32_origin", None, 0, False)):
            pl = pm.Pipeline(10.0, dsk, B_MAX, RHO_KER, 0.8, B_MIN, B_RATIO, K, threads, False)
            if dev is not None:
                pl.setDeviceFrontEnd(dev)
            ts = []
            scans_ = drive_j if dsk else drive
            if key.endswith("_f32_origin"):  # what a sensor driver / a KITTI .bin delivers: float32 coordinates, converted by the caller
                scans_ = [sc.astype(np.float32).astype(np.float64) for sc in drive]
            for d in range(ahead):
                pl.prefetch(scans_[d])
            for i, sc in enumerate(scans_):
                t1 = time.perf_counter()
                if ahead and i + ahead < len(scans_):
                    pl.prefetch(scans_[i + ahead])  # the trees of the next scans are built while this one is registered
                pl.compute(0.1 * i, sc)
                ts.append(time.perf_counter() - t1)
            if ahead:
                ts = ts[:-ahead]  # (the last frames have nothing left to look ahead to)
            gt = np.linalg.inv(synth.path_pose(0.0)) @ synth.path_pose(1.0 * (len(drive) - 1))
            # total time over frames (with a look-ahead the per-frame series is bimodal: its median says nothing)
            pipe[key] = {"ms_per_frame": round(float(np.mean(ts[2:])) * 1e3, 3),
                         "frames_per_s": round(1.0 / float(np.mean(ts[2:])), 1),
                         "ms_per_frame_median": round(float(np.median(ts[2:])) * 1e3, 3),
                         "tree_ms": round(pl.lastBuildMs(), 3), "registration_ms": round(pl.lastIcpMs(), 3),
                         "end_translation_error_m": round(float(np.linalg.norm(np.asarray(pl.currentPose())[:3, 3] - gt[:3, 3])), 4)}
        # the same drive in PLAIN processes (no framework, one context), with the runtime's default hardware queues and with eight
        # (eight: the next scan's upload and construction really run beside the registration, and both lose — profiles/
        # r5_u_lookahead_canary.md)
        plain = {}
        for label, qs in (("hw_queues_default", None), ("hw_queues_8", "8")):
            env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
            if qs:
                env["GPU_MAX_HW_QUEUES"] = qs
            env["LOOKAHEAD_ONLY"] = "device"
            try:
                r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "lookahead_probe.py"), "24"],
                                   env=env, capture_output=True, text=True, timeout=120)
                ms = [float(l.split("mean")[1].split("ms")[0]) for l in r.stdout.splitlines() if l.startswith("device front-end")]
                plain[label] = {"no_lookahead_ms_per_frame": ms[0], "lookahead_ms_per_frame": ms[1]}
            except Exception as e:  # noqa: BLE001
                plain[label] = {"error": str(e)[:120]}
        pipe["lookahead_in_a_plain_process"] = plain
        pipe["default_is_device_front_end"] = bool(pm.Pipeline(10.0, False, B_MAX, RHO_KER, 0.8, B_MIN, B_RATIO, K, threads, False).deviceFrontEnd())
        pipe["note"] = ("Pipeline.compute(stamp, cloud) per frame, cloud in host memory: default = what an UNMODIFIED caller gets "
                        "(no setDeviceFrontEnd call, MAD_ICP_GPU_BUILD unset: the device front-end — round 5 for deskew = false, round 6 "
                        "for deskewed datasets too: default_deskew is the unmodified caller of a `deskew : True` configuration); "
                        "default_f32_origin = the default again on the same scans rounded to float32 and handed over as doubles — "
                        "what every LiDAR driver, KITTI .bin or PointCloud2 delivers: such a cloud crosses PCIe as floats and is widened "
                        "on the device, the same doubles bit for bit (option upload_f32); "
                        "host_path = setDeviceFrontEnd(False) / MAD_ICP_GPU_BUILD=0 (host tree builder, "
                        "bit-identical to the oracle's, + upload); host_path_lookahead = the same with prefetch(scan i + 2) issued "
                        "before compute(scan i): the frame PERIOD of a caller that reads ahead (a dataset), same poses bit for bit; "
                        "device_front_end = setDeviceFrontEnd(True): upload, MAD-tree build and registration on the GPU; "
                        "device_front_end_lookahead = the same with prefetch(scan i + 1) before compute(scan i): the next scan's "
                        "construction is submitted on the library's build stream behind this scan's registration, same poses bit for bit; "
                        "lookahead_in_a_plain_process has the same drive in plain processes under the runtime's default hardware queues "
                        "and under GPU_MAX_HW_QUEUES=8 (where the look-ahead does not pay); "
                        "host_path_deskew / device_front_end_deskew = deskew on (scans with distinct azimuths; the synthetic scans are "
                        "instantaneous, so compensating them moves them: timing keys, not accuracy keys), the host one with the "
                        "azimuth order computed ahead by prefetch(scan i + 1)")
    except Exception as e:  # noqa: BLE001
        pipe = {"error": str(e)[:200]}

    # ---- the headline: streamed registrations, a different scan every step (measured after the secondary figures:
    # the W warm-up steps below are then the only thing between a busy device and the timed region) --------------
    # (at least 40 untimed steps — 9 ms — whatever W says: every stream slot's graph, the kernel-by-kernel route a busy stream
    # takes, the pinned staging of all four slots have then been used, and the device clocks are where a long run keeps them:
    # the driver's --steps 20 --warmup 5 then reports 4 420 against 4 540 for 400 steps — what is left is the one registration
    # of pipeline drain in twenty — instead of 4 150)
    # the host is only one submission ahead of the device here, so a generational garbage collection over the
    # interpreter's (torch-sized) heap would stall the device for tens of milliseconds: keep it out of the timed region,
    # like timeit does — and collect BEFORE the warm-up, so that nothing but the fence separates the warm-up's last
    # registration from the timed region's first (tens of milliseconds of idle device in between showed as a slow start of a
    # 20-step run)
    gc.collect()
    gc.disable()
    streamed_loop(ctx, capi, leaves, guesses, tids, max(args.warmup, MIN_WARMUP))
    fence()
    results = []
    stamps = [] if os.environ.get("MADICP_BENCH_DEBUG") else None

    def timed(n_steps, res=None, st=None):
        """EXACTLY n_steps registrations between two fences (device idle on both sides)"""
        t0 = time.perf_counter()
        streamed_loop(ctx, capi, leaves, guesses, tids, n_steps, res, st)
        fence()
        return time.perf_counter() - t0

    # The timed region — exactly K steps between two fences — is repeated, back to back, and the line carries the spread:
    # `value` / `ms_per_step` are the MEDIAN repetition, p10 / p90 beside it.  (One repetition of the driver's K = 20 is 4.5 ms:
    # a single sample of that says little; rounds 4-5 reported 4 492 .. 4 547 from it against 4 665 .. 4 684 from 400 steps.)
    reps = 5 if args.steps >= 400 else int(min(25, max(5, 2000 // max(1, args.steps))))
    laps = []
    for r_ in range(reps):
        laps.append(timed(args.steps, results if r_ == 0 else None, stamps if r_ == 0 else None))
    rates = np.array([args.steps / t for t in laps])
    elapsed = float(np.median(laps))
    value = args.steps / elapsed
    spread = {"repetitions": reps, "steps_each": args.steps, "median": round(float(np.median(rates)), 1),
              "p10": round(float(np.percentile(rates, 10)), 1), "p90": round(float(np.percentile(rates, 90)), 1),
              "min": round(float(rates.min()), 1), "max": round(float(rates.max()), 1),
              "note": "registrations/s of each repetition of the timed region (exactly --steps steps between two fences, "
                      "repeated back to back); `value` is the median repetition"}
    # ... and the long-run interval beside it, so that a short run can be read against it
    if args.steps != 400:
        long_rates = np.array([400 / timed(400) for _ in range(5)])
        spread["steps400"] = {"repetitions": 5, "median": round(float(np.median(long_rates)), 1),
                              "p10": round(float(np.percentile(long_rates, 10)), 1), "p90": round(float(np.percentile(long_rates, 90)), 1)}
        lo, hi = float(long_rates.min()), float(long_rates.max())
        spread["stability"] = ("this run's %d-step median %.0f is %s the 400-step interval [%.0f, %.0f] measured in the same process "
                               "(a short run ends with one registration of pipeline drain in %d)" % (
                                   args.steps, value, "INSIDE" if lo <= value <= hi else ("%.1f %% below" % (100 * (lo - value) / lo) if value < lo
                                                                                          else "%.1f %% above" % (100 * (value - hi) / hi)),
                                   lo, hi, args.steps))
    else:
        spread["stability"] = "400-step repetitions: p10 .. p90 = %.0f .. %.0f (%.1f %% of the median)" % (
            spread["p10"], spread["p90"], 100 * (spread["p90"] - spread["p10"]) / spread["median"])
    gc.enable()
    if stamps:
        d = np.diff(np.array(stamps)) * 1e6
        print("step us: first50 %.1f mid %.1f last50 %.1f max %.1f n>400us %d ; slowest %s ; first %s" % (
            d[:50].mean(), d[len(d) // 2 - 50:len(d) // 2 + 50].mean(), d[-50:].mean(), d.max(), int((d > 400).sum()),
            [(int(i), int(d[i])) for i in np.argsort(d)[-6:]], np.round(d[:20]).astype(int).tolist()), file=sys.stderr)
    # what was timed is a real registration: every collected result sits at its scan's ground-truth pose
    terr = max(pose_error(pb["query_gt"][q], r["T"]) for q, r in results)
    matched = [int(r["n_matched"]) for _, r in results[:N_DISTINCT]]

    # PCIe-inclusive SINGLE registration with nothing overlapped (submit, then collect at once): the latency a caller
    # sees for one scan — reported, never `value`
    ts = []
    for i in range(10):
        t1 = time.perf_counter()
        tk = ctx.stream_submit(leaves[i % N_DISTINCT], tids, guesses[i % N_DISTINCT], PARAMS, N_ITERS)
        ctx.stream_collect(tk, Ls[i % N_DISTINCT])
        ts.append(time.perf_counter() - t1)
    lat_ms = float(np.median(ts) * 1e3)

    cpu = None
    if args.cpu_baseline == "auto":
        cpu = cpu_baseline(pb, K, args.cpu_seconds)

    out = base_line(args, 1, value, elapsed,
                    "BASELINE configs[2]: %d-pt KITTI-shaped synthetic scans vs %d keyframe MAD-trees, %d GN rounds, b_max=0.2 "
                    "b_min=0.1 rho_ker=0.1 b_ratio=0.02; a step = new scan's leaves in (host) -> registration -> X,H,b,flags out "
                    "(host), %d distinct scans cycled, upload/read-back streamed inside the timed region"
                    % (len(pb["query_scans"][0]), K, N_ITERS, N_DISTINCT),
                    {"keyframes": K, "scans_in_flight": 1, "moving_leaves": Ls, "map_nodes_this_rank": n_nodes,
                     "parallelism": "single GPU"}, warmup_run=max(args.warmup, MIN_WARMUP))
    out.update({
        "nn_mqueries_per_s": round(value * float(np.mean(Ls)) * K * N_ITERS / 1e6, 1),
        "nn_mqueries_note": "(leaf, tree, round) pairs resolved per second = value x L x K x 15; pairs whose descent is "
                            "provably unchanged are resolved without a walk — nodes really walked are in roofline",
        "max_translation_error_m": round(terr, 5),
        "matched_leaves_first_scans": matched,
        "spread": spread,
        "resident_loop": {"registrations_per_s": round(resident1, 1), "scans8_in_flight_registrations_per_s": round(resident8, 1),
                          "batch_registrations_per_s": None if batch_value is None else round(batch_value, 1),
                          "note": "round-1 definition: same pre-uploaded scans re-registered, nothing read back"},
        "single_registration_latency_ms": round(lat_ms, 3),
        "host_tree_build_ms_per_scan": round(t_build * 1e3, 2),
        "stress_k64_b8": stress,
        "front_end": front,
        "pipeline_end_to_end": pipe,
        "nn_descend": nn,
        "roofline": roofline,
        "cpu_baseline": cpu,
    })
    return out


# ---------------------------------------------------------------------------------------------------------
def multi_gpu(args, ctx, capi, synth, pb, leaves, Ls, guesses, fence, max_over_ranks, dist, torch, rank, world, small, backend):
    K = args.keyframes
    out_extra = {}

    def batched(step_scans, tids, steps, warmup, mids):
        """steps x {upload B new scans (every rank: the moving leaves are replicated), register, read B results back}"""
        B = len(mids)

        def one(i):
            qs = [(i * B + s) % N_DISTINCT for s in range(B)]
            for s, q in enumerate(qs):
                ctx.moving_update(mids[s], leaves[q])
            ctx.icp_register_batch_enqueue(mids, tids, np.stack([guesses[q] for q in qs]), PARAMS, N_ITERS)
            return qs, ctx.icp_fetch(B)

        if B == 1:
            # ONE scan in flight: the streamed form of the same registration (madicp_stream_submit / _collect, one submission
            # ahead of the collection — the headline's loop), so that upload and read-back overlap the device work here too
            # instead of standing between two registrations; every rank submits the same sequence
            streamed_loop(ctx, capi, leaves, guesses, tids, warmup)
            fence()
            gc.collect()
            gc.disable()
            res = []
            t0 = time.perf_counter()
            streamed_loop(ctx, capi, leaves, guesses, tids, steps, res)
            fence()
            dt = time.perf_counter() - t0
            gc.enable()
            q, r = res[-1]
            return max_over_ranks(dt), ([q], {"X": r["X"].reshape(1, 12)})
        for i in range(warmup):
            one(i)
        fence()
        gc.collect()
        gc.disable()  # (as in the streamed loop: a generational collection of the torch-sized heap costs tens of ms)
        t0 = time.perf_counter()
        last = None
        for i in range(steps):
            last = one(i)
        fence()
        dt = time.perf_counter() - t0
        gc.enable()
        return max_over_ranks(dt), last

    native = backend == "nccl"  # RCCL inside the library; otherwise its host-staged transport over torch.distributed
    shard_ok = True
    value = elapsed = None
    n_local = 0
    shard_error = None
    if shard_ok and args.mode in ("auto", "shard"):
        # a failure of the sharded path (communicator set-up, a collective that does not complete: MADICP_ERR_COMM after the
        # library's bounded wait) must not cost the run its line: every rank catches, the ranks agree, the line then reports
        # the replica figure and says why
        try:
            from mad_icp_amd import sharded as _shard_plan
            my = _shard_plan.shard_keyframes(K, world, rank)  # (rows of `world` keyframes in alternating direction)
            tids, n_nodes = upload_map(ctx, capi, pb, my)
            n_local = len(tids)
            from mad_icp_amd import sharded as _sh
            if native:
                uid = torch.zeros(128, dtype=torch.uint8, device=small)
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(capi.Context.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(uid, 0)
                ctx.comm_init(bytes(uid.cpu().numpy().tobytes()), world, rank)
            else:
                _sh.init_host_comm(ctx)
            if os.environ.get("MADICP_BENCH_FAIL_SHARD") == "1":  # (development: exercises the fall-back below)
                raise RuntimeError("injected failure of the sharded path")
            if os.environ.get("MADICP_COMM_GRAPH") == "1":
                ctx.set_option("comm_graph", 1)
            B = args.scans if args.scans > 0 else world
            mids = [ctx.moving_upload(leaves[s % N_DISTINCT]) for s in range(B)]
            # strong scaling first (ONE scan in flight over all the GPUs): a secondary figure, and the communicator's first
            # few hundred collectives (channel set-up) are out of the way before the headline is timed
            e1, _ = batched(1, tids, max(20, args.steps // 4), 5, mids[:1])
            out_extra["shard_one_scan"] = {"registrations_per_s": round(max(20, args.steps // 4) / e1, 1), "scaling": "strong",
                                           "note": "one scan in flight, 16 trees over %d %s, %d all-reduces of 240 B" % (
                                               world, "GPUs" if native else "ranks on one GPU", N_ITERS)}
            elapsed, last = batched(B, tids, args.steps, args.warmup, mids)
            value = args.steps * B / elapsed
            terr = max(pose_error(pb["query_gt"][q], capi.pose44(last[1]["X"][s])) for s, q in enumerate(last[0]))
            shard_note = (
                "keyframe sharding was developed on ONE GPU (RCCL with one rank; two ranks through a host-staged transport): this line "
                "is the first multi-GPU measurement of it.  A round is ~14 us of device work per scan and every sharded round adds "
                "icp_reduce (4.5 us) and one %d-byte RCCL all-reduce, so the shard keys are bound by all-reduce latency, not by "
                "xGMI bandwidth; with four or more scans in flight the batch runs as two halves on two streams so that one half's "
                "collective is in flight under the other half's round (measured on one GPU with a 15 us stand-in collective: -14 %% "
                "per registration at 8 scans); `replica` (no collective) is the key that scales with the GPU count"
                % (240 * (args.scans if args.scans > 0 else world)))
            out_extra["shard_note"] = shard_note if native else (
                "%d ranks on ONE GPU (each rank's compute stream on its own slice of the CU mask), (H,b) all-reduced over the "
                "library's host-staged transport: this exercises every line of the world = %d sharded path — partition, zero-tree "
                "ranks, launch sequence, collectives per round, flags — and says NOTHING about scaling: the ranks share one chip "
                "and every round makes a host round trip" % (world, world))
            out_extra["shard_mode"] = {
                "transport": "RCCL (ncclAllReduce enqueued by the library between its kernels)" if native else
                             "host-staged (madicp_comm_init_host): the RCCL path's kernels and ordering, the all-reduce itself over gloo "
                             "on pinned host memory — one host round trip per round",
                "sequence": "icp_round -> icp_reduce -> all-reduce(30 f64 per scan) per round; matched flags OR-ed once",
                "shard_split": ("two half-batches on two streams (one half's all-reduce under the other half's round)" if B >= 4
                                else "off (fewer than four scans in flight)"),
                "shard_tail": "off (the round kernel folding its own rows is measured slower: profiles/r4_c_shard_probe.md)"}
            # per BATCH of B scans: one sum all-reduce per round (two of B/2 scans each with the batch split in halves) and one
            # max-all-reduce of the matched flags per scan
            n_sum = N_ITERS * (2 if B >= 4 else 1)
            out_extra["all_reduces_per_batch"] = {"sum_f64": n_sum, "max_u8_matched_flags": B, "scans_per_batch": B,
                                                  "per_registration": round((n_sum + B) / B, 2)}
            out_extra["all_reduce_payload_bytes_per_round"] = 240 * B
            out_extra["max_translation_error_m"] = round(terr, 5)
            # ---- the same sharding with the per-round join over peer-mapped mailboxes instead of a collective (option
            # "shard_p2p"): side by side with the RCCL figure above, same batch, same trees
            p2p_err = None
            try:
                _sh.attach_peer_mailboxes(ctx, allow_coarse=not native)  # (coarse-grained memory only between ranks of ONE device)
            except Exception as e:  # noqa: BLE001
                p2p_err = "%s: %s" % (type(e).__name__, str(e)[:200])
            bad = torch.tensor([1.0 if p2p_err else 0.0], device=small)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if bad.item() > 0:
                out_extra["shard_p2p"] = {"error": p2p_err or "a peer could not map the mailboxes"}
            else:
                try:
                    ctx.set_option("shard_p2p", 1)
                    ctx.set_option("comm_timeout_ms", 10000)  # (a mailbox that never fills must not hold the line for a minute per wait)
                    e1p, _ = batched(1, tids, max(20, args.steps // 4), 5, mids[:1])
                    ep, lastp = batched(B, tids, args.steps, args.warmup, mids)
                    terr_p = max(pose_error(pb["query_gt"][q], capi.pose44(lastp[1]["X"][s])) for s, q in enumerate(lastp[0]))
                    out_extra["shard_p2p"] = {
                        "registrations_per_s": round(args.steps * B / ep, 1), "scans_in_flight": B,
                        "one_scan_registrations_per_s": round(max(20, args.steps // 4) / e1p, 1),
                        "vs_shard_rccl": round((args.steps * B / ep) / value, 3),
                        "max_translation_error_m": round(terr_p, 5),
                        "mailbox_memory": "fine-grained" if ctx.get_option("p2p_fine_grained") else "coarse-grained (ranks of one device)",
                        "sequence": "icp_round only: workgroup 0 stores the rank's 30 sums per scan into every peer's hipIpc-mapped "
                                    "mailbox, every workgroup polls its own mailbox in the next round's prologue and adds the rows in "
                                    "rank order; icp_final exchanges the matched flags the same way (32 per tagged word): no icp_reduce, "
                                    "no collective, no host step — the single-GPU launch sequence, captured in a hipGraph, results out "
                                    "through the side stream",
                        "note": "developed and tested with up to eight ranks on ONE GPU (an eighth of the CU mask each): %s" %
                                ("this is a world of one — nothing crossed xGMI" if world == 1 else
                                 "this line is its first measurement over xGMI" if native else
                                 "%d ranks on one GPU — nothing crossed xGMI" % world)}
                except Exception as e:  # noqa: BLE001 — (a rank that fails here leaves its peers to their bounded waits)
                    out_extra["shard_p2p"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
                finally:
                    ctx.set_option("shard_p2p", 0)
                    ctx.set_option("comm_timeout_ms", 60000)
            for m in mids:
                ctx.moving_release(m)
            for t in tids:
                ctx.tree_release(t)
            ctx.comm_destroy()
        except Exception as e:  # noqa: BLE001
            shard_error = "%s: %s" % (type(e).__name__, str(e)[:300])
        failed = torch.tensor([1.0 if shard_error else 0.0], device=small)
        dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        if failed.item() > 0:
            value = elapsed = None
            out_extra.pop("shard_one_scan", None)
            out_extra["shard_error"] = shard_error or "the sharded path failed on another rank"
            try:
                ctx.comm_destroy()
            except Exception:  # noqa: BLE001
                pass

    # replicas: the whole map on every GPU, every rank streams its own scans, no collective
    tids, n_nodes = upload_map(ctx, capi, pb, list(range(K)))
    r_leaves = list(leaves)
    r_guess = list(guesses)
    rot = rank % N_DISTINCT  # every rank starts its cycle at a different scan
    r_leaves = r_leaves[rot:] + r_leaves[:rot]
    r_guess = r_guess[rot:] + r_guess[:rot]
    streamed_loop(ctx, capi, r_leaves, r_guess, tids, args.warmup)
    fence()
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    streamed_loop(ctx, capi, r_leaves, r_guess, tids, args.steps)
    fence()
    r_elapsed = max_over_ranks(time.perf_counter() - t0)
    gc.enable()
    replica = args.steps * world / r_elapsed
    out_extra["replica"] = {"registrations_per_s": round(replica, 1), "scaling": "weak",
                            "note": "every rank holds all %d trees and streams its own scans; no data-path collective" % K}
    if value is None or args.mode == "replica":
        value, elapsed = replica, r_elapsed
        why = "the sharded path failed, see shard_error" if "shard_error" in out_extra else "--mode replica"
        workload = "replicas of BASELINE configs[2] (no shard figure: %s)" % why
        par = "replicas"
    else:
        B = args.scans if args.scans > 0 else world
        if native:
            workload = ("BASELINE configs[3]: %d keyframe MAD-trees sharded %d per GPU over %d x MI355X, %d scans in flight (one per "
                        "GPU), RCCL all-reduce of (H,b) over xGMI after every one of the %d GN rounds; a step = %d new scans' leaves in "
                        "-> registration -> %d results out" % (K, n_local, world, B, N_ITERS, B, B))
        else:
            workload = ("BASELINE configs[3] EXECUTED with %d ranks on ONE GPU — functional, not a scaling number: %d keyframe "
                        "MAD-trees sharded %d per rank, every rank's compute stream on its own slice of the CU mask "
                        "(MADICP_CU_MASK=rank/%d), %d scans in flight, all-reduce of (H,b) over the library's host-staged transport "
                        "(gloo) after every one of the %d GN rounds; a step = %d new scans' leaves in -> registration -> %d results "
                        "out" % (world, K, n_local, world, B, N_ITERS, B, B))
        par = "keyframes sharded %d/rank + all-reduce(H,b) per round" % n_local
    out = base_line(args, world, value, elapsed, workload,
                    {"keyframes": K, "scans_in_flight": (args.scans if args.scans > 0 else world), "moving_leaves": Ls,
                     "parallelism": par})
    out.update(out_extra)
    return out


def cpu_baseline(pb, K, budget_s):
    """The restated reference OpenMP path (oracle/) on this box's host cores: same keyframes, same scan,
    same 15 rounds; `omp parallel for` over keyframes exactly like pipeline.cpp:180-183."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cores = os.cpu_count() or 1
    threads = min(cores, 16)
    trees = []
    for s, T in zip(pb["keyframe_scans"], pb["keyframe_poses"]):
        tr = O.Tree(s, B_MAX, B_MIN, 3)
        tr.transform(T[:3, :3], T[:3, 3])
        trees.append(tr)
    q = O.Tree(pb["query_scans"][0], B_MAX, B_MIN, 3)
    T0 = pb["query_guess"][0]
    first = O.icp_register(q, trees, T0, N_ITERS, B_MAX, RHO_KER, B_RATIO, num_threads=threads)["ms"] * 1e-3
    n = int(max(3, min(40, budget_s / max(first, 1e-3))))
    ms = [O.icp_register(q, trees, T0, N_ITERS, B_MAX, RHO_KER, B_RATIO, num_threads=threads)["ms"] for _ in range(n)]
    med = float(np.median(ms)) * 1e-3
    out = {"value": round(1.0 / med, 3), "unit": "registrations/s", "cores": threads, "host_cores": cores,
           "p10": round(1e3 / float(np.percentile(ms, 90)), 3), "p90": round(1e3 / float(np.percentile(ms, 10)), 3),
           "kind": "port",
           "sample": "%d registrations of the same workload (K=%d, L=%d, 15 rounds), median; GN loop only "
                     "(the region the reference stopwatches, pipeline.cpp:171-192)" % (n, K, q.num_leaves),
           "ms_per_registration": round(med * 1e3, 2)}
    out["ref_tus"] = cpu_baseline_ref_tus(pb, K, threads, max(6.0, budget_s / 2))
    return out


REF_TUS_SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import oracle_lib as O
z = np.load(sys.argv[2])
K, threads, budget = int(z["K"]), int(z["threads"]), float(z["budget"])
trees = []
for k in range(K):
    tr = O.Tree(z["scan%d" % k], 0.2, 0.1, 3)
    T = z["pose%d" % k]
    tr.transform(T[:3, :3], T[:3, 3])
    trees.append(tr)
q = O.Tree(z["query"], 0.2, 0.1, 3)
first = O.icp_register(q, trees, z["guess"], 15, 0.2, 0.1, 0.02, num_threads=threads)["ms"] * 1e-3
n = int(max(3, min(40, budget / max(first, 1e-3))))
ms = [O.icp_register(q, trees, z["guess"], 15, 0.2, 0.1, 0.02, num_threads=threads)["ms"] for _ in range(n)]
print(json.dumps({"ms": float(np.median(ms)), "ms_p10": float(np.percentile(ms, 10)), "ms_p90": float(np.percentile(ms, 90)), "n": n,
                  "leaves": int(q.num_leaves)}))
"""


def cpu_baseline_ref_tus(pb, K, threads, budget_s):
    """The same registrations through the REFERENCE'S OWN translation units — tools/mad_tree.cpp, odometry/mad_icp.cpp ...
    compiled from /root/reference against the Eigen stand-in (oracle/_ref/libmad_ref_standin.so, built where the reference
    is and shipped as a file; oracle/build_ref_standin.sh): the reference's control flow, loops and data structures
    (heap-allocated pointer tree, omp parallel for over keyframes) with the oracle's arithmetic primitives in place of
    Eigen's.  Runs in a subprocess (the ctypes binding holds one library per process).  None where the file is missing."""
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "_ref", "libmad_ref_standin.so")
    if not os.path.exists(so):
        return None
    tmp = tempfile.mkdtemp(prefix="madicp_ref_", dir="/tmp")
    try:
        arrays = dict(K=K, threads=threads, budget=budget_s, query=pb["query_scans"][0], guess=pb["query_guess"][0])
        for k in range(K):
            arrays["scan%d" % k] = pb["keyframe_scans"][k]
            arrays["pose%d" % k] = pb["keyframe_poses"][k]
        npz = os.path.join(tmp, "pb.npz")
        np.savez(npz, **arrays)
        root = os.path.dirname(os.path.abspath(__file__))
        r = subprocess.run([sys.executable, "-c", REF_TUS_SCRIPT, root, npz], env=dict(os.environ, MADICP_ORACLE_SO=so),
                           capture_output=True, text=True, timeout=300)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": round(1e3 / d["ms"], 3), "unit": "registrations/s", "cores": threads,
                "p10": round(1e3 / d["ms_p90"], 3), "p90": round(1e3 / d["ms_p10"], 3),
                "kind": "reference translation units + Eigen stand-in",
                "sample": "%d registrations of the same workload, median; GN loop only" % d["n"],
                "ms_per_registration": round(d["ms"], 2),
                "note": "the reference's own mad_tree.cpp / mad_icp.cpp compiled from /root/reference with its flags against "
                        "oracle/eigen_standin (no Eigen in this image): its loops and data structures, the oracle's arithmetic"}
    except Exception as e:  # noqa: BLE001 — a secondary figure
        return {"error": str(e)[:200]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
