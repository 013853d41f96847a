#!/bin/bash
# bench.py's N > 1 branch with a world of ONE (the only way to run its glue — communicator hand-over, sharded batches, RCCL with one
# rank, the peer mailboxes — on a 1-GPU box).  usage: tools/shard_world1.sh <out.json> [--scans B]
OUT=$1; shift
MADICP_BENCH_FORCE_MULTI=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --gpus 1 --steps 60 --warmup 10 --no-rebuild "$@" > $OUT 2> ${OUT%.json}.err
python - "$OUT" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "scaling")})
for k in ("shard_one_scan", "shard_p2p", "all_reduces_per_batch", "shard_error", "replica"):
    if k in d: print(k, d[k])
PY
