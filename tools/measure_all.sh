#!/bin/bash
# the full measurement pass behind profiles/<tag>_*: tests, bench (B=1 with the CPU baseline, B=8, the 64-keyframe stress case),
# kernel trace, PMC passes.  usage (GPU box): tools/measure_all.sh <tag>
set -u
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -5) > $OUT/tests.log 2>&1
python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err
python bench.py --scans 8 --cpu-baseline off > $OUT/bench_n1_scans8.json 2>> $OUT/bench.err
python bench.py --keyframes 64 --scans 8 --steps 30 --warmup 5 --cpu-baseline off > $OUT/bench_n1_k64_scans8.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python bench.py --steps 30 --warmup 5 --cpu-baseline off > $OUT/bench_prof.json 2> $OUT/prof.err
python tools/trace_split.py $OUT/prof/t_kernel_trace.csv 35 > $OUT/kernel_trace_summary.md
tools/pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1
python tools/stamps.py 16 > $OUT/stamps.md 2>&1
cat $OUT/tests.log; cat $OUT/bench_n1.json; cat $OUT/kernel_trace_summary.md; head -3 $OUT/prof/t_kernel_stats.csv
python - <<PY
import json
for f in ("$OUT/bench_n1_scans8.json","$OUT/bench_n1_k64_scans8.json"):
    d=json.load(open(f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"])
PY
grep -E "FETCH_SIZE|WRITE_SIZE" $OUT/pmc/summary.md | head -6
