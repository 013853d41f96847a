#!/bin/bash
# round-3 measurement pass behind profiles/<tag>_*: bench line (400 steps), rocprofv3 kernel trace + stats of bench.py, the
# per-registration trace split, phase stamps.  usage (GPU box): tools/measure_r3.sh <tag>
set -u
TAG=$1
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python bench.py --steps 30 --warmup 5 --cpu-baseline off --pmc off > $OUT/bench_prof.json 2> $OUT/prof.err
cp $(find $OUT/prof -name "t_kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/round_trace.py run 200 > /dev/null 2> $OUT/trace.err
python tools/round_trace.py split $(find $OUT/trace -name "t_kernel_trace.csv" | head -1) > $OUT/round_trace.md
python tools/stamps.py 16 > $OUT/stamps.md 2>&1
rm -rf $OUT/prof $OUT/trace
python tools/show_bench.py $OUT/bench_n1.json; cat $OUT/round_trace.md; head -12 $OUT/kernel_stats.csv; tail -17 $OUT/stamps.md
