#!/bin/bash
# round-4 measurement pass behind profiles/<tag>_*.  usage (GPU box): tools/measure_r4.sh <tag> [what...]
#   what: tests bench trace build stamps big   (default: all)
set -u
TAG=$1; shift
WHAT="${*:-tests bench trace build stamps big}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { case " $WHAT " in *" $1 "*) return 0;; esac; return 1; }
if has tests; then
  timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > $OUT/pytest_gpu.log; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2
fi
if has build; then
  timeout 300 python tools/order_probe.py 3 > $OUT/order_probe.log 2>&1; grep -E "scan120k|kf0|line100|rand1 |rand9" $OUT/order_probe.log
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/btrace -o t -- python tools/build_probe.py 5 > $OUT/build_probe.log 2> $OUT/btrace.err
  python tools/build_trace.py $(find $OUT/btrace -name "t_kernel_trace.csv" | head -1) > $OUT/tree_build_trace.txt
  rm -rf $OUT/btrace
  tail -3 $OUT/build_probe.log | head -1; grep "^#" $OUT/tree_build_trace.txt
fi
if has big; then
  timeout 600 python tools/big_probe.py > $OUT/big_probe.log 2>&1; tail -12 $OUT/big_probe.log
fi
if has bench; then
  timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err
  timeout 600 python bench.py --steps 20 --warmup 5 --no-rebuild > $OUT/bench_steps20.json 2>> $OUT/bench.err
  python tools/show_bench.py $OUT/bench_n1.json
fi
if has trace; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/round_trace.py run 200 > /dev/null 2> $OUT/trace.err
  python tools/round_trace.py split $(find $OUT/trace -name "t_kernel_trace.csv" | head -1) > $OUT/round_trace.md
  cp $(find $OUT/trace -name "t_kernel_stats.csv" | head -1) $OUT/round_trace_kernel_stats.csv
  rm -rf $OUT/trace
  cat $OUT/round_trace.md
fi
if has stamps; then
  timeout 600 python tools/stamps.py 16 > $OUT/phase_stamps.md 2>&1; tail -17 $OUT/phase_stamps.md
fi
