#!/bin/bash
# round-5 measurement pass.  usage (GPU box): tools/measure_r5.sh <tag> [what...]
#   what: kernel (parity tests of the round kernel)  oracle (front-end vs oracle tests)  small (small-cloud listing)
#         k64 (configs[4] A/B + per-round trace)  pmc (configs[4] SQ / cache counters)  trace (headline per-round kernel trace)
#         world1 (bench.py's N > 1 branch with a world of one: shard over RCCL and over the mailboxes)  bench  tests (whole GPU suite)
set -u
TAG=$1; shift
WHAT="${*:-kernel oracle small k64 bench}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { case " $WHAT " in *" $1 "*) return 0;; esac; return 1; }
if has kernel; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 > $OUT/pytest_kernel.log; tail -5 $OUT/pytest_kernel.log
fi
if has oracle; then
  timeout 900 python -m pytest tests/test_gpu_frontend_oracle.py tests/test_boundary.py -m gpu -q -k "not test_default_pipeline_builds" 2>&1 > $OUT/pytest_oracle.log; grep -E "passed|failed" $OUT/pytest_oracle.log | tail -3
fi
if has small; then
  timeout 300 python tools/small_cloud_diff.py > $OUT/small_cloud_diff.log 2>&1; grep -c "=== cloud" $OUT/small_cloud_diff.log
fi
if has k64; then
  MADICP_AB="leaf_major=0;leaf_major=2048" timeout 600 python tools/k64_probe.py > $OUT/k64_ab.log 2>&1; grep "K=64" $OUT/k64_ab.log
  K64_ONLY8=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/k64trace -o t -- python tools/k64_probe.py > $OUT/k64_probe.log 2>&1
  python tools/k64_trace.py $(find $OUT/k64trace -name "t_kernel_trace.csv" | head -1) > $OUT/k64_round_trace.md; rm -rf $OUT/k64trace; cat $OUT/k64_round_trace.md
fi
if has bench; then
  timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err; python tools/show_bench.py $OUT/bench_n1.json | head -40
fi
if has pmc; then
  timeout 600 tools/k64_pmc.sh $OUT/k64_pmc > $OUT/k64_pmc.log 2>&1; cp $OUT/k64_pmc/summary.md $OUT/k64_pmc_summary.md 2>/dev/null; rm -rf $OUT/k64_pmc; grep -A 24 "icp_round" $OUT/k64_pmc_summary.md | head -30
fi
if has trace; then
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/round_trace.py run 200 > /dev/null 2> $OUT/trace.err
  python tools/round_trace.py split $(find $OUT/trace -name "t_kernel_trace.csv" | head -1) > $OUT/round_trace.md
  cp $(find $OUT/trace -name "t_kernel_stats.csv" | head -1) $OUT/round_trace_kernel_stats.csv
  rm -rf $OUT/trace
  cat $OUT/round_trace.md
fi
if has world1; then
  timeout 300 tools/shard_world1.sh $OUT/shard_world1.json; timeout 300 tools/shard_world1.sh $OUT/shard_world1_scans8.json --scans 8
fi
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 > $OUT/pytest_gpu.log; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -3
fi
