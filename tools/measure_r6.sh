#!/bin/bash
# round 6's measurement pass for the device tree builder (profiles/r6_<tag>_*):
#   tools/measure_r6.sh <tag> [probe trace stamps pmc]
#     probe   madicp_tree_build on the bench scan, host clock (tools/build_probe.py), and the unmodified-caller frame
#     trace   rocprofv3 --kernel-trace of the same, kernel-by-kernel timeline of ONE build (tools/build_trace.py)
#     stamps  in-kernel wall-clock stamps per level (a second library with -DMADICP_TB_STAMPS, out of the tree)
#     pmc     SQ / cache / memory counter passes per tb_* kernel (one counter group per pass, --kernel-trace only)
set -u
TAG=$1; shift
WHAT="${*:-probe trace stamps pmc}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { case " $WHAT " in *" $1 "*) return 0;; esac; return 1; }
if has probe; then
  timeout 300 python tools/build_probe.py 40 > $OUT/build_probe.log 2>&1; tail -2 $OUT/build_probe.log | head -1
fi
if has trace; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/btrace -o t -- python tools/build_probe.py 5 > $OUT/build_probe_traced.log 2> $OUT/btrace.err
  python tools/build_trace.py $(find $OUT/btrace -name "t_kernel_trace.csv" | head -1) > $OUT/tree_build_trace.txt
  cp $(find $OUT/btrace -name "t_kernel_stats.csv" | head -1) $OUT/tree_build_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/btrace
  grep "^#" $OUT/tree_build_trace.txt
fi
if has stamps; then
  mkdir -p /tmp/tbstamps
  MADICP_NATIVE_DIR=/tmp/tbstamps MADICP_EXTRA_DEFINES=-DMADICP_TB_STAMPS python -c "from mad_icp_amd import _build; _build.build_hip()" > $OUT/stamps_build.log 2>&1
  MADICP_HIP_LIB=/tmp/tbstamps/libmadicp_hip.so timeout 300 python tools/tb_stamps.py > $OUT/tb_stamps.txt 2>&1; head -14 $OUT/tb_stamps.txt
fi
if has pmc; then
  mkdir -p $OUT/pmc
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
             "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc/pass$i" -o p -- python tools/build_probe.py 5 > "$OUT/pmc/pass$i.log" 2> "$OUT/pmc/pass$i.err" || echo "pass $i failed: $grp"
  done
  python tools/pmc_summary.py "$OUT/pmc" > "$OUT/tree_build_pmc_summary.md" 2>&1
  rm -rf $OUT/pmc
  grep -c "^## " $OUT/tree_build_pmc_summary.md
fi
# ---- the round's closing pass (everything profiles/r6_* quotes): tools/measure_r6.sh <tag> final
if has final; then
  F='^\[mad_icp_amd build\]\|^madicp: MADICP_CU_MASK\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|\[Gloo\]\|Loading frame\|Time for'
  timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "$F" > $OUT/pytest_gpu_full.log; grep -E "passed|failed" $OUT/pytest_gpu_full.log | tail -2
  grep -E "^\[(configs3|lost peer|deskew = true|default Pipeline|pipeline,)|python launcher|bin_runner|passed|failed" $OUT/pytest_gpu_full.log > $OUT/pytest_gpu_summary.log
  timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err; python tools/show_bench.py $OUT/bench_n1.json | head -3
  timeout 600 python bench.py --steps 20 --warmup 5 --no-rebuild > $OUT/bench_steps20.json 2>> $OUT/bench.err
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/round_trace.py run 200 > /dev/null 2> $OUT/trace.err
  python tools/round_trace.py split $(find $OUT/trace -name "t_kernel_trace.csv" | head -1) > $OUT/round_trace.md
  cp $(find $OUT/trace -name "t_kernel_stats.csv" | head -1) $OUT/round_trace_kernel_stats.csv; rm -rf $OUT/trace; head -12 $OUT/round_trace.md
  K64_ONLY8=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/k64trace -o t -- python tools/k64_probe.py > $OUT/k64_probe.log 2>&1
  python tools/k64_trace.py $(find $OUT/k64trace -name "t_kernel_trace.csv" | head -1) > $OUT/k64_round_trace.md; rm -rf $OUT/k64trace; tail -3 $OUT/k64_round_trace.md
  timeout 300 tools/shard_world1.sh $OUT/shard_world1.json; timeout 300 tools/shard_world1.sh $OUT/shard_world1_scans8.json --scans 8
  for N in 8 4; do
    MADICP_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $N --steps 40 --warmup 5 --no-rebuild 2> $OUT/bench_gloo_$N.err | grep '^{"metric"' > $OUT/bench_gloo_$N.json
    python -c "
import json,sys; d=json.load(open('$OUT/bench_gloo_$N.json')); print($N, 'ranks on one GPU:', d['value'], d['shard_p2p'].get('registrations_per_s'), d['replica']['registrations_per_s'])"
  done
  timeout 300 python tools/tree_ready_probe.py 40 2>&1 | tail -1 > $OUT/tree_ready.log; cat $OUT/tree_ready.log
  timeout 600 python tools/builder_soak.py 150 2>&1 | tail -2 > $OUT/builder_soak.log; cat $OUT/builder_soak.log
fi
