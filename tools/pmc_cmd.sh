#!/bin/bash
# SQ / cache counter passes over any command (one counter group per pass; --pmc only with --kernel-trace, as the GPU pool
# requires).   usage: tools/pmc_cmd.sh <outdir> <command...>     e.g.  tools/pmc_cmd.sh gpurun_out/p python tools/round_trace.py run 100
set -u
OUT=$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -o p -- "$@" > "$OUT/pass$i.log" 2> "$OUT/pass$i.err" || echo "pass $i failed: $grp"
done
python tools/pmc_summary.py "$OUT" > "$OUT/summary.md" 2>&1
grep -A 26 "icp_round" "$OUT/summary.md" | head -60
