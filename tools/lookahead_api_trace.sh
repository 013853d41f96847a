#!/bin/bash
# HIP-API trace beside the kernel trace of the look-ahead drive (tools/overlap_trace.py run), in the fast process configuration
# (runtime-default hardware queues) and a slow one (GPU_MAX_HW_QUEUES=8): which host call waits while the device idles.
# usage (GPU box): tools/lookahead_api_trace.sh OUTDIR [frames]; then python tools/lookahead_api_report.py OUTDIR/<cfg>
O=${1:-gpurun_out/la_api}; N=${2:-30}
mkdir -p $O; cd /tmp 2>/dev/null; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in q4 q8; do
  if [ $cfg = q4 ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=8; fi
  timeout 300 rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d $R/$O/$cfg -o t -- \
    python $R/tools/overlap_trace.py run $N > $R/$O/$cfg.log 2>&1 || echo "rocprofv3 failed for $cfg (see $O/$cfg.log)"
  find $R/$O/$cfg -name '*.csv' | head
done
