#!/bin/bash
# Is the slow look-ahead frame the HOST's allocate + copy + free of a cloud (the by-value argument of prefetch() / compute())?
# tools/lookahead_phases.py with canary copies, in the configurations of tools/lookahead_matrix.sh, with glibc's defaults and with
# malloc kept from mmap / munmap / trimming (no fresh pages per cloud).  usage (GPU box): tools/lookahead_canary.sh > out.md
run() { timeout 200 python tools/lookahead_phases.py ${N:-40} 2>&1 | grep -v "^\[" | grep -v "^| 0 \|^|---\|^| look"; }
for q in default 8; do
  for pub in 1 0; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    export MADICP_PUBLISH_SIDE=$pub
    echo "### plain"; LOOKAHEAD_CANARY=0 run
    echo "### with canary"; LOOKAHEAD_CANARY=1 run
    echo "### malloc without mmap / trim"; LOOKAHEAD_CANARY=1 MALLOC_MMAP_THRESHOLD_=1073741824 MALLOC_TRIM_THRESHOLD_=4294967296 MALLOC_TOP_PAD_=67108864 run
    echo
  done
done
