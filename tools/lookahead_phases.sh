#!/bin/bash
# tools/lookahead_phases.py under the process configurations of tools/lookahead_matrix.sh (GPU box): tools/lookahead_phases.sh > out.md
for q in default 8; do
  for pub in 1 0; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    MADICP_PUBLISH_SIDE=$pub timeout 200 python tools/lookahead_phases.py ${1:-40} 2>&1 | grep -v "^\[" ; echo
  done
done
