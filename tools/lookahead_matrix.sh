#!/bin/bash
# The look-ahead frame of the default Pipeline (device front-end, prefetch(i + 1) before compute(i)) under the process
# configurations an embedder may have: hardware queues 4 (runtime default) / 8, side publishing on / off, construction stream on
# all CUs / on a subset.  usage (GPU box): tools/lookahead_matrix.sh > out.md
for q in default 8; do
  for pub in 1 0; do
    for cus in ${LOOKAHEAD_CUS:-0 64 128}; do
      if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
      if [ $cus = 0 ]; then unset MADICP_BUILD_CUS; else export MADICP_BUILD_CUS=$cus; fi
      r=$(MADICP_PUBLISH_SIDE=$pub LOOKAHEAD_ONLY=device timeout 120 python tools/lookahead_probe.py 40 2>/dev/null | grep "device front-end" | sed 's/(median.*//' | tr '\n' ';')
      echo "| GPU_MAX_HW_QUEUES=$q | publish_side=$pub | build CUs=${cus/#0/all} | build_after_registration=${MADICP_BUILD_AFTER_REG:-default} | $r |"
    done
  done
done
