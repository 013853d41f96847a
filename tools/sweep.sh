#!/bin/bash
# registrations/s over (keyframes, scans in flight) — the parity-test configurations of BASELINE.json, timed
for kb in "1 1" "1 8" "16 1" "16 2" "16 4" "16 8" "16 16" "16 32" "64 1" "64 8"; do
  set -- $kb
  python bench.py --keyframes $1 --scans $2 --steps 60 --warmup 10 --cpu-baseline off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('K=%3d B=%2d  %9.1f reg/s  %8.4f ms/step  %7.1f G NN/s  avg launch %7.2f us  err %.4f' % ($1, $2, d['value'], d['ms_per_step'], d['nn_mqueries_per_s']/1e3, r['avg_launch_us'], d['final_translation_error_m']))"
done
