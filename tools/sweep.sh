#!/bin/bash
# registrations/s over (keyframes, scans in flight) — the parity-test configurations of BASELINE.json, timed.
# value = streamed, one scan in flight; batch = resident loop with B scans in flight (resident_loop.batch_registrations_per_s)
for kb in "1 1" "1 8" "16 1" "16 8" "16 16" "64 1" "64 8"; do
  set -- $kb
  python bench.py --no-rebuild --keyframes $1 --scans $2 --steps 100 --warmup 10 --cpu-baseline off --pmc off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; rl=d['resident_loop']
print('K=%3d B=%2d  streamed(1 scan) %8.1f reg/s  resident(1) %8.1f  resident(B) %s  icp_round avg %6.2f us  err %.4f m' % ($1, $2, d['value'], rl['registrations_per_s'], rl['batch_registrations_per_s'], r['avg_launch_us'], d['max_translation_error_m']))"
done
