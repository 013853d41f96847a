(python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3)
for i in 1 2; do python bench.py --cpu-baseline off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_us'])"; done
python bench.py --scans 8 --cpu-baseline off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_us'])"
