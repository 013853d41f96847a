(python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -5)
python bench.py --cpu-baseline off > gpurun_out/b1.json 2>gpurun_out/b1.err; python bench.py --scans 8 --cpu-baseline off > gpurun_out/b8.json 2>>gpurun_out/b1.err
python -c "
import json
for f in ('gpurun_out/b1.json','gpurun_out/b8.json'):
    d=json.load(open(f)); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_us'], r['first_round_launch_us'], r['final_launch_us'])"
python tools/stamps.py 16 2>&1 | tail -17
