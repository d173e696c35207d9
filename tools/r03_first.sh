#!/bin/bash
# (GPU box) round 3, first call: GPU parity suite on the cleaned-up library, the driver-style bench line, the other configs with
# their oracle check.  Everything lands in gpurun_out/r03a/.
D=gpurun_out/r03a
mkdir -p $D
timeout 900 python -m pytest tests -m gpu -x -q > $D/pytest.txt 2>&1
tail -5 $D/pytest.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench.json 2> $D/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end > $D/bench_long.json 2>> $D/bench.err
timeout 600 python tools/bench_configs.py --only 3,4,5,6,7,10,11,12 > $D/other_configs.jsonl 2> $D/other_configs.err
python3 -c "
import json
for f in ('bench.json','bench_long.json'):
    d=json.loads(open('$D/'+f).read().strip().splitlines()[-1])
    print(f,'launch us', d['roofline']['launch_ms']*1e3, 'frac', d['roofline']['frac'], 'value M/s', d['value']/1e6, d['config']['parity'])
    if d.get('end_to_end'): print(' e2e', d['end_to_end'].get('value'), 'dev_entropy', d['end_to_end'].get('device_entropy',{}).get('value'))
    if d.get('cpu_baseline'): print(' cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('synthesis_only',{}).get('value'))
"
cut -c1-330 $D/other_configs.jsonl
tail -3 $D/bench.err $D/other_configs.err
