#!/bin/bash
# (GPU box) sharder on rings, tap-hash fixture on the HIP path, the whole GPU suite, driver-style bench line with the sharder leg
D=gpurun_out/r03d
mkdir -p $D
timeout 1200 python -m pytest tests -m gpu -x -q > $D/pytest.txt 2>&1
tail -6 $D/pytest.txt
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench.json 2> $D/bench.err
python3 -c "
import json
d=json.loads(open('$D/bench.json').read().strip().splitlines()[-1])
print('launch us', d['roofline']['launch_ms']*1e3, 'frac', d['roofline']['frac'], 'value M/s', d['value']/1e6, d['config']['parity'])
e=d['end_to_end']
print(' e2e host', e.get('value'), 'dev_entropy', e.get('device_entropy',{}).get('value'), 'large', e.get('device_entropy',{}).get('large_batches',{}).get('value'))
print(' sharder', json.dumps(e.get('sharder'))[:600])
"
tail -3 $D/bench.err
