#!/bin/bash
# (GPU box) the measurement set of the round's last kernel change (k_big; the other translation units' device code is unchanged,
# profiles/r04_device_code.sha256): GPU parity suite + smoke, the driver-style bench line, rocprofv3 kernel stats and PMC passes
# of the 4096 / 8192-point configurations.  Everything lands in gpurun_out/r04_big/.
D=gpurun_out/r04_big
mkdir -p $D
timeout 900 python -m pytest tests -m gpu -x -q > $D/pytest.txt 2>&1
tail -3 $D/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench.json 2> $D/bench.err
timeout 300 bash tools/prof_cfg.sh 11,13 200 r04_big/prof > $D/prof_summary.txt 2>&1
timeout 300 bash tools/pmc_cfg.sh 11 r04_big > $D/pmc_11_stdout.txt 2>&1
timeout 300 bash tools/pmc_cfg.sh 13 r04_big > $D/pmc_13_stdout.txt 2>&1
timeout 300 python tools/bench_configs.py --only 11,13 > $D/other_configs.jsonl 2> $D/other_configs.err
timeout 300 python tools/bench_configs.py --only 11,13 --packets 16384 --steps 100 >> $D/other_configs.jsonl 2>> $D/other_configs.err
python3 -c "
import json
d=json.loads(open('$D/bench.json').read().strip().splitlines()[-1])
print('launch us', d['roofline']['launch_ms']*1e3, 'frac', d['roofline']['frac'], 'value M/s', d['value']/1e6)
for k, e in d['other_configs'].items(): print(k, e['us_per_launch'], e['frac'], e['kernels'])
e=d['end_to_end']; print('e2e', e['value'], 'dev', e.get('device_entropy',{}).get('value'), 'sharder', (e.get('sharder') or {}).get('value'))
"
tail -8 $D/prof_summary.txt
cut -c1-230 $D/other_configs.jsonl
