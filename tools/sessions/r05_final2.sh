#!/bin/bash
# (GPU box) round 5, closing session on the final tree (device code = profiles/r05_device_code.sha256, unchanged since r05_final;
# host side: copier thread for tenants, lw_batch_state_bytes): the whole GPU suite, smoke, the driver's bench line, the
# differential campaigns, the end-to-end rates.  -> gpurun_out/r05_final2/
D=gpurun_out/r05_final2; mkdir -p $D
( time timeout 1200 python -m pytest tests -m gpu -q ) > $D/pytest.txt 2>&1; tail -5 $D/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.txt 2>&1; tail -1 $D/smoke.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $D/bench.json 2> $D/bench.err; tail -4 $D/bench.err
timeout 300 python tools/fuzz_gpu_mixed.py --rounds 100 --seed 81 2>&1 | tail -1 > $D/fuzz_gpu_mixed.txt
timeout 300 python tools/fuzz_gpu_mixed.py --rounds 100 --seed 82 --mid 2>&1 | tail -1 >> $D/fuzz_gpu_mixed.txt
timeout 300 python tools/fuzz_gpu_mixed.py --rounds 60 --seed 83 --big 2>&1 | tail -1 >> $D/fuzz_gpu_mixed.txt
{ for S in stereo surround51_bookless; do timeout 200 python tools/fuzz_gpu_entropy.py --packets 40000 --setup $S 2>&1 | tail -1 | cut -c1-200; done; } > $D/fuzz_gpu_entropy.txt 2>&1
{ echo "e2e_sharder: $(timeout 100 python tools/e2e_sharder.py 2>&1 | tail -1 | cut -c1-420)"
  echo "single ring, device entropy: $(timeout 100 python tools/e2e.py --batches 300 --device-entropy 2>&1 | tail -1 | cut -c1-300)"
  for i in 1 2 3; do timeout 30 python tools/probe/sharder_probe.py 2 400 4096 2>&1 | head -2; done; } > $D/end_to_end_sharder.txt 2>&1
python3 -c "
import json
d=json.loads([l for l in open('$D/bench.json') if l.startswith('{')][-1])
print('launch us', d['roofline']['launch_ms']*1e3, 'frac', d['roofline']['frac'], 'value M/s', d['value']/1e6, 'state', d['roofline'].get('state_bytes_per_launch'))
e=d['end_to_end']; print('e2e', e['value'], 'dev', e.get('device_entropy',{}).get('value'), 'sharder', (e.get('sharder') or {}).get('value'))
for k,v in d['other_configs'].items(): print('  ', k, v.get('us_per_launch'), v.get('frac'), v.get('frac_incl_state'), v.get('kernels'), (v.get('parity') or v.get('error'))[:40])
"
cat $D/fuzz_gpu_mixed.txt $D/fuzz_gpu_entropy.txt $D/end_to_end_sharder.txt | cut -c1-300
