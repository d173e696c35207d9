#!/bin/bash
# (GPU box) ring tests + end-to-end sweep through lw_ring + bench line with end_to_end
D=gpurun_out/r02_c2
mkdir -p $D
timeout 600 python -m pytest tests/test_gpu_ring.py -x -q > $D/pytest_ring.log 2>&1; echo "rc=$?" >> $D/pytest_ring.log
tail -n 4 $D/pytest_ring.log
for t in 16 32 64 128; do
  timeout 120 python tools/e2e.py --batches 48 --threads $t 2>&1 | tail -1
done
timeout 120 python tools/e2e.py --batches 48 --threads 32 --slots 2 2>&1 | tail -1
timeout 120 python tools/e2e.py --batches 48 --threads 32 --slots 4 2>&1 | tail -1
timeout 120 python tools/e2e.py --batches 48 --threads 32 --device-vq 2>&1 | tail -1
timeout 120 python tools/e2e.py --batches 48 --threads 64 --device-vq 2>&1 | tail -1
timeout 120 python tools/e2e.py --batches 48 --threads 32 --callers 2 2>&1 | tail -1
timeout 120 python tools/e2e.py --batches 48 --threads 32 --callers 2 --device-vq 2>&1 | tail -1
timeout 120 python tools/e2e.py --batches 48 --threads 16 --callers 4 --device-vq 2>&1 | tail -1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver.json 2> $D/bench_driver.err
python3 -c "
import json
d=json.loads(open('$D/bench_driver.json').read().strip().splitlines()[-1])
print('launch', d['roofline']['launch_ms']*1e3, 'frac', d['roofline']['frac'], 'value', d['value']/1e6)
print(json.dumps(d['end_to_end'])[:900])
"
