#!/bin/bash
# (GPU box) quick round trip: k_long parity subset + bench (4000 steps and the driver's 20) [+ extra build flags in $1]
D=gpurun_out/${TAG:-quick}
mkdir -p $D
if [ -n "$1" ]; then LW_EXTRA_FLAGS="$1" python lewton_amd/build.py --force > $D/build.log 2>&1; fi
timeout 600 python -m pytest tests/test_gpu_shapes.py tests/test_gpu_parity.py -x -q -k "dense or long_block or one_packet or full_size or read_audio_packet or batch_many" > $D/pytest.log 2>&1; echo "rc=$?" >> $D/pytest.log
timeout 300 python bench.py --no-cpu-baseline > $D/bench_4000.json 2> $D/bench_4000.err
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $D/bench_20.json 2> $D/bench_20.err
tail -n 2 $D/pytest.log
for f in $D/bench_4000.json $D/bench_20.json; do python3 -c "
import json
for l in open('$f'):
    if l.startswith('{'):
        d=json.loads(l); print('$f', round(d['roofline']['launch_ms']*1e3,3), 'us frac', round(d['roofline']['frac'],4), d['config'].get('parity'))
"; done
