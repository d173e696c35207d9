#!/bin/bash
# (GPU box) round-2 first call: parity at the quoted shapes, full GPU suite, smoke, the driver's bench command, the
# end-to-end rate with the round-1-final host stage, and design probes for k_long (single-channel units, 2 rounds).
D=gpurun_out/r02_c1
mkdir -p $D
timeout 600 python -m pytest tests/test_gpu_shapes.py -x -q > $D/pytest_shapes.log 2>&1; echo "rc=$?" >> $D/pytest_shapes.log
timeout 900 python -m pytest tests -m gpu -x -q > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $D/bench_driver.json 2> $D/bench_driver.err
timeout 300 python bench.py --no-cpu-baseline > $D/bench_4000.json 2> $D/bench_4000.err
timeout 200 python tools/bench_configs.py --steps 800 --only 3,5,6,7,8,9 > $D/configs.jsonl 2> $D/configs.err
for t in 32 64 128; do
  timeout 120 python tools/e2e.py --batches 48 --threads $t > $D/e2e_t$t.txt 2>&1
done
timeout 120 python tools/e2e.py --batches 48 --threads 64 --device-vq > $D/e2e_vq.txt 2>&1
timeout 120 python tools/e2e.py --batches 48 --threads 64 --callers 2 > $D/e2e_c2.txt 2>&1
timeout 120 python tools/e2e.py --batches 48 --threads 64 --callers 2 --device-vq > $D/e2e_c2_vq.txt 2>&1
# solo critical path of one wave (pair): only wave 0 of every workgroup works
LW_EXTRA_FLAGS="-DLW_EXP_ACTIVE_WAVES=1" python lewton_amd/build.py --force > $D/build_solo.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --steps 2000 --warmup 200 > $D/bench_solo1.json 2> $D/bench_solo1.err
LW_EXTRA_FLAGS="-DLW_EXP_ACTIVE_WAVES=4" python lewton_amd/build.py --force > $D/build_solo.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --steps 2000 --warmup 200 > $D/bench_solo4.json 2> $D/bench_solo4.err
nproc > $D/host.txt; lscpu | head -20 >> $D/host.txt
tail -n 3 $D/pytest_shapes.log; tail -n 3 $D/pytest.log; tail -1 $D/smoke.log; head -c 1500 $D/bench_driver.json; echo; cat $D/configs.jsonl | cut -c1-220; grep -h "end-to-end" $D/e2e_*.txt
for f in $D/bench_4000.json $D/bench_solo1.json $D/bench_solo4.json; do python3 -c "
import json,sys
for l in open('$f'):
    if l.startswith('{'):
        d=json.loads(l); print('$f', d['roofline']['launch_ms']*1e3, 'us', d['config'].get('parity'))
"; done
