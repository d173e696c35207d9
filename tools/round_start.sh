#!/bin/bash
# (GPU box) first call of a round: re-validate the shipped build and re-measure everything that moved on the host side
# since the last GPU run -- parity tests, the bench line, the end-to-end rate over host thread counts and with the entropy
# stage on the device.
#   usage: tools/round_start.sh <tag>      e.g.  gpurun --timeout 1500 -- 'tools/round_start.sh r02_start'
TAG=${1:-round_start}
D=gpurun_out/$TAG
mkdir -p $D
timeout 900 python -m pytest tests -m gpu -x -q > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1
timeout 300 python bench.py > $D/bench.json 2> $D/bench.err
for t in 16 32 64 128 0; do
  timeout 120 python tools/e2e.py --batches 48 --threads $t > $D/e2e_t$t.txt 2>&1
done
timeout 120 python tools/e2e.py --batches 48 --callers 2 > $D/e2e_c2.txt 2>&1
timeout 120 python tools/e2e.py --batches 96 --device-entropy > $D/e2e_dev.txt 2>&1
timeout 120 python tools/e2e.py --batches 48 --packets 16384 --device-entropy > $D/e2e_dev16k.txt 2>&1
timeout 120 python tools/batch_host_bench.py --threads 1 16 32 64 128 > $D/batch_host.txt 2>&1
tail -n 3 $D/pytest.log; cat $D/smoke.log | tail -1; cat $D/bench.json | head -c 600; echo; grep -h "end-to-end" $D/e2e_*.txt
