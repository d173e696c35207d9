#!/bin/bash
# (GPU box) round 5, session 10: two tenants on one GPU.  Which CUs a masked stream gets (tools/micro/cumask), parity of the
# sharder with its logical shards on their own XCDs, and the sharder probe with and without the shares.
D=gpurun_out/r05_s10; mkdir -p $D
( cd tools/micro && timeout 120 ./cumask ) > $D/cumask.txt 2>&1; cat $D/cumask.txt
( time timeout 900 python -m pytest tests/test_gpu_shapes.py -m gpu -q -k "tenants or sharder" ) > $D/pytest.txt 2>&1; tail -5 $D/pytest.txt
for share in 1 0 1 0; do
  timeout 300 python tools/probe/sharder_probe.py 2 200 4096 $share 2>&1 | head -2 | tee -a $D/probe.txt
done
timeout 300 python tools/probe/sharder_probe.py 1 200 8192 1 2>&1 | head -2 | tee -a $D/probe.txt
timeout 300 python tools/probe/sharder_probe.py 4 200 2048 1 2>&1 | head -2 | tee -a $D/probe.txt
timeout 300 python tools/probe/sharder_probe.py 4 200 2048 0 2>&1 | head -2 | tee -a $D/probe.txt
bash tools/probe/sharder_trace.sh 2 > $D/trace_share.txt 2>&1; tail -40 $D/trace_share.txt
