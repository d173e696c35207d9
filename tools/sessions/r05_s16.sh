#!/bin/bash
# (GPU box) round 5, session 16: two tenants on one GPU, final form (copier thread per device; CU shares optional)
D=gpurun_out/r05_s16; mkdir -p $D
( time timeout 300 python -m pytest tests/test_gpu_shapes.py tests/test_gpu_ring.py -m gpu -q -x ) > $D/pytest.txt 2>&1; tail -4 $D/pytest.txt
for rep in 1 2 3; do
  for cfg in "0 -1" "1 -1" "0 0"; do set -- $cfg
    timeout 30 python tools/probe/sharder_probe.py 2 400 4096 $1 $2 > /tmp/o.txt 2>&1; echo "rc $?" >> /tmp/o.txt; head -2 /tmp/o.txt | tee -a $D/probe.txt; grep "^rc" /tmp/o.txt
  done
done
timeout 30 python tools/probe/sharder_probe.py 1 400 8192 0 -1 2>&1 | head -2 | tee -a $D/probe.txt
timeout 30 python tools/probe/sharder_probe.py 4 400 2048 0 -1 2>&1 | head -2 | tee -a $D/probe.txt
timeout 30 python tools/probe/sharder_probe.py 4 400 2048 0 0 2>&1 | head -2 | tee -a $D/probe.txt
( time timeout 200 python bench.py --no-other-configs --no-cpu-baseline ) > $D/bench.txt 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r05_s16/bench.txt"):
    if ln.startswith("{"):
        d = json.loads(ln); e = d["end_to_end"]
        print("value %.1f M; device_entropy %.2f M large %.2f M; sharder %.2f M" % (d["value"] / 1e6, e["device_entropy"]["value"] / 1e6,
              e["device_entropy"]["large_batches"]["value"] / 1e6, e["sharder"]["value"] / 1e6))
PY
