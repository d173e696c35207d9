#!/bin/bash
# (GPU box) round 4, session 3: tail split of k_long -- parity, then interleaved A/B of LW_TAIL_FROM = 8 / 10 / 12 / 16 (off)
D=gpurun_out/r04_s3; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py tests/test_gpu_quoted_shapes.py -m gpu -x -q > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log
tail -5 $D/pytest.log
tools/ab_so.sh 3 2000 tail16 tail8 tail10 tail12 > $D/ab.txt 2>&1
cat $D/ab.txt
for v in tail16 tail8; do cp lewton_amd/_lib/variant_$v.so lewton_amd/_lib/liblewton_amd.so; python tools/bench_configs.py --only 10,5 --steps 200 > $D/cfg_$v.jsonl 2>&1; done
cat $D/cfg_*.jsonl | cut -c1-230
