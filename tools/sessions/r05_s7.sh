#!/bin/bash
# (GPU box) round 5, session 7: the short role's slot descriptor requested in front of the barrier (k_mix, k_mix10) against HEAD, interleaved
D=gpurun_out/r05_s7; mkdir -p $D
timeout 600 python -m pytest tests/test_gpu_long10.py tests/test_gpu_quoted_shapes.py -m gpu -x -q -k "mix" > $D/pytest_mix.log 2>&1; echo "rc=$?" >> $D/pytest_mix.log
tail -4 $D/pytest_mix.log
for c in 3 14 15; do
  timeout 600 tools/ab_cfg.sh $c 3 800 4096 head pre > $D/ab$c.txt 2>&1; echo "config $c"; cat $D/ab$c.txt
done
