#!/bin/bash
# (GPU box) round 5, session 4: k_long12 (blocksize_1 = 12, one wave per channel) against the oracle and k_big, its timing and counters
D=gpurun_out/r05_s4; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_long12.py -m gpu -x -q > $D/pytest_long12.log 2>&1; echo "rc=$?" >> $D/pytest_long12.log
tail -25 $D/pytest_long12.log
for k in 11; do
  timeout 300 python tools/bench_configs.py --only $k --steps 400 >> $D/cfg.jsonl 2>> $D/cfg.err
  timeout 300 python tools/bench_configs.py --only $k --steps 300 --packets 16384 >> $D/cfg.jsonl 2>> $D/cfg.err
done
cat $D/cfg.jsonl
timeout 900 python -m pytest tests/test_gpu_quoted_shapes.py -m gpu -x -q > $D/pytest_quoted.log 2>&1; echo "rc=$?" >> $D/pytest_quoted.log
tail -5 $D/pytest_quoted.log
timeout 600 bash tools/pmc_cfg.sh 11 r05_s4 > $D/pmc11.txt 2>&1; tail -45 $D/pmc11.txt
