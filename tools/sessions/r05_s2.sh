#!/bin/bash
# (GPU box) round 5, session 2: k_long10 incl. its EDGE form and the 5.1 fixture, the block kernel's changed L = 16 instantiation, the
# mixed 512/1024 and 256/1024 lines, tuning variants of k_long10, its rocprofv3 duration and counters
D=gpurun_out/r05_s2; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_long10.py -m gpu -x -q > $D/pytest_long10.log 2>&1; echo "rc=$?" >> $D/pytest_long10.log
tail -15 $D/pytest_long10.log
timeout 900 python -m pytest tests/test_gpu_quoted_shapes.py tests/test_gpu_parity.py -m gpu -x -q -k "block_kernel or 9_10 or 8_10 or 8_9 or 1024" > $D/pytest_blk.log 2>&1; echo "rc=$?" >> $D/pytest_blk.log
tail -6 $D/pytest_blk.log
for k in 14 15 12; do
  timeout 300 python tools/bench_configs.py --only $k --steps 600 >> $D/cfg.jsonl 2>> $D/cfg.err
done
cat $D/cfg.jsonl
timeout 900 tools/ab_cfg.sh 12 3 800 4096 base ff4 ff10 ff16 pl16 pl8 pace2 pace16 > $D/ab12.txt 2>&1; cat $D/ab12.txt
timeout 300 bash tools/prof_cfg.sh 12 200 r05_s2/prof12 > $D/prof12.txt 2>&1; tail -6 $D/prof12.txt
timeout 600 bash tools/pmc_cfg.sh 12 r05_s2 > $D/pmc12.txt 2>&1; tail -60 $D/pmc12.txt
