#!/bin/bash
# (GPU box) round 5, session 5: k_mix10 against the oracle and the two-launch path, the mixed lines, k_long10 after its refactoring
D=gpurun_out/r05_s5; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_long10.py -m gpu -x -q > $D/pytest_long10.log 2>&1; echo "rc=$?" >> $D/pytest_long10.log
tail -25 $D/pytest_long10.log
for k in 14 15 12 3; do
  timeout 300 python tools/bench_configs.py --only $k --steps 600 >> $D/cfg.jsonl 2>> $D/cfg.err
done
cat $D/cfg.jsonl
timeout 600 python tools/fuzz_gpu_mixed.py --rounds 60 --seed 61 --mid > $D/fuzz_mid.txt 2>&1; tail -3 $D/fuzz_mid.txt
timeout 600 python tools/fuzz_gpu_mixed.py --rounds 40 --seed 62 --big > $D/fuzz_big.txt 2>&1; tail -3 $D/fuzz_big.txt
timeout 300 bash tools/prof_cfg.sh 14 200 r05_s5/prof14 > $D/prof14.txt 2>&1; tail -5 $D/prof14.txt
