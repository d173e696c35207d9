#!/bin/bash
# (GPU box) round 5, session 3: the conflict-free gather of the block kernels (k_short<8 / 16 / 32>, k_mix, k_long10) against the
# oracle, pacing variants of k_long10, the mixed lines and their kernel breakdown
D=gpurun_out/r05_s3; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_quoted_shapes.py tests/test_gpu_long10.py tests/test_gpu_parity.py -m gpu -x -q > $D/pytest_a.log 2>&1; echo "rc=$?" >> $D/pytest_a.log
tail -6 $D/pytest_a.log
timeout 600 python tools/fuzz_gpu_mixed.py --rounds 40 --seed 51 > $D/fuzz_mixed.txt 2>&1; tail -4 $D/fuzz_mixed.txt
timeout 600 python tools/fuzz_gpu_mixed.py --rounds 60 --seed 52 --mid > $D/fuzz_mid.txt 2>&1; tail -4 $D/fuzz_mid.txt
timeout 1200 tools/ab_cfg.sh 12 3 800 4096 p1 p2 p3 p4 p8 p2ff4 p2ff5 p2ff9 p2pl14 p2pl16 p4ff4 p3ff5 > $D/ab12.txt 2>&1; cat $D/ab12.txt
timeout 600 tools/ab_cfg.sh 12 2 600 16384 p1 p2 p3 p4 > $D/ab12_16k.txt 2>&1; cat $D/ab12_16k.txt
for k in 3 14 15; do
  timeout 300 python tools/bench_configs.py --only $k --steps 600 >> $D/cfg.jsonl 2>> $D/cfg.err
done
cat $D/cfg.jsonl
timeout 300 bash tools/prof_cfg.sh 14 200 r05_s3/prof14 > $D/prof14.txt 2>&1; tail -5 $D/prof14.txt
timeout 300 bash tools/prof_cfg.sh 3 200 r05_s3/prof3 > $D/prof3.txt 2>&1; tail -4 $D/prof3.txt
