#!/bin/bash
# (GPU box) round 5, session 19: k_mix's dense form -- parity against the two-launch path and the oracle, then the mixed shapes at
# 4096 / 16 384 / 65 536 packets per launch
D=gpurun_out/r05_s19; mkdir -p $D
( time timeout 500 python -m pytest tests/test_gpu_quoted_shapes.py -m gpu -q -x -k "mix or mixed" ) > $D/pytest.txt 2>&1; tail -6 $D/pytest.txt
timeout 120 python tools/bench_configs.py --only 3 --steps 600 2>&1 | grep "^{" | tee -a $D/cfg.jsonl | cut -c1-400
timeout 120 python tools/bench_configs.py --only 3 --packets 16384 --steps 300 2>&1 | grep "^{" | tee -a $D/cfg.jsonl | cut -c1-400
timeout 120 python tools/bench_configs.py --only 3 --packets 65536 --steps 100 2>&1 | grep "^{" | tee -a $D/cfg.jsonl | cut -c1-400
timeout 200 python tools/fuzz_gpu_mixed.py --rounds 40 --seed 91 2>&1 | tail -1 | tee $D/fuzz.txt
