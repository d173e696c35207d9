#!/bin/bash
# (GPU box) round 5, session 20: k_mix's dense form against the two launches it would replace, and its long role by itself
D=gpurun_out/r05_s20; mkdir -p $D
( time timeout 300 python -m pytest tests/test_gpu_quoted_shapes.py -m gpu -q -x -k "dense_mixed" ) > $D/pytest.txt 2>&1; tail -4 $D/pytest.txt
for m in 0 2 0 2; do
  V=""; [ $m = 3 ] && V="--no-verify"
  echo "mix $m: $(timeout 120 python tools/bench_configs.py --only 3 --packets 16384 --steps 300 --mix $m $V 2>&1 | grep '^{' | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["us_per_launch"], d["kernels"], d["parity"][:40])')" | tee -a $D/ab.txt
done
for m in 0 2; do
  echo "65536, mix $m: $(timeout 120 python tools/bench_configs.py --only 3 --packets 65536 --steps 100 --mix $m 2>&1 | grep '^{' | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["us_per_launch"], d["kernels"], d["parity"][:40])')" | tee -a $D/ab.txt
done
timeout 200 python tools/fuzz_gpu_mixed.py --rounds 30 --seed 92 2>&1 | tail -1 | tee $D/fuzz.txt
