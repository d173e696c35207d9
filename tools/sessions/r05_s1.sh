#!/bin/bash
# (GPU box) round 5, session 1: the new k_long10 (blocksize_1 = 10) and the k_mix error exit against the oracle, then the timing of
# the bs1 = 10 shapes with both kernels, then the rest of the GPU suite
D=gpurun_out/r05_s1; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_long10.py -m gpu -x -q > $D/pytest_long10.log 2>&1; echo "rc=$?" >> $D/pytest_long10.log
tail -15 $D/pytest_long10.log
timeout 600 python -m pytest tests/test_gpu_quoted_shapes.py -m gpu -x -q > $D/pytest_quoted.log 2>&1; echo "rc=$?" >> $D/pytest_quoted.log
tail -8 $D/pytest_quoted.log
for k in 12 14 15 3; do
  timeout 300 python tools/bench_configs.py --only $k --steps 600 >> $D/cfg.jsonl 2>> $D/cfg.err
done
timeout 300 python tools/bench_configs.py --only 12 --steps 600 --packets 16384 >> $D/cfg.jsonl 2>> $D/cfg.err
cat $D/cfg.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_long10.py --deselect tests/test_gpu_quoted_shapes.py > $D/pytest_rest.log 2>&1; echo "rc=$?" >> $D/pytest_rest.log
tail -8 $D/pytest_rest.log
