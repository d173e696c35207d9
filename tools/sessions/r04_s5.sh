#!/bin/bash
D=gpurun_out/r04_s5; mkdir -p $D
for t in 0 2 4; do
  python tools/e2e.py --batches 150 --packets 16384 --threads $t --device-entropy | tail -1 | cut -c1-200
  python tools/e2e.py --batches 300 --packets 4096 --threads $t --device-entropy | tail -1 | cut -c1-200
done
tools/e2e_gaps.sh 16384 60 $D
tools/e2e_gaps.sh 4096 120 $D | head -12
