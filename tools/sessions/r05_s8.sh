#!/bin/bash
# (GPU box) round 5, session 8: descriptor preload behind the image rows (variant pre2) against HEAD
D=gpurun_out/r05_s8; mkdir -p $D
for c in 3 14; do
  timeout 600 tools/ab_cfg.sh $c 3 800 4096 head pre2 > $D/ab$c.txt 2>&1; echo "config $c"; cat $D/ab$c.txt
done
