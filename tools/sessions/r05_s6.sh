#!/bin/bash
# (GPU box) round 5, session 6: pacing variants of k_long12; PMC of the mixed configurations (bank conflicts of the block kernels)
D=gpurun_out/r05_s6; mkdir -p $D
timeout 900 tools/ab_cfg.sh 11 3 400 4096 q1 q2 q3 q4 q2pl16 q16 > $D/ab11.txt 2>&1; cat $D/ab11.txt
timeout 600 bash tools/pmc_cfg.sh 3 r05_s6 > $D/pmc3.txt 2>&1; tail -42 $D/pmc3.txt | head -40
timeout 600 bash tools/pmc_cfg.sh 14 r05_s6 > $D/pmc14.txt 2>&1; tail -42 $D/pmc14.txt | head -40
