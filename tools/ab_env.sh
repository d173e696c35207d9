#!/bin/bash
# usage: tools/ab_env.sh "<ENV=.. for A>" "<ENV=.. for B>" [reps]  -- interleaved A/B of two run-time settings, default bench
EA=$1; EB=$2; R=${3:-3}
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  env $EA python bench.py --no-cpu-baseline > gpurun_out/ab/A$r.json 2>/dev/null
  env $EB python bench.py --no-cpu-baseline > gpurun_out/ab/B$r.json 2>/dev/null
done
