#!/bin/bash
# usage: tools/ab.sh "<flags A>" "<flags B>" [reps]  -- interleaved A/B of two builds with the default bench (200 steps)
FA=$1; FB=$2; R=${3:-2}
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then F="$FA"; else F="$FB"; fi
    LW_EXTRA_FLAGS="$F" python lewton_amd/build.py --force > /dev/null 2>&1
    python bench.py --no-cpu-baseline > gpurun_out/ab/$v$r.json 2>/dev/null
  done
done
