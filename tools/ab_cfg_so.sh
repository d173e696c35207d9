#!/bin/bash
# (GPU box) interleaved comparison of lewton_amd/_lib/variant_<X>.so builds on one tools/bench_configs.py configuration
# usage: tools/ab_cfg_so.sh <config> <reps> <variants...>
C=$1; R=$2; shift; shift
cp lewton_amd/_lib/liblewton_amd.so /tmp/keep.so
for r in $(seq 1 $R); do
  for v in "$@"; do
    cp lewton_amd/_lib/variant_$v.so lewton_amd/_lib/liblewton_amd.so
    python tools/bench_configs.py --only $C --steps 400 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['us_per_launch'], d['pct_of_8TBps'], d['kernels'], d['parity'][:40])
"
  done
done
cp /tmp/keep.so lewton_amd/_lib/liblewton_amd.so
