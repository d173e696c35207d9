#!/bin/bash
# (GPU box) k_big variants x passes per workgroup: tools/big_sweep.sh "<variants>" "<passes list>" [config]
V=${1:-"w2"}; P=${2:-"5"}; C=${3:-11}
cp lewton_amd/_lib/liblewton_amd.so /tmp/keep.so
for v in $V; do
  cp lewton_amd/_lib/variant_$v.so lewton_amd/_lib/liblewton_amd.so
  for p in $P; do
    echo -n "variant $v passes $p config $C: "
    LW_TMP_BIG_PASSES=$p python tools/bench_configs.py --only $C --steps 200 --no-verify 2>&1 | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['us_per_launch'], 'us', d['pct_of_8TBps'], '%')"
  done
done
cp /tmp/keep.so lewton_amd/_lib/liblewton_amd.so
