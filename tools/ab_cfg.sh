#!/bin/bash
# (GPU box) interleaved comparison of lewton_amd/_lib/variant_<X>.so builds on ONE tools/bench_configs.py configuration.
# usage: tools/ab_cfg.sh <config> <reps> <steps> <packets> variants...
C=$1; R=$2; K=$3; P=$4; shift; shift; shift; shift
V=$@
O=gpurun_out/ab_cfg_$C; mkdir -p $O; rm -f $O/*.json
cp lewton_amd/_lib/liblewton_amd.so /tmp/keep.so
for r in $(seq 1 $R); do
  for v in $V; do
    cp lewton_amd/_lib/variant_$v.so lewton_amd/_lib/liblewton_amd.so
    python tools/bench_configs.py --only $C --steps $K --packets $P --no-verify > $O/$v$r.json 2>/dev/null
  done
done
cp /tmp/keep.so lewton_amd/_lib/liblewton_amd.so
python3 - $O $V <<PY
import json, glob, sys
o = sys.argv[1]
for v in sys.argv[2:]:
    xs = []
    for f in sorted(glob.glob("%s/%s[0-9].json" % (o, v))):
        for l in open(f):
            if l.startswith("{"):
                xs.append(json.loads(l)["us_per_launch"])
    print(v, " ".join("%.2f" % x for x in xs), "median %.2f us" % sorted(xs)[len(xs) // 2] if xs else "no data")
PY
