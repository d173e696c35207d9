#!/bin/bash
# (GPU box) interleaved comparison of lewton_amd/_lib/variant_<X>.so builds with the default bench.
# usage: tools/ab_so.sh [reps] [steps] [variants...]   (default variants: A B)
R=${1:-3}; K=${2:-2000}; shift; shift
V=${@:-A B}
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/*.json
cp lewton_amd/_lib/liblewton_amd.so /tmp/keep.so
for r in $(seq 1 $R); do
  for v in $V; do
    cp lewton_amd/_lib/variant_$v.so lewton_amd/_lib/liblewton_amd.so
    python bench.py --no-cpu-baseline --no-end-to-end --no-other-configs --steps $K --warmup 200 > gpurun_out/ab/$v$r.json 2>/dev/null
  done
done
cp /tmp/keep.so lewton_amd/_lib/liblewton_amd.so
python3 - $V <<PY
import json, glob, sys
for v in sys.argv[1:]:
    xs = []
    for f in sorted(glob.glob("gpurun_out/ab/%s[0-9]*.json" % v)):
        for l in open(f):
            if l.startswith("{"):
                xs.append(json.loads(l)["roofline"]["launch_ms"] * 1e3)
    print(v, " ".join("%.2f" % x for x in xs), "median %.2f us" % sorted(xs)[len(xs) // 2])
PY
