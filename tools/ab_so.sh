#!/bin/bash
# (GPU box) interleaved A/B of lewton_amd/_lib/variant_A.so and variant_B.so with the default bench. usage: tools/ab_so.sh [reps] [steps]
R=${1:-3}; K=${2:-2000}
mkdir -p gpurun_out/ab
cp lewton_amd/_lib/liblewton_amd.so /tmp/keep.so
for r in $(seq 1 $R); do
  for v in A B; do
    cp lewton_amd/_lib/variant_$v.so lewton_amd/_lib/liblewton_amd.so
    python bench.py --no-cpu-baseline --steps $K --warmup 200 > gpurun_out/ab/$v$r.json 2>/dev/null
  done
done
cp /tmp/keep.so lewton_amd/_lib/liblewton_amd.so
python3 - <<PY
import json, glob
for v in "AB":
    xs = []
    for f in sorted(glob.glob("gpurun_out/ab/%s*.json" % v)):
        for l in open(f):
            if l.startswith("{"):
                xs.append(json.loads(l)["roofline"]["launch_ms"] * 1e3)
    print(v, " ".join("%.2f" % x for x in xs), "median %.2f us" % sorted(xs)[len(xs) // 2])
PY
