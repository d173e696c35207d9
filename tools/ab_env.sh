#!/bin/bash
# (GPU box) interleaved comparison of environment settings with the default bench (current library).
# usage: tools/ab_env.sh <reps> <steps> "VAR=a" "VAR=b" ...
R=$1; K=$2; shift; shift
mkdir -p gpurun_out/abenv; rm -f gpurun_out/abenv/*.json
for r in $(seq 1 $R); do
  i=0
  for v in "$@"; do
    env $v python bench.py --no-cpu-baseline --no-end-to-end --steps $K --warmup 200 > gpurun_out/abenv/v$i.$r.json 2>/dev/null
    i=$((i+1))
  done
done
python3 - "$@" <<PY
import json, glob, sys
for i, v in enumerate(sys.argv[1:]):
    xs = []
    for f in sorted(glob.glob("gpurun_out/abenv/v%d.*.json" % i)):
        for l in open(f):
            if l.startswith("{"):
                xs.append(json.loads(l)["roofline"]["launch_ms"] * 1e3)
    print(v, " ".join("%.2f" % x for x in xs), "median %.2f us" % sorted(xs)[len(xs) // 2])
PY
