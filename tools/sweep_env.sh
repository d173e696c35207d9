#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 ..." [reps] [steps] -- (GPU box) bench the current library under VAR=v for each v, interleaved
VAR=$1; VALS=$2; R=${3:-2}; K=${4:-2000}
mkdir -p gpurun_out/sweep; rm -f gpurun_out/sweep/*.json
for r in $(seq 1 $R); do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-cpu-baseline --steps $K --warmup 200 > gpurun_out/sweep/${v}_$r.json 2>/dev/null
  done
done
python3 - "$VALS" <<'PY'
import json, glob, sys
for v in sys.argv[1].split():
    xs = []
    for f in sorted(glob.glob("gpurun_out/sweep/%s_*.json" % v)):
        for l in open(f):
            if l.startswith("{"):
                xs.append(json.loads(l)["roofline"]["launch_ms"] * 1e3)
    print(v, " ".join("%.2f" % x for x in xs))
PY
