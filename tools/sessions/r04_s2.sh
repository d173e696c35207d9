#!/bin/bash
# (GPU box) round 4, session 2: submission variants of the bench line at K = 20; timing-only tail variants of k_long
D=gpurun_out/r04_s2; mkdir -p $D
Q="--no-cpu-baseline --no-end-to-end --no-other-configs --steps 20 --warmup 5"
for i in 1 2 3; do
  python bench.py $Q --graph-segments "" > $D/k20_one_$i.json 2>/dev/null
  python bench.py $Q --graph-segments "1" > $D/k20_s1_$i.json 2>/dev/null
  python bench.py $Q --graph-segments "1,3" > $D/k20_s13_$i.json 2>/dev/null
  python bench.py $Q --graph-segments "2,6" > $D/k20_s26_$i.json 2>/dev/null
done
tools/ab_so.sh 3 2000 magic skip14 skip12 skip8 > $D/ab.txt 2>&1
cat $D/ab.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$D/k20_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], "wall us/step %.2f"%(d["ms_per_step"]*1e3), "events %.2f"%(d["roofline"]["launch_ms"]*1e3), "value %.1fM"%(d["value"]/1e6), d["config"]["parity"][:20])
PY
