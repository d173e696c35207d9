#!/bin/bash
# (GPU box) round 4, session 1: parity of the working tree, the bench line's host-side variants at the driver's K = 20, the new
# other_configs object, and an interleaved A/B of k_long variants (lewton_amd/_lib/variant_*.so)
D=gpurun_out/r04_s1; mkdir -p $D
timeout 900 python -m pytest tests -m gpu -x -q > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log
Q="--no-cpu-baseline --no-end-to-end --no-other-configs --steps 20 --warmup 5"
for i in 1 2 3; do
  python bench.py $Q > $D/k20_default_$i.json 2>/dev/null
  python bench.py $Q --no-active-wait > $D/k20_nowait_$i.json 2>/dev/null
  python bench.py $Q --gate-us 20 > $D/k20_gate_$i.json 2>/dev/null
done
( time python bench.py --steps 20 --warmup 5 ) > $D/bench_full.json 2> $D/bench_full.err
tools/ab_so.sh 3 2000 base magic nodec noxq noboth > $D/ab.txt 2>&1
tail -3 $D/pytest.log; cat $D/ab.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$D/k20_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], "wall us/step %.2f"%(d["ms_per_step"]*1e3), "events %.2f"%(d["roofline"]["launch_ms"]*1e3), "value %.1fM"%(d["value"]/1e6))
PY
