#!/bin/bash
D=gpurun_out/r04_s7; mkdir -p $D; rm -f $D/*.json
Q="--no-cpu-baseline --no-end-to-end --no-other-configs --steps 2000 --warmup 200"
cp lewton_amd/_lib/liblewton_amd.so /tmp/keep.so
for i in 1 2; do
  cp lewton_amd/_lib/variant_tmp.so lewton_amd/_lib/liblewton_amd.so; python bench.py $Q --no-prefetch-next > $D/temporal_pfoff_$i.json 2>$D/err.txt
  python bench.py $Q > $D/temporal_pfon_$i.json 2>>$D/err.txt
done
cp /tmp/keep.so lewton_amd/_lib/liblewton_amd.so
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$D/*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], "events %.2f us"%(d["roofline"]["launch_ms"]*1e3), d["config"]["parity"][-24:])
PY
