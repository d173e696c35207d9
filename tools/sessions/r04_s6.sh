#!/bin/bash
# (GPU box) round 4, session 6: full GPU parity of the working tree, the whole bench line (end_to_end incl. the sharder, other_configs)
D=gpurun_out/r04_s6; mkdir -p $D
timeout 900 python -m pytest tests -m gpu -x -q > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log
tail -4 $D/pytest.log
( time python bench.py --steps 20 --warmup 5 ) > $D/bench_k20.json 2> $D/bench_k20.err
python3 - <<PY
import json
for l in open("$D/bench_k20.json"):
    if l.startswith("{"):
        d=json.loads(l); e=d["end_to_end"]
        print("value %.1fM wall %.2f events %.2f frac %.4f"%(d["value"]/1e6,d["ms_per_step"]*1e3,d["roofline"]["launch_ms"]*1e3,d["roofline"]["frac"]))
        print("e2e host %.2fM; device entropy %.2fM (d2h %.1f GB/s); large %.2fM (d2h %.1f); sharder %s"%(e["value"]/1e6, e["device_entropy"]["value"]/1e6, e["device_entropy"]["d2h_GBps"], e["device_entropy"]["large_batches"]["value"]/1e6, e["device_entropy"]["large_batches"]["d2h_GBps"], {k:e["sharder"].get(k) for k in ("value","shards","packets_per_call","error")}))
        for k,v in d["other_configs"].items(): print(" ",k, v.get("us_per_launch"), v.get("frac"), v.get("parity","")[:30], v.get("error",""))
PY
tail -3 $D/bench_k20.err
