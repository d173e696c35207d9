#!/bin/bash
# (GPU box) round 4, session 4: kernel arguments in device memory (HIP_FORCE_DEV_KERNARG); configs[2] / n = 1024 at larger launches
D=gpurun_out/r04_s4; mkdir -p $D
Q="--no-cpu-baseline --no-end-to-end --no-other-configs --steps 2000 --warmup 200"
for i in 1 2 3; do
  python bench.py $Q > $D/ka0_$i.json 2>/dev/null
  HIP_FORCE_DEV_KERNARG=1 python bench.py $Q > $D/ka1_$i.json 2>/dev/null
done
python tools/bench_configs.py --only 3,12 --steps 400 > $D/cfg_ka0.jsonl 2>/dev/null
HIP_FORCE_DEV_KERNARG=1 python tools/bench_configs.py --only 3,12 --steps 400 > $D/cfg_ka1.jsonl 2>/dev/null
python tools/bench_configs.py --only 3,12 --packets 16384 --steps 200 > $D/cfg_16k.jsonl 2>/dev/null
python tools/bench_configs.py --only 3,12 --packets 65536 --steps 100 > $D/cfg_64k.jsonl 2>/dev/null
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$D/ka*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], "events %.2f us"%(d["roofline"]["launch_ms"]*1e3), "wall %.2f"%(d["ms_per_step"]*1e3))
for f in sorted(glob.glob("$D/cfg_*.jsonl")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], d["config"][:28], d["packets_per_launch"], d["us_per_launch"], "us", d["pct_of_8TBps"], "%", d["parity"][:24])
PY
