#!/bin/bash
# (GPU box) round 6: k_prep after its restructuring (tables up front), the graph-branches experiment, another slice of the setup campaign
mkdir -p gpurun_out/s10
timeout 600 python -m pytest tests/test_gpu_prep.py tests/test_gpu_random_setups.py -x -q -m gpu 2>&1 | tail -3
P='import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d["config"][:50], d["packets_per_launch"], "branches", d.get("graph_branches"), d["us_per_launch"], d["pct_of_8TBps"], d["kernels"], d["parity"][:30])
'
python tools/bench_configs.py --only 19 2>/dev/null | python -c "$P"
(for B in 1 2 3; do python tools/bench_configs.py --only 3,14,15 --mix 0 --branches $B; python tools/bench_configs.py --only 12,7,9,11 --branches $B; done) 2>/dev/null > gpurun_out/s10/branches.jsonl
python -c "$P" < gpurun_out/s10/branches.jsonl
timeout 300 python tools/fuzz_gpu_setups.py --setups 1500 --seed 40000 --procs 14 --quiet 2>&1 | grep "^fuzz"
