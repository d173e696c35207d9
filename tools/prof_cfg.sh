#!/bin/bash
# rocprofv3 kernel stats of tools/bench_configs.py (GPU box)
OUT=$GRAFT_REPO_ROOT/gpurun_out/cfg/prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only 3 --steps 80 > $OUT/out.txt 2> $OUT/rocprof.log
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/stats_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.5:
        print("%-60s calls %6s avg %8.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
