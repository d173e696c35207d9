#!/bin/bash
# rocprofv3 kernel stats of tools/bench_configs.py (GPU box).  usage: tools/prof_cfg.sh [configs, default 3] [steps] [out dir under gpurun_out]
C=${1:-3}; K=${2:-80}; OUT=$GRAFT_REPO_ROOT/gpurun_out/${3:-prof_cfg}
mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only $C --steps $K --no-verify > $OUT/out.txt 2> $OUT/rocprof.log
cut -c1-200 $OUT/out.txt
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/stats_kernel_stats.csv")):
    print("%-60s calls %6s avg %10.1f us  total %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
