#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (run on the GPU box through gpurun). Usage: tools/prof.sh <tag>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.log
python3 - <<PY
import csv, json
rows = [r for r in csv.DictReader(open("$OUT/stats_kernel_trace.csv")) if "k_long" in r["Kernel_Name"]]
def gs(r):
    return int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
big = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if gs(r) == 256 * 1024]
small = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if gs(r) != 256 * 1024]
out = {"kernel": "k_long<0,false>", "launches_4096_packets": len(big), "avg_ns_4096_packets": sum(big) / max(1, len(big)),
       "min_ns_4096_packets": min(big) if big else None, "max_ns_4096_packets": max(big) if big else None,
       "other_launches": len(small), "avg_ns_other": sum(small) / max(1, len(small)) if small else None}
json.dump(out, open("$OUT/../kernel_duration_summary.json", "w"), indent=1)
print(json.dumps(out))
PY
