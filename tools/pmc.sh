#!/bin/bash
# PMC passes for the dominant kernel (run on the GPU box through gpurun). Usage: tools_pmc.sh <tag> [bench args...]
# Counters are collected in their own rocprofv3 runs with --kernel-trace only (no other trace domains).
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if ("k_long" in k or "generic" in k) and len(next(iter(v.values()))) > 4:
            print(f.split("/")[-2], k, {c:(sum(x)/len(x)) for c,x in v.items()}, "n=",len(next(iter(v.values()))))
PY
