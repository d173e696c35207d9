#!/bin/bash
# instruction-mix / front-end counters of k_long (4096-packet launches only). Usage: tools/pmc_mix.sh <tag>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmcmix
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PA="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32"
PB="SQ_WAVES SQC_ICACHE_MISSES SQC_ICACHE_HITS SQC_ICACHE_REQ SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE"
PC="SQ_WAVES SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL"
i=1
for P in "$PA" "$PB" "$PC"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --settle-ms 5 --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_long" in r["Kernel_Name"] and int(r["Grid_Size"]) == 256 * 1024:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in agg.items()}
w = out.get("SQ_WAVES", 1)
print(json.dumps({k: round(v / w, 2) for k, v in sorted(out.items())}, indent=1))
PY
