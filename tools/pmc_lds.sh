#!/bin/bash
# LDS-pipe counters of k_long (4096-packet launches only). Usage: tools/pmc_lds.sh <tag>
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmclds
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PA="SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
PB="SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU"
PC="SQ_WAVES SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"
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
tail -3 $OUT/p1.log $OUT/p2.log $OUT/p3.log | grep -i "error\|invalid\|not" | head
