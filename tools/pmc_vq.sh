#!/bin/bash
# PMC counters of k_residue_vq (GPU box).  Usage: tools/pmc_vq.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcvq
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"
P2="SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM"
i=1
for P in "$P1" "$P2"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline --device-vq --settle-ms 10 > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_residue_vq" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 256 * 1024:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in agg.items()}
w = out.get("SQ_WAVES", 1)
print(json.dumps({k: round(v / w, 1) for k, v in sorted(out.items())}, indent=1))
PY
