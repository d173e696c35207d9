#!/bin/bash
# (GPU box) instruction counts and wait cycles of k_entropy per wave (one rocprofv3 --pmc pass each)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ep
i=1
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/ep/p$i -o p -- python $GRAFT_REPO_ROOT/tools/ent_bench.py --reps 20 > /dev/null 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/ep/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_entropy" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
pm = {k: sum(v) / len(v) for k, v in agg.items()}
w = pm.get("SQ_WAVES", 1)
print("waves", w)
for k in sorted(pm):
    print("%-22s %12.0f per wave" % (k, pm[k] / w))
PY
