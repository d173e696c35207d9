#!/bin/bash
# (GPU box) one PMC pass (instruction counts + wave cycles) per library variant: tools/pmc_variants.sh O E G ...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcv
mkdir -p $OUT
cp $R/lewton_amd/_lib/liblewton_amd.so /tmp/keep.so
P="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH"
for v in "$@"; do
  cp $R/lewton_amd/_lib/variant_$v.so $R/lewton_amd/_lib/liblewton_amd.so
  (cd /tmp && rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/$v -o p -- python $R/bench.py --steps 16 --warmup 4 --settle-ms 5 --no-cpu-baseline > $OUT/$v.log 2>&1)
  (cd /tmp && rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d $OUT/${v}_2 -o p -- python $R/bench.py --steps 16 --warmup 4 --settle-ms 5 --no-cpu-baseline > $OUT/${v}_2.log 2>&1)
done
cp /tmp/keep.so $R/lewton_amd/_lib/liblewton_amd.so
python3 - "$@" <<PY
import csv, glob, collections, sys
for v in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % v, recursive=True) + glob.glob("$OUT/%s_2/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_long" in r["Kernel_Name"] and int(r["Grid_Size"]) == 256 * 1024:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    o = {k: sum(x) / len(x) for k, x in agg.items()}
    w = o.get("SQ_WAVES", 1) or 1
    print(v, " ".join("%s=%.0f" % (k.replace("SQ_", ""), o[k] / w) for k in sorted(o) if k != "SQ_WAVES"))
PY
