#!/bin/bash
# (GPU box) PMC passes for every kernel of one tools/bench_configs.py configuration; one rocprofv3 --pmc set per run,
# --kernel-trace only.  -> gpurun_out/<tag>/pmc_cfg_<config>.json      usage: tools/pmc_cfg.sh <config number> <tag>
CFG=${1:-11}; TAG=${2:-pmc_cfg}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_cfg_$CFG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=1
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only $CFG --steps 16 --no-verify > $OUT/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, cs in agg.items():
    pm = {c: sum(v) / len(v) for c, v in cs.items()}
    w = pm.get("SQ_WAVES", 0)
    d = {"launches_per_counter": min(len(v) for v in cs.values()), "waves": w}
    if w:
        d["per_wave"] = {c: round(pm[c] / w, 1) for c in pm if c.startswith("SQ_") and c != "SQ_WAVES"}
    for c in ("GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"):
        if c in pm:
            d[c] = pm[c]
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        d["hbm_read_bytes_per_launch"] = 2.0 * pm["FETCH_SIZE"] * 1024   # gfx950: FETCH_SIZE counts half of a wide read
        d["hbm_write_bytes_per_launch"] = pm["WRITE_SIZE"] * 1024
    out[name] = d
json.dump(out, open("$OUT/../pmc_cfg_$CFG.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
