#!/bin/bash
# (GPU box) PMC passes for the two kernels of the mixed short/long configuration (BASELINE configs[2]): k_long<EDGE>, k_short.
# One rocprofv3 --pmc set per run, --kernel-trace only.  -> gpurun_out/<tag>/pmc_mixed.json      usage: tools/pmc_mixed.sh <tag>
TAG=${1:-pmc_mixed}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG/pmc_mixed
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=1
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_configs.py --only 3 --steps 16 --no-verify > /dev/null 2>&1
  i=$((i+1))
done
python3 - <<PY
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace(" ", "")
        name = "k_mix" if "k_mix" in k else "k_short" if "k_short" in k else "k_long<EDGE>" if "k_long" in k and k.split("(")[0].endswith("true>") else None
        if name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
tot_r = tot_w = 0.0
for name, cs in agg.items():
    pm = {c: sum(v) / len(v) for c, v in cs.items()}
    w = pm.get("SQ_WAVES", 0)
    d = {"launches_per_counter": min(len(v) for v in cs.values()), "waves": w}
    if w:
        d["per_wave"] = {c: round(pm[c] / w, 1) for c in pm if c.startswith("SQ_") and c != "SQ_WAVES"}
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
        d["hbm_read_bytes_per_launch"] = 2.0 * pm["FETCH_SIZE"] * 1024   # gfx950: FETCH_SIZE counts half of a wide read
        d["hbm_write_bytes_per_launch"] = pm["WRITE_SIZE"] * 1024
        # (a step's kernels are launched dozens of times; the few launches of another kernel are the priming of the streams)
        if d["launches_per_counter"] >= 16:
            tot_r += d["hbm_read_bytes_per_launch"]
            tot_w += d["hbm_write_bytes_per_launch"]
            d["kernel_of_the_step"] = True
    out[name] = d
out["total_hbm_bytes_per_step"] = tot_r + tot_w
json.dump(out, open("$OUT/../pmc_mixed.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
