#!/bin/bash
# PMC passes for the dominant kernel (run on the GPU box through gpurun). Usage: tools/pmc.sh <tag> [bench args...]
# Counters are collected in their own rocprofv3 runs with --kernel-trace only (no other trace domains), one --pmc set per
# pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Only the 4096-packet launches of the timed
# workload (256 workgroups x 1024 threads) are aggregated; the small priming / parity launches are ignored.
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
out["launches_per_counter"] = {k: len(v) for k, v in agg.items()}
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read
    # (MI355X_MICROARCH.md, HBM section) -> doubled
    out["hbm_read_bytes_per_launch"] = 2.0 * out["FETCH_SIZE"] * 1024
    out["hbm_write_bytes_per_launch"] = out["WRITE_SIZE"] * 1024
    out["hbm_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
if "SQ_WAVES" in out:
    w = out["SQ_WAVES"]
    out["per_wave"] = {k: out[k] / w for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in out}
json.dump(out, open("$OUT/../pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
