#!/bin/bash
# (GPU box) measurement set of the device entropy stage (k_entropy): kernel trace + stats of tools/ent_bench.py at two batch
# sizes, PMC passes (HBM bytes, instruction counts), the end-to-end sweep through the ring, one stream through the Ogg layer.
# Everything lands in gpurun_out/r02_tierc/.
D=$GRAFT_REPO_ROOT/gpurun_out/r02_tierc
mkdir -p $D
export TMPDIR=/tmp
cd /tmp
for n in 4096 16384; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof$n -o s -- python $GRAFT_REPO_ROOT/tools/ent_bench.py --packets $n --streams $((n/16)) --reps 100 > $D/ent_bench_$n.txt 2>/dev/null
done
i=1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D/pmc$i -o p -- python $GRAFT_REPO_ROOT/tools/ent_bench.py --reps 30 > /dev/null 2>&1
  i=$((i+1))
done
python3 - <<PY > $D/k_entropy_summary.json
import csv, glob, json, collections
out = {}
for n in (4096, 16384):
    rows = [r for r in csv.DictReader(open("$D/prof%d/s_kernel_trace.csv" % n)) if r["Kernel_Name"].startswith("k_entropy")]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    out["k_entropy_%d_packets" % n] = {"launches": len(d), "avg_us": sum(d) / len(d) / 1e3, "min_us": min(d) / 1e3, "max_us": max(d) / 1e3}
agg = collections.defaultdict(list)
for f in glob.glob("$D/pmc*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_entropy"):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
pm = {k: sum(v) / len(v) for k, v in agg.items()}
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    pm["hbm_read_bytes_per_launch"] = 2.0 * pm["FETCH_SIZE"] * 1024   # gfx950: FETCH_SIZE counts half of a wide read (MI355X_MICROARCH.md)
    pm["hbm_write_bytes_per_launch"] = pm["WRITE_SIZE"] * 1024
if "SQ_WAVES" in pm:
    pm["per_wave"] = {k: pm[k] / pm["SQ_WAVES"] for k in pm if k.startswith("SQ_INSTS") or k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")}
out["pmc_4096_packets"] = pm
print(json.dumps(out, indent=1, sort_keys=True))
PY
cd $GRAFT_REPO_ROOT
{
for cfg in "3 4096 256 1" "3 4096 256 2" "4 4096 256 2" "3 8192 512 2" "4 16384 1024 2"; do set -- $cfg
  echo "slots $1, $2 packets per batch, $4 host thread(s): $(timeout 120 python tools/e2e.py --batches $((1228800/$2)) --threads $4 --device-entropy --slots $1 --packets $2 --streams $3 2>&1 | tail -1 | cut -c1-260)"
done
echo "host stage for comparison: $(timeout 120 python tools/e2e.py --batches 300 --threads 20 2>&1 | tail -1 | cut -c1-260)"
make -C examples perf > /dev/null 2>&1; python tools/make_long_ogg.py /tmp/long.ogg 200000 > /dev/null 2>&1
for cfg in "4096 12" "4096 2 dev" "16384 2 dev"; do echo "examples/perf long.ogg $cfg: $(examples/perf /tmp/long.ogg $cfg 2>&1 | tail -1)"; done
} > $D/end_to_end.txt 2>&1
cat $D/k_entropy_summary.json | head -60; cat $D/end_to_end.txt
