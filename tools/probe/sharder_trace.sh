#!/bin/bash
# (GPU box) kernel + copy timeline of the sharder probe (two logical shards on one GPU)
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/gs
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gs -o t -- python $R/tools/probe/sharder_probe.py ${1:-2} ${2:-60} 4096 ${3:-1} > /tmp/gs.log 2>&1
grep -A1 "^shards=" /tmp/gs.log
python3 - <<PY
import csv, glob
kf = glob.glob("/tmp/gs/**/*kernel_trace.csv", recursive=True)[0]
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:9], r.get("Queue_Id", "")) for r in csv.DictReader(open(kf))]
mf = glob.glob("/tmp/gs/**/*memory_copy_trace.csv", recursive=True)[0]
m = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "")[12:], "") for r in csv.DictReader(open(mf)) if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 20000]
ev = sorted(k + m)
n = len(ev)
sel = ev[int(n * 0.6):int(n * 0.6) + ${4:-70}]
t0 = sel[0][0]
for s, e, nm, q in sel:
    print("%8.0f .. %8.0f (%6.0f us) %-22s q%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, nm, q))
PY
