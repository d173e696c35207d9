#!/bin/bash
# (GPU box) timeline of the device-entropy staging ring: kernel and memory-copy trace (no counters), gaps between consecutive
# k_entropy launches and between consecutive large D2H copies, and how much of the copy engine's time the ring keeps busy
#   usage: tools/e2e_gaps.sh [packets per batch] [batches] [out dir]
P=${1:-4096}; B=${2:-60}; OUT=${3:-gpurun_out/e2e_gaps}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/$OUT; cd /tmp; rm -rf /tmp/gp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gp -o t -- python $R/tools/e2e.py --batches $B --packets $P --threads 0 --device-entropy --slots 3 > /tmp/gp.log 2>&1
grep -h "end-to-end" /tmp/gp.log | cut -c1-200
python3 - $P > $R/$OUT/timeline_$P.txt <<PY
import csv, glob, sys
P = int(sys.argv[1])
kf = glob.glob("/tmp/gp/**/*kernel_trace.csv", recursive=True)[0]
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:10]) for r in csv.DictReader(open(kf))]
k.sort()
ent = [x for x in k if x[2].startswith("k_entropy")]
mf = glob.glob("/tmp/gp/**/*memory_copy_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(mf)))
m = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")) for r in rows)
big = [x for x in m if x[1] - x[0] > 100000 and "DEVICE_TO_HOST" in x[2].upper().replace(" ", "_")]
if not big:
    big = [x for x in m if x[1] - x[0] > 100000]
n0 = len(big) // 4
sel = big[n0:len(big) - 2]
dur = [(e - s) / 1e3 for s, e, _ in sel]
gaps = [(sel[i + 1][0] - sel[i][1]) / 1e3 for i in range(len(sel) - 1)]
per = (sel[-1][0] - sel[0][0]) / 1e3 / (len(sel) - 1)
print("packets per batch %d" % P)
print("large D2H copies: %d, duration avg %.0f us (min %.0f max %.0f), gap to the next avg %.0f us (min %.0f max %.0f), period %.0f us" % (
    len(sel), sum(dur) / len(dur), min(dur), max(dur), sum(gaps) / len(gaps), min(gaps), max(gaps), per))
e2 = ent[len(ent) // 4:len(ent) - 2]
d = [(e - s) / 1e3 for s, e, _ in e2]
print("k_entropy: duration avg %.0f us, period %.0f us" % (sum(d) / len(d), (e2[-1][0] - e2[0][0]) / 1e3 / (len(e2) - 1)))
# one steady-state stretch, relative times: kernels and copies interleaved
t0 = sel[2][0]
ev = [(s, e, "D2H") for s, e, _ in sel[2:6]] + [(s, e, n) for s, e, n in k if sel[2][0] - 3e6 < s < sel[5][1]]
ev += [(s, e, "copy " + d_) for s, e, d_ in m if sel[2][0] - 3e6 < s < sel[5][1] and (s, e, d_) not in sel and e - s > 20000]
for s, e, n in sorted(ev):
    print("%9.0f .. %9.0f us  (%7.0f us)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
cat $R/$OUT/timeline_$P.txt | head -60
