export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/gp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gp -o t -- python $GRAFT_REPO_ROOT/tools/e2e.py --batches 60 --threads 1 --device-entropy --slots 3 > /tmp/gp.log 2>&1
tail -1 /tmp/gp.log | cut -c1-100
python3 - <<PY
import csv
k = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:12]) for r in csv.DictReader(open("/tmp/gp/t_kernel_trace.csv"))]
k.sort()
ent = [x for x in k if x[2].startswith("k_entropy")]
ent = ent[10:50]
gaps = [(ent[i+1][0] - ent[i][1]) / 1e3 for i in range(len(ent) - 1)]
dur = [(e - s) / 1e3 for s, e, _ in ent]
print("k_entropy dur avg %.0f us; gap between consecutive k_entropy: avg %.0f us min %.0f max %.0f" % (sum(dur)/len(dur), sum(gaps)/len(gaps), min(gaps), max(gaps)))
print("period %.0f us" % ((ent[-1][0] - ent[0][0]) / 1e3 / (len(ent) - 1)))
try:
    m = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", ""))) for r in csv.DictReader(open("/tmp/gp/t_memory_copy_trace.csv"))]
    m.sort()
    big = [x for x in m if x[1] - x[0] > 100000][10:40]
    print("large copies: avg %.0f us" % (sum(e - s for s, e, _ in big) / len(big) / 1e3), big[0][2])
except Exception as e:
    print("no copy trace", e)
PY
