#!/bin/bash
# (GPU box) time k_residue_vq in the bench workload: tools/vq_exp.sh ["<extra hipcc flags>" ...]
for F in "${@:-}"; do
  if [ -n "$F" ]; then LW_EXTRA_FLAGS="$F" python lewton_amd/build.py --force > /dev/null 2>&1; fi
  tools/prof.sh vqx --device-vq --steps 160 --warmup 16 --settle-ms 20 > /dev/null 2>&1
  python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open("gpurun_out/vqx/prof/stats_kernel_stats.csv")):
    if "k_residue_vq" in r["Name"] or "k_long<0, false>" in r["Name"]:
        print("flags [%s] %-14s calls %s avg %.1f us" % (sys.argv[1], r["Name"][:14], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
