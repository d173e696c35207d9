#!/bin/bash
# (GPU box) average duration of k_entropy per 4096-packet batch under rocprofv3, for settings given as "VAR=val" arguments
export TMPDIR=/tmp
for v in "$@"; do
  rm -rf /tmp/entprof
  (cd /tmp && env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/entprof -o s -- python $GRAFT_REPO_ROOT/tools/e2e.py --batches 24 --threads 2 --device-entropy > /tmp/ent_e2e.log 2>&1)
  echo "$v: k_entropy $(grep k_entropy /tmp/entprof/s_kernel_stats.csv | awk -F, '{printf "%.0f us avg", $(NF-4)/1000}')  $(tail -1 /tmp/ent_e2e.log | cut -c1-70)"
done
