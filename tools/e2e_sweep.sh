#!/bin/bash
# (GPU box) end-to-end rate through the staging ring over callers x host threads; long enough (>= 1 s each) for the
# container's CFS quota (cpu.max) to show
B=${1:-600}
for cfg in "1 8" "1 12" "1 16" "1 20" "1 24" "1 32" "2 8" "2 12"; do set -- $cfg; echo "callers $1 threads $2: $(python tools/e2e.py --callers $1 --threads $2 --batches $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%.2f M pkt/s  host alone %.2f M' % (d['value'] / 1e6, d['host_entropy_stage_alone'] / 1e6))")"; done
cat /sys/fs/cgroup/cpu.max; grep "nr_throttled\|throttled_usec" /sys/fs/cgroup/cpu.stat; uptime
