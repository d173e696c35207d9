#!/bin/bash
# (GPU box) what the container may use of the host: cgroup CPU quota, affinity, NUMA
echo "nproc: $(nproc)"; echo "affinity: $(taskset -p $$ 2>/dev/null)"
for f in /sys/fs/cgroup/cpu.max /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us /sys/fs/cgroup/cpuset.cpus.effective /sys/fs/cgroup/cpuset/cpuset.cpus /sys/fs/cgroup/cpu.stat; do [ -r $f ] && echo "$f: $(cat $f | tr '\n' ' ')"; done
grep -i "Cpus_allowed_list\|Mems_allowed_list" /proc/self/status
numactl -H 2>/dev/null | head -8
uptime
