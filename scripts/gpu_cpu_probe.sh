#!/bin/bash
# what CPU share does this box give us? (for bench.py's all-cores baseline)
echo "nproc: $(nproc)  getconf: $(getconf _NPROCESSORS_ONLN)"
cat /sys/fs/cgroup/cpu.max 2>/dev/null || echo "no cgroup v2 cpu.max"
cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null || echo "no cgroup v1 quota"
cat /proc/self/cgroup | head -5
python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"
env | grep -i -E "omp|cpu|thread|slurm|nproc" | head
cat /proc/loadavg
