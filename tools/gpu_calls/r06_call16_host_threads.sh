#!/bin/bash
# Round 6, call 16: how many host threads the box really gives (nproc, cgroup quota) and what own_bvh.h's build costs there by thread count and phase (no GPU work).
O=gpurun_out/r06_call16; mkdir -p $O
{ echo "nproc: $(nproc)  online: $(getconf _NPROCESSORS_ONLN)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cfs quota: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
  grep -m1 "model name" /proc/cpuinfo; python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"
  timeout 600 tools/bin/own_bvh_bench 8700000; timeout 300 tools/bin/own_bvh_bench 2450000; } > $O/host_threads.log 2>&1
cat $O/host_threads.log
