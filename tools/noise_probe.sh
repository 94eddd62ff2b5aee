#!/bin/bash
# Back-to-back bench runs on one box with the cgroup throttle counters in between.  Before bench.py capped its CPU thread
# pools to the container's CPU quota (nsdp_amd/cpu_budget.py), later runs took 60-90 ms per step with unchanged kernel
# durations: 128-256 spinning OpenMP workers under a 16-CPU quota throttled the thread that enqueues the GPU work.
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --no-cpu-baseline --steps 15 --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('run $i', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], 'x3 avg us', round(1e3*d['roofline']['avg_launch_ms'],1))"
  grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr "\n" " "; echo
done
