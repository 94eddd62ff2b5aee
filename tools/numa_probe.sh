for f in /sys/class/drm/card*/device/numa_node; do echo $f $(cat $f); done 2>/dev/null | head
lscpu | grep -i "numa\|socket\|model name" | head -12
run() { timeout 300 taskset -c $1 python bench.py --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('cpus $1:', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"; }
run 0-15; run 64-79; run 128-143; run 192-207; run 0-15; run 64-79
