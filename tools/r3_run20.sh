cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_gpu.py -x -q 2>&1 | grep -E "passed|failed|^E |FAILED" | head -8
for wl in forward_eval dense_inference; do python bench.py --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline'] or {}; print('$wl', d['ms_per_step'], d['ms_per_step_reps'], d['step_launch'][:30], r.get('kernel'), r.get('frac'), round(d['value']/1e6,2))"; done
