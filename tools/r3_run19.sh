cd /root/repo; export TMPDIR=/tmp
for wl in forward_eval dense_inference; do for g in "" "--graph"; do python bench.py --no-cpu-baseline --workload $wl $g --reps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$wl $g', d['ms_per_step'], d['host_enqueue_unblocked_ms'], d['step_launch'][:60])"; done; done
