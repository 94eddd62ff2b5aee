cd /root/repo; export TMPDIR=/tmp
for rep in 1 2; do for v in normal high; do NSDP_GRAPH_SIDE_PRIO=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('SIDE_PRIO=$v', d['ms_per_step'])"; done; done
NSDP_GRAPH_SIDE_PRIO=high python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 --dtype bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('bf16 SIDE_PRIO=high', d['ms_per_step'])"
python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 --dtype bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('bf16 SIDE_PRIO=normal', d['ms_per_step'])"
