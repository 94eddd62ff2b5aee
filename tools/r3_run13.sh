cd /root/repo; export TMPDIR=/tmp
for rep in 1 2; do for v in 1 2 3 4 6; do NSDP_GRAPH_STREAMS=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('STREAMS=$v', d['ms_per_step'], d['step_launch'][-90:])"; done; done
for v in 1 2 4; do NSDP_GRAPH_STREAMS=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 --dtype bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('bf16 STREAMS=$v', d['ms_per_step'])"; done
