cd /root/repo; export TMPDIR=/tmp
for rep in 1 2; do for v in normal low; do NSDP_GRAPH_SIDE_PRIO=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('SIDE_PRIO=$v', d['ms_per_step'])"; done; done
for v in 2 4; do NSDP_GRAPH_STREAMS=$v python bench.py --no-cpu-baseline --steps 10 --warmup 3 --reps 1 --dtype bf16 --workload arbitrary_train 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('arbitrary bf16 STREAMS=$v', d['ms_per_step'])"; done
for v in 2 4; do NSDP_GRAPH_STREAMS=$v python bench.py --no-cpu-baseline --steps 10 --warmup 3 --reps 1 --batch 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=8 STREAMS=$v', d['ms_per_step'])"; done
