cd /root/repo; export TMPDIR=/tmp
for rep in 1 2; do for cfg in "NSDP_WGRAD_STREAMS=1 NSDP_GRAPH_STREAMS=2" "NSDP_WGRAD_STREAMS=2 NSDP_GRAPH_STREAMS=4" "NSDP_WGRAD_STREAMS=3 NSDP_GRAPH_STREAMS=5" "NSDP_WGRAD_STREAMS=1 NSDP_GRAPH_STREAMS=3"; do env $cfg python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 --batch 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=8 $cfg', d['ms_per_step'], d['step_launch'][-70:])"; done; done
for cfg in "NSDP_WGRAD_STREAMS=1 NSDP_GRAPH_STREAMS=2" "NSDP_WGRAD_STREAMS=2 NSDP_GRAPH_STREAMS=4"; do env $cfg python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=32 $cfg', d['ms_per_step'])"; done
