#!/bin/bash
# interleaved A/B of whole environments inside one GPU box:  tools/ab_env.sh REPS "ENV1" "ENV2" ...   (an ENV is "A=1 B=2"; "-" = none)
reps=$1; shift
for rep in $(seq $reps); do
  for v in "$@"; do
    e=$v; [ "$v" = "-" ] && e="NSDP_AB_NONE=1"
    env $e python bench.py --no-cpu-baseline --steps 15 --warmup 3 ${AB_BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], 'loss', d['final_loss'])"
  done
done
