#!/bin/bash
# A/B of an environment knob inside one GPU box:  tools/ab.sh KNOB "v1 v2 ..." [bench args]
knob=$1; vals=$2; shift 2
for rep in 1 2; do
  for v in $vals; do
    env $knob=$v python bench.py "$@" --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$knob=$v', d['ms_per_step'], 'loss', d['final_loss'])"
  done
done
