for rep in 1 2 3; do
  for v in "NSDP_K4_LINK=0" "NSDP_K4_LINK=1" "NSDP_K4_TAIL=0"; do
    env $v python bench.py --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', d['ms_per_step'], 'loss', d['final_loss'])"
  done
done
