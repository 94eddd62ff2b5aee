cd /root/repo; export TMPDIR=/tmp
for rep in 1 2; do for v in 0 1; do NSDP_PAIR_MASK=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('PAIR_MASK=$v', d['ms_per_step'], d['final_loss'])"; done; done
for v in 0 1; do NSDP_PAIR_MASK=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 --batch 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=8 PAIR_MASK=$v', d['ms_per_step'])"; done
for v in 0 1; do NSDP_PAIR_MASK=$v python bench.py --no-cpu-baseline --steps 10 --warmup 3 --reps 1 --workload arbitrary_train 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('arbitrary f32 PAIR_MASK=$v', d['ms_per_step'])"; done
