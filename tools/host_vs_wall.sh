for a in "--dtype bf16 --workload arbitrary_train" "--dtype bf16" "--batch 8" "--dtype bf16 --batch 8" "" "--workload arbitrary_train"; do
  python bench.py $a --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$a', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"
done
