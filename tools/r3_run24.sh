cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_linear_gpu.py tests/test_bf16x3_gpu.py tests/test_model_gpu.py tests/test_graph_exec_gpu.py tests/test_rccl_gpu.py -x -q 2>&1 | grep -E "passed|failed|^E |FAILED" | head -8
for rep in 1 2; do for v in batched immediate; do NSDP_WGRAD_REDUCE=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('REDUCE=$v', d['ms_per_step'], d['final_loss'], d['step_launch'][-95:-60])"; done; done
for v in batched immediate; do NSDP_WGRAD_REDUCE=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 --batch 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=8 REDUCE=$v', d['ms_per_step'], d['step_launch'][-95:-60])"; done
