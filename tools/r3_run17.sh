cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_graph_exec_gpu.py tests/test_eval_harness_gpu.py tests/test_bench_gpu.py -x -q 2>&1 | grep -E "passed|failed|^E |FAILED" | head -8
for flags in "--batch 8" "--batch 8 --dtype bf16" "--batch 2" "--batch 2 --eager" "" "--dtype bf16"; do python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 $flags 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$flags', d['ms_per_step'], d['step_launch'][-70:])"; done
bash tools/knob_matrix.sh NSDP_WGRAD_STREAM=0 NSDP_PARAM_GRADS=autograd NSDP_SCATTER_ROWS=atomic
