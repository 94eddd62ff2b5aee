cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3/tests_full.txt 2>&1; tail -5 gpurun_out/r3/tests_full.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step_reps'], d['step_launch'][:40], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
