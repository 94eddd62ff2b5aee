cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r3/tests_full.txt 2>&1; tail -12 gpurun_out/r3/tests_full.txt
for flags in "" "--eager" "--dtype bf16" "--dtype bf16 --eager" "--batch 8" "--batch 8 --eager" "--batch 8 --dtype bf16" "--workload arbitrary_train --dtype bf16" "--workload arbitrary_train --dtype bf16 --eager"; do
  name=$(echo "bench$flags" | tr ' -' '__' | tr -s '_')
  timeout 900 python bench.py --no-cpu-baseline $flags > gpurun_out/r3/$name.json 2> gpurun_out/r3/$name.err
  python - "$name" "$flags" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open('gpurun_out/r3/%s.json'%sys.argv[1]) if l.startswith('{')][0])
    print(sys.argv[2] or '(default)', d['ms_per_step'], d['ms_per_step_reps'], 'host unblocked', d['host_enqueue_unblocked_ms'], d['step_launch'][:60], 'frac', d['roofline']['frac'] if d['roofline'] else None, d.get('parity_l2_vs_fp32'))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
