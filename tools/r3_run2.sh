cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_geometry_gpu.py tests/test_pointnet2_modules.py tests/test_alternates.py tests/test_rccl_gpu.py tests/test_bench_gpu.py -x -q > gpurun_out/r3/tests2.txt 2>&1
tail -15 gpurun_out/r3/tests2.txt
timeout 600 python tools/bench_grouping.py > gpurun_out/r3/grouping.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" gpurun_out/r3/grouping.txt
timeout 600 python bench.py --force-reducer --no-cpu-baseline > gpurun_out/r3/bench_force_reducer_nccl.json 2> gpurun_out/r3/bench_force_reducer.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3/bench_force_reducer_nccl.json') if l.startswith('{')][0])
print({k:d[k] for k in ('ms_per_step','ms_per_step_reps','host_enqueue_ms_per_step','host_enqueue_unblocked_ms','comm')})
PY
