cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_graph_exec_gpu.py tests/test_bench_gpu.py -x -q > gpurun_out/r3/tests6.txt 2>&1; tail -8 gpurun_out/r3/tests6.txt
for args in "8 f32" "32 f32" "32 bf16" "8 bf16" "32 bf16 arbitrary"; do
  echo "=== $args"; timeout 600 python tools/try_graph_exec.py $args 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"
done > gpurun_out/r3/graph_exec.txt 2>&1
cat gpurun_out/r3/graph_exec.txt
