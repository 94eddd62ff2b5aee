cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
for args in "8 f32" "32 f32" "32 bf16" "8 bf16"; do
  echo "=== $args"; timeout 600 python tools/try_graph_exec.py $args 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"
done > gpurun_out/r3/graph_exec.txt 2>&1
cat gpurun_out/r3/graph_exec.txt
