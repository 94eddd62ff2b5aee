cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_geometry_gpu.py tests/test_pointnet2_modules.py tests/test_alternates.py -x -q > gpurun_out/r3/tests11.txt 2>&1; tail -5 gpurun_out/r3/tests11.txt
timeout 600 python tools/bench_grouping.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" > gpurun_out/r3/grouping.txt
grep "group_points_grad\|gather_points_grad" gpurun_out/r3/grouping.txt
