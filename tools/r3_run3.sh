cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_geometry_gpu.py tests/test_pointnet2_modules.py tests/test_alternates.py tests/test_bf16x3_gpu.py tests/test_linear_gpu.py -x -q > gpurun_out/r3/tests3.txt 2>&1
tail -8 gpurun_out/r3/tests3.txt
timeout 600 python tools/bench_grouping.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" > gpurun_out/r3/grouping.txt
sed -n '/group_points/,$p' gpurun_out/r3/grouping.txt
timeout 600 python tools/ab_x3_wres.py > gpurun_out/r3/ab_x3_wres.txt 2>&1
cat gpurun_out/r3/ab_x3_wres.txt | grep "M="
for rep in 1 2; do for v in 0 256; do NSDP_X3_DBG=$v python bench.py --no-cpu-baseline --steps 15 --warmup 3 --reps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('NSDP_X3_DBG=$v', d['ms_per_step'], 'loss', d['final_loss'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done; done
