cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_bf16x3_gpu.py tests/test_linear_gpu.py "tests/test_model_gpu.py" -x -q > gpurun_out/r3/tests8.txt 2>&1; tail -6 gpurun_out/r3/tests8.txt
python - <<'PY' 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"
import torch, sys
sys.path.insert(0, '.')
from nsdp_amd import _lib, hip_linear as hl
DEV = torch.device("cuda:0")
L = _lib.lib()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in [(1835008, 200, 200), (320000, 256, 256), (655360, 120, 120), (262144, 200, 128)]:
    x = torch.relu(torch.randn(M, K, device=DEV)); w = torch.randn(N, K, device=DEV) / K ** 0.5; b = torch.randn(N, device=DEV)
    wp, _ = hl.pack_weight_x3(w, True, False)
    for bits, name in [(0, "lazy epilogue (default)"), (2048, "eager epilogue"), (2048 + 8, "eager, no stores"), (8, "lazy build, no stores")]:
        L.nsdp_debug_set(6, bits)
        us = t(lambda: hl._fwd_x3(x, wp, N, b, None, None, None, False, True))
        print(f"{M} x {K} -> {N}  {name:26s} {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TF  {4*(M*(N+K))/us/1e6:5.2f} TB/s")
    L.nsdp_debug_set(6, 0)
PY
for rep in 1 2; do for v in 0 2048; do NSDP_X3_DBG=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('NSDP_X3_DBG=$v', d['ms_per_step'], 'loss', d['final_loss'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done; done
