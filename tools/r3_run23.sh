cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bf16x3_gpu.py tests/test_linear_gpu.py -x -q 2>&1 | grep -E "passed|failed|^E |FAILED" | head -5
python - <<'PY' 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids"
import torch, sys
sys.path.insert(0, '.')
from nsdp_amd import _lib, hip_linear as hl
DEV = torch.device("cuda:0")
L = _lib.lib()
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (51200, 40000, 80000, 102400, 320000):
    N = K = 256
    x = torch.relu(torch.randn(M, K, device=DEV)); w = torch.randn(N, K, device=DEV) / K ** 0.5; b = torch.randn(N, device=DEV)
    wp, _ = hl.pack_weight_x3(w, True, False)
    out = []
    for bits in (0, 4096):
        L.nsdp_debug_set(6, bits)
        out.append((t(lambda: hl._fwd_x3(x, wp, N, b, None, None, None, False, True)), hl._fwd_x3(x, wp, N, b, None, None, None, False, True)))
    L.nsdp_debug_set(6, 0)
    print(f"{M} x 256 x 256: chosen {out[0][0]:7.1f} us   three-row-tile form {out[1][0]:7.1f} us   identical {torch.equal(out[0][1], out[1][1])}")
PY
for rep in 1 2; do for v in 0 4096; do NSDP_X3_DBG=$v python bench.py --no-cpu-baseline --steps 20 --warmup 3 --reps 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('X3_DBG=$v', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done; done
