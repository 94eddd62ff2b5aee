"""Does kernel time depend on the operand VALUES?  (It does: MI355X clocks follow switching power.)
Times the bf16x3 forward GEMM and weight-gradient kernels on the same shape with dense random operands, with half
of the entries zero (what ReLU networks feed them), and with all-zero operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import hip_linear
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
M, N, K = 1835008, 200, 200
w = torch.randn(N, K, device=dev) * 0.1
b = torch.zeros(N, device=dev)
pack = hip_linear.pack_weight_x3(w)[0]
for name, f in [("dense randn", lambda t: t), ("half zeros", lambda t: t * (torch.rand_like(t) > 0.5)), ("all zeros", lambda t: t * 0)]:
    x = f(torch.randn(M, K, device=dev)); dy = f(torch.randn(M, N, device=dev))
    tf = timeit(lambda: hip_linear._fwd_x3(x, pack, N, b, None, None, None, False, False))
    tw = timeit(lambda: hip_linear._wgrad_x3(dy, x, None, False, True))
    print(f"{name:12s}: forward {tf:.3f} ms   wgrad {tw:.3f} ms")
