import sys, torch
sys.path.insert(0, "/root/repo")
from nsdp_amd import hip_linear
dev = torch.device("cuda:0")
for M, N, K in ((800, 256, 256), (800, 120, 120), (800, 256, 512), (3200, 256, 256), (2000, 200, 200)):
    dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
    f = lambda: hip_linear._wgrad(dy, x, None, False, True)
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): f()
    b.record(); torch.cuda.synchronize()
    dw, db = f()
    ref = dy.double().t() @ x.double()
    print(M, N, K, f"{a.elapsed_time(b) / 200 * 1e3:.1f} us/call", "err", float((dw.double() - ref).abs().max() / ref.abs().max()))
