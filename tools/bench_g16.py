"""Row-major against G16 operands, launch by launch, at the shapes of the B = 32 train step (isolated kernels).
    python tools/bench_g16.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import hip_linear as hl

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


print(f"{'shape':22s} {'form':34s} {'row-major us':>12s} {'G16 us':>9s}  ratio")
for M, K, N in [(1835008, 200, 200), (320000, 256, 256), (51200, 256, 256), (655360, 120, 120), (256000, 120, 120), (262144, 128, 128)]:
    x = torch.relu(torch.randn(M, K, device=dev)); w = torch.randn(N, K, device=dev) * K ** -0.5
    wp = hl.pack_weight_x3(w)[0]
    b = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev); mask = torch.relu(torch.randn(M, K, device=dev))
    rows = [
        ("fwd L0: bias + relu -> Y", lambda: hl._fwd_x3(x, wp, N, b, None, None, None, False, True),
         lambda: hl._fwd_x3_g16(x, wp, N, b, None, None, None, False, True, hl.LAY_Y)),
        ("fwd L1: X -> bias", lambda: hl._fwd_x3(x, wp, N, b, None, None, None, False, False),
         lambda: hl._fwd_x3_g16(x, wp, N, b, None, None, None, False, False, hl.LAY_X)),
        ("dX L0: X, mask + residual", lambda: hl._fwd_x3(x, wp, N, None, res, mask, None, False, False),
         lambda: hl._fwd_x3_g16(x, wp, N, None, res, mask, None, False, False, hl.LAY_X)),
    ]
    for name, f0, f1 in rows:
        t0, t1 = timeit(f0), timeit(f1)
        print(f"{M:8d}x{K:3d}x{N:3d}   {name:34s} {t0:12.1f} {t1:9.1f}  {t1 / t0:5.3f}")
    dy = torch.randn(M, N, device=dev)
    for name, lay, mk in [("wgrad: dY + mask G16", 1, True), ("wgrad: X G16", 2, False)]:
        if not hl.lib().nsdp_linear_wgrad_bf16x3_g16_supported(hl._ll(M), N, K, lay, int(mk)):
            continue
        m = torch.relu(torch.randn(M, N, device=dev)) if mk else None
        fn = hl._wgrad_g16_fn(lay)
        t0, t1 = timeit(lambda: hl._wgrad_x3(dy, x, m, False, True)), timeit(lambda: fn(dy, x, m, False, True))
        print(f"{M:8d}x{K:3d}x{N:3d}   {name:34s} {t0:12.1f} {t1:9.1f}  {t1 / t0:5.3f}")
    del x, res, mask, dy
