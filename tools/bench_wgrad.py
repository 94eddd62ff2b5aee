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
for (M, N, K, mask, rx) in [(4096, 200, 200, False, False), (4099, 120, 128, True, True), (65536, 200, 200, True, False), (1835008, 200, 200, False, False), (1835008, 200, 200, True, False),
                            (262144, 128, 200, False, False), (262144, 128, 128, True, True), (256000, 120, 120, False, False), (655360, 120, 120, False, False)]:
    dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
    m = torch.randn(M, N, device=dev) if mask else None
    dw3, db3 = hip_linear._wgrad_x3(dy, x, m, rx, True)
    hip_linear._USE_X3 = False
    dw0, db0 = hip_linear._wgrad(dy, x, m, rx, True)
    t0 = timeit(lambda: hip_linear._wgrad(dy, x, m, rx, True))
    hip_linear._USE_X3 = True
    t3 = timeit(lambda: hip_linear._wgrad_x3(dy, x, m, rx, True))
    dyp = dy * (m > 0) if mask else dy
    xp = torch.relu(x) if rx else x
    R = min(M, 65536)
    ref = dyp[:R].double().t() @ xp[:R].double() if M <= 65536 else None
    if ref is not None:
        e3 = ((dw3.double() - ref).abs().max() / ref.abs().max()).item(); e0 = ((dw0.double() - ref).abs().max() / ref.abs().max()).item()
    else:
        e3 = ((dw3 - dw0).abs().max() / dw0.abs().max()).item(); e0 = float("nan")
    eb = ((db3 - db0).abs().max() / db0.abs().max()).item()
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} mask={mask} relu_x={rx}: fp32 {t0:.3f} ms {fl/t0/1e9:.1f} TF | bf16x3 {t3:.3f} ms {fl/t3/1e9:.1f} TF | err x3 {e3:.2e} fp32 {e0:.2e} db {eb:.1e}")
