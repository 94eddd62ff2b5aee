"""The anti-phase bf16x3 kernel (nsdp_debug_set(6, 128)) against the standard one and fp64; timing of both.
    python tools/check_x3_ap.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from nsdp_amd import _lib, hip_linear as hl

DEV = torch.device("cuda:0")
L = _lib.lib()


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = [  # M, K, N, bias, residual, out_mask, relu_in, relu_out
    (524288 + 128, 200, 200, True, False, False, False, True),
    (600001, 200, 200, True, True, False, True, False),
    (524288 + 129 * 7, 64, 144, False, False, True, False, False),
    (700000, 256, 160, True, False, False, False, False),
    (655360, 120, 200, True, False, False, True, True),
    (1835008, 200, 200, True, False, False, False, False),
]
g = torch.Generator(device="cpu").manual_seed(1)
ok = True
for M, K, N, bias, res, omask, relu_in, relu_out in cases:
    x = torch.randn(M, K, device=DEV)
    if relu_out or relu_in:
        pass
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV) if bias else None
    r = torch.randn(M, N, device=DEV) if res else None
    o = torch.randn(M, N, device=DEV) if omask else None
    wp = hl.pack_weight_x3(w)[0]
    run = lambda: hl._fwd_x3(x, wp, N, b, r, None, o, relu_in, relu_out)
    L.nsdp_debug_set(6, 0)
    y0 = run()
    t0 = t(run)
    for bits in (128, 128 | 0x300):
        L.nsdp_debug_set(6, bits)
        y1 = run()
        torch.cuda.synchronize()
        t1 = t(run)
        # fp64 reference on a sample of rows
        idx = torch.cat([torch.arange(0, 300, device=DEV), torch.arange(M - 300, M, device=DEV),
                         torch.randint(0, M, (2000,), device=DEV)])
        xi = x[idx].double()
        if relu_in:
            xi = F.relu(xi)
        ref = xi @ w.double().t()
        if b is not None:
            ref = ref + b.double()
        if r is not None:
            ref = ref + r[idx].double()
        if relu_out:
            ref = F.relu(ref)
        if o is not None:
            ref = ref * (o[idx] > 0)
        scale = float(ref.abs().max())
        e1 = float((y1[idx].double() - ref).abs().max()) / scale
        e0 = float((y0[idx].double() - ref).abs().max()) / scale
        d01 = float((y1 - y0).abs().max()) / scale
        good = e1 <= 1.5e-6 and d01 <= 3e-6
        ok &= good
        print(f"{M:8d} x {K:3d} -> {N:3d} dbg {bits:4d}: std {t0:7.1f} us  ap {t1:7.1f} us  ({t0 / t1:4.2f}x)  err vs fp64 {e1:.2e} (std {e0:.2e})  "
              f"max |ap - std| {d01:.2e}  {'ok' if good else 'WRONG'}")
    L.nsdp_debug_set(6, 0)
    del x, r, o
print("ALL OK" if ok else "FAILURES")
