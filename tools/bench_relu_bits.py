"""dX launches of the bf16x3 kernel with the ReLU mask as an fp32 tensor (PRE = 1) against the mask as bits (PRE = 3) and
against no mask at all, on the masked shapes of the train step.   python tools/bench_relu_bits.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nsdp_amd import hip_linear as hl
DEV = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K, om in [(1835008, 200, 200, False), (320000, 256, 256, False), (262144, 128, 128, True), (458752, 200, 200, False), (655360, 120, 120, False)]:
    x = torch.randn(M, K, device=DEV); w = torch.randn(N, K, device=DEV) / K ** 0.5; b = torch.randn(N, device=DEV)
    dy = torch.randn(M, N, device=DEV)
    wp, wpt = hl.pack_weight_x3(w, True, True)
    bits = hl.relu_bits(M, N, DEV)
    f_plain = t(lambda: hl._fwd_x3(x, wp, N, b, None, None, None, False, True))
    f_bits = t(lambda: hl._fwd_x3(x, wp, N, b, None, None, None, False, True, None, bits))
    y = hl._fwd_x3(x, wp, N, b, None, None, None, False, True, None, bits)
    o = x if om else None
    d_none = t(lambda: hl._fwd_x3(dy, wpt, K, None, None, None, o, False, False))
    d_mask = t(lambda: hl._fwd_x3(dy, wpt, K, None, None, y, o, False, False))
    d_bits = t(lambda: hl._fwd_x3(dy, wpt, K, None, None, None, o, False, False, bits))
    print(f"{M:8d} x {K:3d} -> {N:3d}{' +out_mask' if om else ''}: forward {f_plain:7.1f} us, writing bits {f_bits:7.1f};  dX unmasked {d_none:7.1f}, fp32 mask {d_mask:7.1f}, bit mask {d_bits:7.1f}")
    del x, dy, y
