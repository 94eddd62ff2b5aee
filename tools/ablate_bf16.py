#!/usr/bin/env python
"""Phase ablation of the bf16-storage dense kernels (nsdp_debug_set keys 7 / 8: results are wrong, timing only).

    python tools/ablate_bf16.py [M N K]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nsdp_amd import _lib, hip_linear_bf16 as hb

DEV = torch.device("cuda:0")
BF = torch.bfloat16
M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1835008, 200, 200)
dy = torch.randn(M, N, device=DEV).to(BF)
x = torch.randn(M, K, device=DEV).to(BF)
w = torch.randn(N, K, device=DEV) / K ** 0.5
wp, _ = hb.pack_weight_b16(w, True, False)


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


L = _lib.lib()
print(f"linear {M} x {K} -> {N}")
for dbg, name in [(0, "full"), (1, "no mfma"), (2, "no stores"), (3, "loads only")]:
    L.nsdp_debug_set(8, dbg)
    print(f"  {name:12s} {t(lambda: hb.run(x, wp, N, None, None, None, None, False, False)):8.1f} us")
L.nsdp_debug_set(8, 0)
print(f"wgrad {M} x ({N}, {K})")
for dbg, name in [(0, "full"), (1, "no mfma"), (2, "no transpose"), (3, "dma only"), (4, "no dma"), (6, "mfma only"), (7, "loop only")]:
    L.nsdp_debug_set(7, dbg)
    print(f"  {name:12s} {t(lambda: hb.wgrad(dy, x, None, False, True)):8.1f} us")
L.nsdp_debug_set(7, 0)
