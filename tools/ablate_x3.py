#!/usr/bin/env python
"""Phase ablation of linear_bf16x3_kernel through nsdp_debug_set(6, bits): 1 = no weight DMA in the k loop, 8 = no output
stores, 32 = one wave per SIMD forms, 64 = nontemporal stores (results are wrong with 1 / 8: timing only).

    python tools/ablate_x3.py [M N K]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nsdp_amd import _lib, hip_linear as hl

DEV = torch.device("cuda:0")
shapes = [tuple(int(a) for a in sys.argv[1:4])] if len(sys.argv) >= 4 else [(1835008, 200, 200), (320000, 256, 256), (262144, 128, 128)]


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
for M, N, K in shapes:
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    wp, _ = hl.pack_weight_x3(w, True, False)
    print(f"linear_bf16x3 {M} x {K} -> {N}   ({4*M*(N+K)/1e6:.0f} MB, {2*M*N*K/1e9:.0f} GFLOP)")
    x = torch.relu(x)       # (ReLU-sparse operands, as in the step)
    for bits, name in [(0, "full (staged epilogue)"), (8192, "direct epilogue (16 rows x 64 B stores)"), (8, "no stores"), (1, "no weight DMA"), (9, "neither"), (64, "nontemporal stores"), (32, "one wave per SIMD"),
                       (1024, "contiguous X DMA (1 KiB / instr)"), (2048, "contiguous stores (1 KiB / instr)"),
                       (1024 + 2048, "contiguous X DMA + stores"), (1024 + 2048 + 1, "contiguous X + stores, no weight DMA"),
                       (512, "1 of 3 LDS weight reads"), (512 + 1, "1 of 3 LDS reads, no weight DMA"), (512 + 9, "1 of 3 LDS reads, no DMA, no stores")]:
        L.nsdp_debug_set(6, bits)
        us = t(lambda: hl._fwd_x3(x, wp, N, b, None, None, None, False, False))
        print(f"  {name:38s} {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TF")
    L.nsdp_debug_set(6, 0)
    del x
