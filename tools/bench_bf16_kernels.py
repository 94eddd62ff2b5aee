#!/usr/bin/env python
"""Isolated timing of the bf16-storage dense kernels on the shapes of the B = 32 train step (GB/s of algorithmic bytes).

    python tools/bench_bf16_kernels.py [lin|wgrad|all]
"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nsdp_amd import hip_linear_bf16 as hb

DEV = torch.device("cuda:0")
BF = torch.bfloat16
SHAPES = [(1835008, 200, 200), (320000, 256, 256), (655360, 120, 120), (262144, 128, 128), (262144, 200, 128),
          (262144, 128, 200), (256000, 120, 120), (51200, 256, 256), (3200, 256, 256)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


what = sys.argv[1] if len(sys.argv) > 1 else "all"
for M, K, N in SHAPES:
    x = torch.randn(M, K, device=DEV).to(BF)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    dy = torch.randn(M, N, device=DEV).to(BF)
    wp, _ = hb.pack_weight_b16(w, True, False)
    if what in ("lin", "all"):
        t = timeit(lambda: hb.run(x, wp, N, b, None, None, None, False, True))
        print(f"linear  {M:8d} x {K:3d} -> {N:3d}          {t*1e3:8.1f} us  {2*M*(K+N)/t/1e6:7.1f} GB/s  {2*M*N*K/t/1e9:6.1f} TF")
        mk = torch.relu(torch.randn(M, K, device=DEV)).to(BF)
        t = timeit(lambda: hb.run(x, wp, N, None, None, mk, None, False, False))
        print(f"linear  {M:8d} x {K:3d} -> {N:3d} mask     {t*1e3:8.1f} us  {2*M*(2*K+N)/t/1e6:7.1f} GB/s")
    if what in ("wgrad", "all"):
        t = timeit(lambda: hb.wgrad(dy, x, None, False, True))
        print(f"wgrad   {M:8d} x ({N:3d} , {K:3d})        {t*1e3:8.1f} us  {2*M*(K+N)/t/1e6:7.1f} GB/s  {2*M*N*K/t/1e9:6.1f} TF")
        mk = torch.relu(torch.randn(M, N, device=DEV)).to(BF)
        t = timeit(lambda: hb.wgrad(dy, x, mk, False, True))
        print(f"wgrad   {M:8d} x ({N:3d} , {K:3d}) mask   {t*1e3:8.1f} us  {2*M*(K+2*N)/t/1e6:7.1f} GB/s")
    del x, dy
