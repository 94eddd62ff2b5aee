#!/usr/bin/env python
"""bf16x3 forward GEMM on the step's large shapes, isolated (us per launch): python tools/bench_x3_shapes.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsdp_amd.hip_linear import _fwd_x3, pack_weight_x3  # noqa: E402

dev = torch.device("cuda:0")
for M, K, N in ((1_048_576, 200, 200), (327_680, 256, 256), (262_144, 128, 128), (65_536, 200, 200)):
    x = torch.randn(M, K, device=dev).relu_()
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    w3 = pack_weight_x3(w)[0]
    for _ in range(5):
        _fwd_x3(x, w3, N, b, None, None, None, False, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        _fwd_x3(x, w3, N, b, None, None, None, False, False)
    e1.record()
    torch.cuda.synchronize()
    print(M, K, N, f"{e0.elapsed_time(e1) / 30 * 1e3:.1f} us")
