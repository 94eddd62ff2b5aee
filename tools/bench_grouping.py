#!/usr/bin/env python
"""HBM rate of the standalone grouping / sampling kernels (gather_rows = index_points, scatter_add_rows, group_points,
kNN, FPS) at the sizes of the train step and at a size large enough to leave the launch-latency regime.

    python tools/bench_grouping.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nsdp_amd import pointnet2_utils as pu

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


for (B, N, C, S, what) in [(32, 2048, 120, 500, "step: FPS centres of level 1"), (32, 2048, 120, 8000, "step: 500 x 16 neighbours"),
                           (32, 100, 200, 57344, "step: decoder anchors, 8192 x 7"), (32, 8192, 128, 262144, "large"),
                           (8, 100000, 256, 400000, "large, wide rows")]:
    pts = torch.randn(B, N, C, device=dev)
    idx = torch.randint(0, N, (B, S), device=dev, dtype=torch.int32)
    t = timeit(lambda: pu.gather_rows(pts, idx))
    out_b = B * S * C * 4
    print(f"gather_rows      B={B} N={N} C={C} S={S:7d} ({what}): {t*1e6:8.1f} us  {2*out_b/t/1e9:7.1f} GB/s (gathered read + write)")
    g = torch.randn(B, S, C, device=dev)
    t = timeit(lambda: pu.scatter_add_rows(g, idx, N))
    print(f"scatter_add_rows same shape: {t*1e6:8.1f} us  {out_b/t/1e9:7.1f} GB/s (read; atomics into [B,N,C])")
