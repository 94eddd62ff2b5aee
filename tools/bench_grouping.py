#!/usr/bin/env python
"""HBM rate of the standalone grouping / sampling kernels (gather_rows = index_points, scatter_add_rows, group_points,
kNN, FPS) at the sizes of the train step and at a size large enough to leave the launch-latency regime.

Every GB/s figure is ALGORITHMIC bytes / time (index + source table ONCE + output; for a gradient
output <-> grad) and every fraction is against the 8 TB/s HBM peak -- gathered re-reads hit L2 and are not counted, so no
figure can exceed the peak.  The search kernels (ball query, 3-NN, kNN) move almost no bytes: they are reported as
distance tests / s against the fp32 VALU peak (256 CUs x 64 lanes x 2.4 GHz = 39.3 T lane-ops/s; a test is 8 unfused
fp32 operations -- three subtractions, three multiplications, two additions, one rounding each as the reference's
(dx*dx + dy*dy) + dz*dz demands -- plus the compare / select of the selection, ~12 lane-ops: ~3.3 T tests/s).

    python tools/bench_grouping.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import ctypes

from nsdp_amd import pointnet2_utils as pu
from nsdp_amd._lib import fptr, iptr, lib, stream_ptr

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
    alg = 4.0 * (B * S + B * N * C + B * S * C)          # index + table once + output
    print(f"gather_rows      B={B} N={N} C={C} S={S:7d} ({what}): {t*1e6:8.1f} us  {alg/t/1e9:7.1f} GB/s = {alg/t/8e12:.2f} of HBM peak")
    g = torch.randn(B, S, C, device=dev)
    pu._SCATTER_INVERSE = False
    t = timeit(lambda: pu.scatter_add_rows(g, idx, N))
    print(f"scatter_add_rows same shape, global atomics: {t*1e6:8.1f} us  {alg/t/1e9:7.1f} GB/s = {alg/t/8e12:.2f}")
    pu._SCATTER_INVERSE = True
    if C % 4 == 0 and N <= 8192:
        t = timeit(lambda: pu.scatter_add_rows(g, idx, N))
        def cold():
            idx.__dict__.pop("_nsdp_inverse", None)
            return pu.scatter_add_rows(g, idx, N)
        tc = timeit(cold)
        print(f"scatter_add_rows same shape, inverse lists + segment sum: {t*1e6:8.1f} us  {alg/t/1e9:7.1f} GB/s = {alg/t/8e12:.2f} "
              f"with the lists cached on the index tensor; {tc*1e6:8.1f} us = {alg/tc/8e12:.2f} including the list build")
    if N <= 128 and C % 4 == 0 and 16 < C <= 208:
        from nsdp_amd import hip_attention as ha
        t = timeit(lambda: ha.onehot_scatter(g, idx, N))
        print(f"scatter as a GEMM (one-hot, fp32, deterministic) same shape: {t*1e6:8.1f} us  {alg/t/1e9:7.1f} GB/s = {alg/t/8e12:.2f}")

# ---- the channel-major `pointnet2_ops._ext` operators (reference group_points_gpu.cu:8-64, sampling_gpu.cu:8-57,
# ball_query_gpu.cu:9-44, interpolate_gpu.cu:9-141): achieved GB/s against their ALGORITHMIC bytes (SURVEY.md section 8d:
# output bytes + index bytes + one read of the source; the gathered source reads hit L2 and are not counted twice)
print()
for (B, C, N, NP, NS, what) in [(32, 128, 2048, 500, 16, "SA level 1 shape"), (32, 256, 500, 100, 16, "SA level 2 shape"),
                                (32, 64, 8192, 2048, 32, "PointNet++ SSG first level"), (16, 128, 16384, 4096, 32, "large")]:
    feats = torch.randn(B, C, N, device=dev, requires_grad=True)
    idx = torch.randint(0, N, (B, NP, NS), device=dev, dtype=torch.int32)
    t = timeit(lambda: pu.grouping_operation(feats.detach(), idx))
    alg = 4.0 * (B * C * NP * NS + B * NP * NS + B * C * N)
    print(f"group_points      B={B} C={C} N={N} np={NP} ns={NS} ({what}): {t*1e6:8.1f} us  {alg/t/1e9:7.1f} GB/s = {alg/t/8e12:.2f} of HBM peak")
    g = torch.randn(B, C, NP * NS, device=dev)
    idx2 = idx.view(B, NP * NS)
    # (the kernels behind GroupingOperation.backward, called directly: torch.autograd.grad alone costs ~50 us of Python)
    t = timeit(lambda: pu._scatter_cm(g, idx2, N, idx_obj=idx))
    def cold():
        idx.__dict__.pop("_nsdp_inverse", None)
        return pu._scatter_cm(g, idx2, N, idx_obj=idx)
    tc = timeit(cold)
    gp = torch.empty(B, C, N, device=dev)
    ta = timeit(lambda: lib().nsdp_group_points_grad(fptr(g), iptr(idx), ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int(N),
                                                     ctypes.c_int(NP), ctypes.c_int(NS), fptr(gp), stream_ptr()))
    if pu._scatter_cm(g, idx2, N, idx_obj=idx) is None:
        print(f"group_points_grad same shape: LDS-table form {ta*1e6:.1f} us = {alg/ta/8e12:.2f} (no inverse lists for this shape)")
    else:
        print(f"group_points_grad same shape: {t*1e6:8.1f} us  {alg/t/1e9:7.1f} GB/s = {alg/t/8e12:.2f} (inverse lists cached on the "
              f"index tensor; {tc*1e6:.1f} us = {alg/tc/8e12:.2f} including the list build; LDS-table form {ta*1e6:.1f} us = {alg/ta/8e12:.2f})")
    idx1 = torch.randint(0, N, (B, NP), device=dev, dtype=torch.int32)
    t = timeit(lambda: pu.gather_operation(feats.detach(), idx1))
    alg1 = 4.0 * (B * C * NP + B * NP + min(B * C * N, B * C * NP * 16))      # (a gather of np columns touches <= np 64-B sectors per row)
    print(f"gather_points     B={B} C={C} N={N} np={NP}: {t*1e6:8.1f} us  {alg1/t/1e9:7.1f} GB/s = {alg1/t/8e12:.2f}")
    g1 = torch.randn(B, C, NP, device=dev)
    t = timeit(lambda: pu._scatter_cm(g1, idx1, N))
    alg1g = 4.0 * (B * C * NP + B * NP + B * C * N)                           # (the gradient tensor is zero-filled and written once)
    if pu._scatter_cm(g1, idx1, N) is None:          # (shapes outside the list kernels: the LDS-table / atomic entry point)
        gp1 = torch.empty(B, C, N, device=dev)
        t = timeit(lambda: lib().nsdp_gather_points_grad(fptr(g1), iptr(idx1), ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int(N),
                                                         ctypes.c_int(NP), fptr(gp1), stream_ptr()))
        print(f"gather_points_grad same shape (LDS-table form): {t*1e6:8.1f} us  {alg1g/t/1e9:7.1f} GB/s = {alg1g/t/8e12:.2f}")
    else:
        print(f"gather_points_grad same shape (inverse lists, cached): {t*1e6:8.1f} us  {alg1g/t/1e9:7.1f} GB/s = {alg1g/t/8e12:.2f}")

print()
for (B, N, M, ns, r, what) in [(32, 2048, 500, 16, 0.2, "SA level 1"), (32, 8192, 2048, 32, 0.1, "SSG first level"), (16, 16384, 4096, 32, 0.08, "large")]:
    xyz = torch.rand(B, N, 3, device=dev) - 0.5
    new_xyz = xyz[:, :M].contiguous()
    t = timeit(lambda: pu.ball_query(r, ns, xyz, new_xyz))
    alg = 4.0 * (B * M * ns + 3 * B * (N + M))
    VALU_TESTS = 256 * 64 * 2.4e9 / 12          # distance tests / s the fp32 VALUs could do (12 lane-ops per test)
    print(f"ball_query        B={B} N={N} M={M} ns={ns} r={r} ({what}): {t*1e6:8.1f} us  {B*M*N/t/1e9:7.1f} G distance tests/s "
          f"= {B*M*N/t/VALU_TESTS:.2f} of the VALU peak ({alg/t/1e9:.1f} GB/s of algorithmic bytes: not HBM-bound)")
    t = timeit(lambda: pu.three_nn(new_xyz, xyz))
    alg = 4.0 * (6 * B * M + 3 * B * (N + M))
    print(f"three_nn          B={B} n={M} m={N}: {t*1e6:8.1f} us  {B*M*N/t/1e9:7.1f} G distance tests/s = {B*M*N/t/VALU_TESTS:.2f} "
          f"of the VALU peak")
    t = timeit(lambda: pu.knn(new_xyz, xyz, 16))
    print(f"knn (k = 16)      B={B} n={M} m={N}: {t*1e6:8.1f} us  {B*M*N/t/1e9:7.1f} G distance tests/s = {B*M*N/t/VALU_TESTS:.2f} "
          f"of the VALU peak")

print()
for (B, C, M, N, what) in [(32, 256, 100, 500, "FP level 2 -> 1"), (32, 128, 500, 2048, "FP level 1 -> 0"), (16, 128, 4096, 16384, "large")]:
    feats = torch.randn(B, C, M, device=dev, requires_grad=True)
    idx = torch.randint(0, M, (B, N, 3), device=dev, dtype=torch.int32)
    w = torch.rand(B, N, 3, device=dev)
    t = timeit(lambda: pu.three_interpolate(feats.detach(), idx, w))
    alg = 4.0 * (B * C * N + 6 * B * N + B * C * M)
    print(f"three_interpolate B={B} c={C} m={M} n={N} ({what}): {t*1e6:8.1f} us  {alg/t/1e9:7.1f} GB/s = {alg/t/8e12:.2f}")
    o = pu.three_interpolate(feats, idx, w)
    g = torch.randn_like(o)
    # (the kernels behind ThreeInterpolate.backward, called directly: torch.autograd.grad alone costs ~50 us of Python)
    from nsdp_amd import hip_attention as ha
    gp3 = torch.empty(B, C, M, device=dev)
    if lib().nsdp_three_interpolate_grad_lists_supported(ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int(N), ctypes.c_int(M)):
        off, ent = ha.inverse_lists(idx, M)
        t = timeit(lambda: lib().nsdp_three_interpolate_grad_lists(fptr(g), fptr(w), iptr(off), iptr(ent), ctypes.c_int(B), ctypes.c_int(C),
                                                                   ctypes.c_int(N), ctypes.c_int(M), fptr(gp3), stream_ptr()))
        def cold():
            idx.__dict__.pop("_nsdp_inverse", None)
            o2, e2 = ha.inverse_lists(idx, M)
            lib().nsdp_three_interpolate_grad_lists(fptr(g), fptr(w), iptr(o2), iptr(e2), ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int(N),
                                                    ctypes.c_int(M), fptr(gp3), stream_ptr())
        tc = timeit(cold)
        print(f"three_interpolate_grad same shape, inverse lists (cached): {t*1e6:8.1f} us  {alg/t/1e9:7.1f} GB/s = {alg/t/8e12:.2f}; "
              f"{tc*1e6:.1f} us = {alg/tc/8e12:.2f} including the list build")
    ta = timeit(lambda: lib().nsdp_three_interpolate_grad(fptr(g), iptr(idx), fptr(w), ctypes.c_int(B), ctypes.c_int(C), ctypes.c_int(N),
                                                          ctypes.c_int(M), fptr(gp3), stream_ptr()))
    print(f"three_interpolate_grad same shape, LDS-table / atomic form: {ta*1e6:8.1f} us  {alg/ta/1e9:7.1f} GB/s = {alg/ta/8e12:.2f}")
