"""Do a weight-gradient kernel (MFMA-bound, persistent, one 450-register wave per SIMD, 156 KB of LDS per workgroup) and an
HBM-bound attention kernel share the chip?  Each alone, then both at once on two streams (the weight gradient leaving
`reserve` compute units free, as the train step's side stream does), at the decoder's B = 32 shapes.
    python tools/bench_complement.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import hip_attention as A, hip_linear as hl

dev = torch.device("cuda:0")
B, n, N, k, d = 32, 8192, 100, 7, 200
M = B * n * k
a = torch.randn(B, n, k, d, device=dev); pos = torch.randn(B, n, k, d, device=dev)
vf = torch.randn(B, N, d, device=dev); idx = torch.randint(0, N, (B, n, k), device=dev).int()
dy = torch.randn(M, d, device=dev); x = torch.randn(M, d, device=dev)
x2 = torch.randn(M, d, device=dev); w = torch.randn(d, d, device=dev) * d ** -0.5
wp = hl.pack_weight_x3(w)[0]
L = hl.lib()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def attn():
    return A.attn_post(a, vf, pos, idx)          # attn_post_fwd: reads 2 [M, d] tensors (2.9 GB)


def gemm():
    return hl._fwd_x3(x2, wp, d, None, None, None, None, False, False)


def wgrad(reserve):
    L.nsdp_debug_set(9, reserve)
    try:
        return hl._wgrad_x3(dy, x, None, False, True)
    finally:
        L.nsdp_debug_set(9, 0)


def both(f1, f2, reps=10):
    for _ in range(2):
        f1(); f2()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    e[0].record()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        e[1].record()
        for _ in range(reps): f1()
        e[2].record()
    with torch.cuda.stream(s2):
        e[3].record()
        for _ in range(reps): f2()
        e[4].record()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e[5].record()
    torch.cuda.synchronize()
    return e[1].elapsed_time(e[2]) / reps, e[3].elapsed_time(e[4]) / reps, e[0].elapsed_time(e[5]) / reps


def alone(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    t.record(); torch.cuda.synchronize()
    return s.elapsed_time(t) / reps


ta, tg = alone(attn), alone(gemm)
print(f"alone: attn_post_fwd {ta * 1e3:.0f} us, x3 GEMM 1.8M x 200 x 200 {tg * 1e3:.0f} us")
for reserve in (0, 48, 96, 128):
    tw = alone(lambda: wgrad(reserve))
    a1, w1, tot = both(attn, lambda: wgrad(reserve))
    g1, w2, tot2 = both(gemm, lambda: wgrad(reserve))
    print(f"reserve {reserve:3d}: wgrad alone {tw * 1e3:.0f} us | beside attn: attn {a1 * 1e3:.0f} wgrad {w1 * 1e3:.0f} pair {tot * 1e3:.0f} us "
          f"(serial {1e3 * (ta + tw):.0f}) | beside the GEMM: gemm {g1 * 1e3:.0f} wgrad {w2 * 1e3:.0f} pair {tot2 * 1e3:.0f} us (serial {1e3 * (tg + tw):.0f})")
