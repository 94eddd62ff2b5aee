"""attn_post_fwd (2 reads) and attn_pre_fwd (1 read, 1 write) at the same bytes for d = 200 / 256 / 192 / 128 / 64 / 120 channels:
what idle lanes and misaligned rows cost (docs/EXPERIMENTS.md, round 6).    python tools/bench_attention_d.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import hip_attention as A
dev = torch.device("cuda:0")
def t(fn,reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/reps*1e-3
for d, n in [(200, 8192), (256, 6400), (192, 8528), (128, 12800), (64, 25600), (120, 13648)]:
    B, N, k = 32, 100, 7
    a = torch.randn(B, n, k, d, device=dev); pos = torch.randn(B, n, k, d, device=dev); vf = torch.randn(B, N, d, device=dev)
    idx = torch.randint(0, N, (B, n, k), device=dev).int()
    R = a.numel() * 4 / 1e12
    x = t(lambda: A.attn_post(a, vf, pos, idx))
    q = torch.randn(B, n, d, device=dev); kf = torch.randn(B, N, d, device=dev)
    x2 = t(lambda: A.attn_pre(q, kf, pos, idx))
    print(f"d={d} n={n}: [rows,d] {R*1e3:.2f} GB; post fwd {x*1e6:.0f} us = {2*R/x:.2f} TB/s (2R); pre fwd {x2*1e6:.0f} us = {2*R/x2:.2f} TB/s (1R1W)")
    del a, pos
