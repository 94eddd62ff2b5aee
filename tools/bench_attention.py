"""The fused attention glue kernels on the shapes of the B=32 train step, forward and backward, with the bytes each
launch has to move (GB/s against that)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import hip_attention as A
dev = torch.device("cuda:0")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
# (B, n centres, N sources, k, d, one query per shape)
SHAPES = [(32, 100, 100, 100, 256, False), (32, 100, 500, 16, 256, False), (32, 500, 2048, 16, 120, False),
          (32, 500, 500, 16, 120, False), (32, 2048, 2048, 10, 120, False), (32, 8192, 100, 7, 200, True)]
for (B, n, N, k, d, per_shape) in SHAPES:
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, 1 if per_shape else n, d, device=dev); kf = torch.randn(B, N, d, device=dev); vf = torch.randn(B, N, d, device=dev)
    pos = torch.randn(B, n, k, d, device=dev); a = torch.randn(B, n, k, d, device=dev)
    idx = torch.randint(0, N, (B, n, k), device=dev).int()
    du = torch.randn(B, n, k, d, device=dev); dy = torch.randn(B, n, d, device=dev)
    R = B * n * k * d * 4 / 1e9            # one [rows, d] tensor in GB
    qq, kk, pp = (t.clone().requires_grad_(True) for t in (q, kf, pos))
    u = A.attn_pre(qq, kk, pp, idx)
    t_pre_f = timeit(lambda: A.attn_pre(q, kf, pos, idx))
    t_pre_b = timeit(lambda: torch.autograd.grad(u, [qq, kk, pp], du, retain_graph=True))
    link = A.pos_grad_link(); link.dpos = torch.zeros_like(pos)
    def pre_b_acc():
        link.dpos = acc
        return torch.autograd.grad(u2, [qq, kk, pp], du, retain_graph=True)
    acc = torch.zeros_like(pos); u2 = A.attn_pre(qq, kk, pp, idx, link)
    t_pre_ba = timeit(pre_b_acc)
    aa, vv, pp2 = (t.clone().requires_grad_(True) for t in (a, vf, pos))
    y = A.attn_post(aa, vv, pp2, idx)
    t_post_f = timeit(lambda: A.attn_post(a, vf, pos, idx))
    t_post_b = timeit(lambda: torch.autograd.grad(y, [aa, vv, pp2], dy, retain_graph=True))
    print(f"B={B} n={n} N={N} k={k} d={d}{' q/shape' if per_shape else ''}: [rows,d] = {R*1e3:.0f} MB | pre fwd {t_pre_f*1e3:.0f} us ({2*R/t_pre_f*1e3/1e3:.2f} TB/s) "
          f"bwd {t_pre_b*1e3:.0f} us ({R/t_pre_b:.2f} TB/s) bwd+dpos {t_pre_ba*1e3:.0f} us ({3*R/t_pre_ba:.2f} TB/s) | "
          f"post fwd {t_post_f*1e3:.0f} us ({2*R/t_post_f:.2f} TB/s) bwd {t_post_b*1e3:.0f} us ({4*R/t_post_b:.2f} TB/s)")
