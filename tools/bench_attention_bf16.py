"""tools/bench_attention.py for the bf16-storage instantiation of the attention glue kernels (backward only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import hip_attention as A
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
SHAPES = [(32, 100, 100, 100, 256, False), (32, 100, 500, 16, 256, False), (32, 500, 2048, 16, 120, False),
          (32, 500, 500, 16, 120, False), (32, 2048, 2048, 10, 120, False), (32, 8192, 100, 7, 200, True)]
for (B, n, N, k, d, per_shape) in SHAPES:
    mk = lambda *s: torch.randn(*s, device=dev).to(BF)
    q, kf, vf, pos, a, du, dy = mk(B, 1 if per_shape else n, d), mk(B, N, d), mk(B, N, d), mk(B, n, k, d), mk(B, n, k, d), mk(B, n, k, d), mk(B, n, d)
    idx = torch.randint(0, N, (B, n, k), device=dev).int()
    R = B * n * k * d * 2 / 1e9
    qq, kk, pp = (t.clone().requires_grad_(True) for t in (q, kf, pos))
    link = A.pos_grad_link(); acc = torch.zeros_like(pos); u2 = A.attn_pre(qq, kk, pp, idx, link)
    def pre_b_acc():
        link.dpos = acc
        return torch.autograd.grad(u2, [qq, kk, pp], du, retain_graph=True)
    t_pre = timeit(pre_b_acc)
    aa, vv, pp2 = (t.clone().requires_grad_(True) for t in (a, vf, pos))
    y = A.attn_post(aa, vv, pp2, idx)
    t_post = timeit(lambda: torch.autograd.grad(y, [aa, vv, pp2], dy, retain_graph=True))
    print(f"bf16 B={B} n={n} N={N} k={k} d={d}{' q/shape' if per_shape else ''}: [rows,d] = {R*1e3:.0f} MB | pre bwd+dpos {t_pre*1e3:.0f} us "
          f"({3*R/t_pre:.2f} TB/s) | post bwd {t_post*1e3:.0f} us ({4*R/t_post:.2f} TB/s)")
