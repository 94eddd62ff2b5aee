#!/usr/bin/env python
"""The decoder's anchor-table scatters at the bench shape (B = 32 x 57 344 rows over 100 anchors, d = 200): fp32 one-hot
GEMM scatter against the register-table atomic kernel, and the deterministic stream form of attn_post_bwd against the
LDS-table atomic form.  GPU box only.  Prints us per launch and the SURVEY 8d algorithmic-byte rate."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import hip_attention as ha  # noqa: E402
from nsdp_amd._lib import fptr, iptr, lib, stream_ptr  # noqa: E402

DEV = torch.device("cuda:0")


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
    return e0.elapsed_time(e1) / n * 1e3


def main():
    B, n, k, N, d = 32, 8192, 7, 100, 200
    rows = n * k
    g = torch.Generator().manual_seed(0)
    src = torch.randn(B, rows, d, generator=g).to(DEV)
    xyz = torch.rand(B, n, 3, generator=g).to(DEV)
    anchors = torch.rand(B, N, 3, generator=g).to(DEV)
    from nsdp_amd import pointnet2_utils as pu
    idx = pu.knn(xyz, anchors, k)                       # realistic index distribution
    flat = idx.reshape(B, rows)
    bytes_alg = 4.0 * (B * rows * (d + 1) + B * N * d)
    t_new = timeit(lambda: ha.onehot_scatter(src, flat, N))
    L = lib()
    table = torch.zeros(B, N, d, device=DEV)
    dq = torch.zeros(B, 1, d, device=DEV)

    def regtab():
        L.nsdp_attn_pre_bwd(fptr(src), iptr(idx), B, n, N, k, d, 1, fptr(dq), fptr(table), ctypes.c_void_p(0), stream_ptr())
    t_old = timeit(regtab)
    print(f"scatter [B={B}, rows={rows}, d={d}] -> [{N}, {d}]: one-hot f32 {t_new:.0f} us = {bytes_alg / t_new / 1e6:.2f} TB/s; "
          f"register-table atomics {t_old:.0f} us = {bytes_alg / t_old / 1e6:.2f} TB/s")
    ref = -ha.onehot_scatter(src, flat, N)
    regtab()
    print("   max |one-hot - regtab| / max:", float((ref - table).abs().max() / table.abs().max()))

    # attn_post_bwd: LDS-table atomic form against stream + one-hot scatter
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    a, pos, vf, a_g, v_g, dy = mk(B, n, k, d), mk(B, n, k, d), mk(B, N, d), mk(B, d), mk(B, d), mk(B, n, d)
    for flag in (True, False):
        ha.ONEHOT_SCATTER_F32 = flag

        def run():
            ts = [t.clone().requires_grad_(True) for t in (a, vf, pos, a_g, v_g)]
            y = ha.attn_post(ts[0], ts[1], ts[2], idx, ts[3], ts[4])
            return ts, y
        ts, y = run()
        f = lambda: torch.autograd.grad(y, ts, dy, retain_graph=True)
        t = timeit(f, 5)
        by = 4.0 * B * n * k * d * 4
        print(f"attn_post backward (decoder, B={B}): {'stream + one-hot scatter' if flag else 'LDS-table atomics'} {t:.0f} us "
              f"({by / t / 1e6:.2f} TB/s on 4 [R,d] streams)")


if __name__ == "__main__":
    main()
