import os, sys
sys.path.insert(0, "/root/repo")
import torch
from nsdp_amd import _lib, hip_linear_bf16 as hb
DEV = torch.device("cuda:0"); BF = torch.bfloat16
L = _lib.lib()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K in [(1835008, 200, 200), (320000, 256, 256), (262144, 128, 128), (655360, 120, 120)]:
    dy = torch.randn(M, N, device=DEV).to(BF); x = torch.randn(M, K, device=DEV).to(BF)
    row = []
    for dbg, name in [(0, "full"), (1, "no A reads / MFMA"), (4, "no DMA"), (5, "loop + B reads")]:
        L.nsdp_debug_set(7, dbg)
        row.append(f"{name} {t(lambda: hb.wgrad(dy, x, None, False, True)):7.1f}")
    L.nsdp_debug_set(7, 0)
    print(f"{M} x ({N},{K}): " + "  ".join(row) + f"   [{2*M*(N+K)/1e6:.0f} MB]")
