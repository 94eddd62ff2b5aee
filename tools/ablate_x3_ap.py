import os, sys
sys.path.insert(0, "/root/repo")
import torch
from nsdp_amd import _lib, hip_linear as hl
DEV = torch.device("cuda:0")
L = _lib.lib()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M, N, K = 1835008, 200, 200
for sparse in (False, True):
    x = torch.randn(M, K, device=DEV)
    if sparse: x = torch.relu(x)
    w = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    wp, _ = hl.pack_weight_x3(w, True, False)
    for base, name in ((0, "std"), (128, "ap ")):
        row = []
        for bits, nm in ((0, "full"), (8, "no stores"), (1, "no W DMA"), (9, "neither")):
            L.nsdp_debug_set(6, base | bits)
            row.append(f"{nm} {t(lambda: hl._fwd_x3(x, wp, N, b, None, None, None, False, False)):7.1f}")
        print(("relu'd " if sparse else "dense  ") + name, "  ".join(row))
L.nsdp_debug_set(6, 0)
