"""fp32 (bf16x3) weight gradient: the row-wise producer + transpose-read kernel against the column-wise one
(nsdp_debug_set(9, 0)) -- results against fp64 and time.     python tools/check_wgrad_x3_rows.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
ok = True
for M, N, K, mask, relu_x in [(1835008, 200, 200, False, False), (1835008, 200, 200, True, False), (320000, 256, 256, False, False),
                              (320000, 256, 256, True, True), (262144, 128, 128, False, True), (655360, 120, 120, True, False),
                              (51200, 256, 256, False, False), (3200, 256, 256, False, False), (70001, 200, 128, True, True),
                              (4099, 120, 256, False, False), (262144, 128, 200, False, False), (2049, 36, 64, True, False)]:
    dy = torch.randn(M, N, device=DEV); x = torch.randn(M, K, device=DEV)
    mk = torch.randn(M, N, device=DEV) if mask else None
    run = lambda: hl._wgrad_x3(dy, x, mk, relu_x, True)
    L.nsdp_debug_set(9, 0); dw0, db0 = run(); t0 = t(run)
    L.nsdp_debug_set(9, 1); dw1, db1 = run(); torch.cuda.synchronize(); t1 = t(run)
    idx = torch.randperm(M, device=DEV)[:min(M, 200000)] if M > 400000 else None
    dyr = dy.double() * (mk > 0) if mask else dy.double()
    xr = torch.relu(x.double()) if relu_x else x.double()
    ref_w, ref_b = dyr.t() @ xr, dyr.sum(0)
    sw, sb = float(ref_w.abs().max()), float(ref_b.abs().max())
    ew1, ew0 = float((dw1.double() - ref_w).abs().max()) / sw, float((dw0.double() - ref_w).abs().max()) / sw
    eb1 = float((db1.double() - ref_b).abs().max()) / sb
    good = ew1 <= max(2.0 * ew0, 3e-7) and eb1 < 2e-6
    ok &= good
    print(f"{M:8d} x ({N:3d},{K:3d}){' mask' if mask else '     '}{' relu' if relu_x else '     '}: columns {t0:7.1f} us  rows {t1:7.1f} us ({t0/t1:4.2f}x)  "
          f"err dW {ew1:.1e} (columns {ew0:.1e}) db {eb1:.1e}  {'ok' if good else 'WRONG'}")
    del dy, x, mk
L.nsdp_debug_set(9, 1)
print("ALL OK" if ok else "FAILURES")
