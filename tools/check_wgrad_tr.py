"""bf16 weight gradient: the transpose-read kernel against the transposing one (nsdp_debug_set(7, 8)) -- results and time.
    python tools/check_wgrad_tr.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
ok = True
for M, N, K, mask, relu_x in [(1835008, 200, 200, False, False), (1835008, 200, 200, True, False), (320000, 256, 256, False, False),
                              (320000, 256, 256, True, True), (262144, 128, 128, False, True), (655360, 120, 120, True, False),
                              (51200, 256, 256, False, False), (3200, 256, 256, False, False), (70001, 200, 128, True, True),
                              (4099, 120, 256, False, False), (262144, 128, 200, False, False), (65, 64, 64, True, False)]:
    dy = torch.randn(M, N, device=DEV).to(BF); x = torch.randn(M, K, device=DEV).to(BF)
    mk = torch.randn(M, N, device=DEV).to(BF) if mask else None
    run = lambda: hb.wgrad(dy, x, mk, relu_x, True)
    L.nsdp_debug_set(7, 8); dw0, db0 = run(); t0 = t(run)
    L.nsdp_debug_set(7, 0); dw1, db1 = run(); torch.cuda.synchronize(); t1 = t(run)
    dyr = dy.double() * (mk > 0) if mask else dy.double()
    xr = torch.relu(x.double()) if relu_x else x.double()
    ref_w, ref_b = dyr.t() @ xr, dyr.sum(0)
    ew = float((dw1.double() - ref_w).abs().max() / ref_w.abs().max()); eb = float((db1.double() - ref_b).abs().max() / ref_b.abs().max())
    same = bool(torch.equal(dw0, dw1))
    good = ew < 1e-5 and eb < 1e-5
    ok &= good
    print(f"{M:8d} x ({N:3d},{K:3d}){' mask' if mask else '     '}{' relu' if relu_x else '     '}: old {t0:7.1f} us  tr {t1:7.1f} us ({t0/t1:4.2f}x)  "
          f"err dW {ew:.1e} db {eb:.1e}  dW bit-identical to old: {same}  {'ok' if good else 'WRONG'}")
    del dy, x, mk
print("ALL OK" if ok else "FAILURES")
