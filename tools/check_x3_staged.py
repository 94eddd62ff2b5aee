#!/usr/bin/env python
"""Staged against direct epilogue of linear_bf16x3_kernel (nsdp_debug_set(6, 8192)): where do they differ?  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nsdp_amd import _lib, hip_linear as hl

DEV = torch.device("cuda:0")
L = _lib.lib()
for (M, K, N, bias) in [(256 * 256 * 3 + 77, 200, 200, True), (256 * 256 * 2, 200, 200, False), (256 * 192 * 3 + 5, 256, 256, True),
                        (512 * 256 + 100, 128, 200, True), (300000, 200, 128, True), (700000, 64, 144, True)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(M, K, generator=g).to(DEV)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV) if bias else None
    wp = hl.pack_weight_x3(w)[0]
    L.nsdp_debug_set(6, 8192)
    y_direct = hl._fwd_x3(x, wp, N, b, None, None, None, False, False)
    L.nsdp_debug_set(6, 0)
    y_staged = hl._fwd_x3(x, wp, N, b, None, None, None, False, False)
    torch.cuda.synchronize()
    bad = (y_direct != y_staged)
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print(f"M={M} K={K} N={N}: {int(bad.sum())} differing elements in {rows.numel()} rows; "
          f"rows {rows[:6].tolist()} .. {rows[-3:].tolist() if rows.numel() else []}; cols {cols[:8].tolist()} .. {cols[-3:].tolist() if cols.numel() else []}")
    if rows.numel():
        tiles = torch.unique(rows // 256)
        print("   256-row tiles affected:", tiles[:12].tolist(), "... count", tiles.numel(), " rows%256 set:", torch.unique(rows % 256)[:20].tolist())
    if rows.numel():
        idx = bad.nonzero()[:12]
        for r, c in idx.tolist():
            print(f"     [{r},{c}] direct {float(y_direct[r, c]):+.6f} staged {float(y_staged[r, c]):+.6f}  bias {float(b[c]) if b is not None else 0:+.6f}  "
                  f"direct[r,c+1..3] {[round(float(v), 4) for v in y_direct[r, c + 1:c + 4]]}  tile-row {r % 256} lane-group {c % 16 // 4}")
        # is the staged value some OTHER element of the direct result?
        r, c = idx[0].tolist()
        hit = (y_direct == y_staged[r, c]).nonzero()[:4].tolist()
        print("     staged value found in direct at:", hit)
