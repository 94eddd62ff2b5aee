"""Run one linear kernel variant a few times (for rocprofv3 PMC passes): python tools/run_x3.py M K N [variant]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd.hip_linear import _fwd_wp, pack_weight, _fwd_x3, pack_weight_x3
M, K, N = (int(v) for v in sys.argv[1:4]); variant = sys.argv[4] if len(sys.argv) > 4 else "x3"
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
wp = pack_weight(w)[0]; w3 = pack_weight_x3(w)[0]
import time
from nsdp_amd._lib import lib
for dbg in ([0, 32] if variant == "x3dbg" else [int(variant[4:])] if variant.startswith("x3d=") else [0]):
  lib().nsdp_debug_set(6, dbg)
  torch.cuda.synchronize(); t0 = time.time()
  for _ in range(5):
      y = _fwd_x3(x, w3, N, b, None, None, None, False, True) if variant.startswith("x3") else _fwd_wp(x, wp, N, b, None, None, None, False, True)
  torch.cuda.synchronize(); print("dbg", dbg, "ms/launch", (time.time() - t0) / 5 * 1e3)
lib().nsdp_debug_set(6, 0)
ref = torch.relu(x[:8192].double() @ w.double().t() + b.double())
print("max rel err", ((y[:8192].double() - ref).abs().max() / ref.abs().max()).item(), "finite", bool(torch.isfinite(y).all()))
