import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import hip_linear
dev = torch.device("cuda:0")
for (M, N, K, mask) in [(4096, 200, 200, True), (4099, 120, 128, True), (65536, 200, 200, True), (65536, 128, 200, False)]:
    dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
    m = torch.randn(M, N, device=dev) if mask else None
    print("launch", M, N, K, mask, flush=True)
    dw3, db3 = hip_linear._wgrad_x3(dy, x, m, False, True)
    torch.cuda.synchronize()
    dyp = dy * (m > 0) if mask else dy
    ref = (dyp.t() @ x).double()
    print("  err", ((dw3.double() - ref).abs().max() / ref.abs().max()).item(), "db err", ((db3.double() - dyp.double().sum(0)).abs().max() / dyp.double().sum(0).abs().max()).item(), flush=True)
