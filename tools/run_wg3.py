"""One weight-gradient problem, five launches (for rocprofv3 / PMC passes): run_wg3.py M N K [mask: none|ones|self]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import hip_linear
M, N, K = (int(v) for v in sys.argv[1:4])
kind = sys.argv[4] if len(sys.argv) > 4 else "none"
dev = torch.device("cuda:0")
dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
mask = None if kind == "none" else torch.ones(M, N, device=dev) if kind == "ones" else dy
for _ in range(5):
    hip_linear._wgrad_x3(dy, x, mask, False, True)
torch.cuda.synchronize()
