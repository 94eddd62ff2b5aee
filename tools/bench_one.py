import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd.hip_linear import _fwd, _wgrad
dev = torch.device('cuda:0')
M, K, N = 1835008, 208, 208
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); dy = torch.randn(M, N, device=dev)
for _ in range(3):
    _fwd(x, w, b, None, None, None, False, True)
    _wgrad(dy, x, None, False, True)
torch.cuda.synchronize()
