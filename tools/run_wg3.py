import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import hip_linear
M, N, K = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
dy = torch.randn(M, N, device=dev); x = torch.randn(M, K, device=dev)
for _ in range(5):
    hip_linear._wgrad_x3(dy, x, None, False, True)
torch.cuda.synchronize()
