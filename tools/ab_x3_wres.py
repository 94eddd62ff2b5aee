"""A/B of the bf16x3 GEMM with the weight planes RESIDENT in LDS (N, K <= 128) against the streaming form
(nsdp_debug_set(6, 256)): time per launch on the <= 128-wide layer shapes of the train step, ReLU-sparse and dense inputs,
results bit-compared.    python tools/ab_x3_wres.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd.hip_linear import _fwd_x3, pack_weight_x3
from nsdp_amd._lib import lib
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (M, K, N) in [(655360, 120, 120), (256000, 120, 120), (262144, 128, 128), (65536, 120, 120), (1835008, 128, 128), (262144, 128, 64)]:
    for relu_data in (False, True):
        x = torch.randn(M, K, device=dev)
        if relu_data:
            x = torch.relu(x)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)
        w3, _ = pack_weight_x3(w)
        out = {}
        for name, knob in (("resident", 0), ("streaming", 256)):
            lib().nsdp_debug_set(6, knob)
            t0 = timeit(lambda: _fwd_x3(x, w3, N, b, None, None, None, False, True))
            t1 = timeit(lambda: _fwd_x3(x, w3, N, b, res, None, None, True, False))
            out[name] = (t0, t1, _fwd_x3(x, w3, N, b, res, None, None, True, False))
        lib().nsdp_debug_set(6, 0)
        same = torch.equal(out["resident"][2], out["streaming"][2])
        byt = 4.0 * (M * (K + N) + N * K)
        print(f"M={M:8d} K={K} N={N} {'relu' if relu_data else 'dense'} data: plain+relu_out resident {out['resident'][0]:7.1f} us "
              f"({byt/out['resident'][0]/1e6:5.2f} TB/s) streaming {out['streaming'][0]:7.1f} us | relu_in+residual resident "
              f"{out['resident'][1]:7.1f} us streaming {out['streaming'][1]:7.1f} us | bit-identical {same}")
