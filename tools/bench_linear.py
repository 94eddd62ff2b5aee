import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd.hip_linear import _fwd, _wgrad, _fwd_wp, pack_weight, _fwd_x3, pack_weight_x3
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (M, K, N) in [(1835008, 200, 200), (1835008, 208, 208), (655360, 120, 120), (655360, 128, 128), (262144, 200, 128), (262144, 128, 128), (320000, 256, 256), (256000, 256, 256), (65536, 120, 120)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev)
    from nsdp_amd._lib import lib
    t1 = timeit(lambda: _fwd(x, w, b, None, None, None, False, True))
    res = torch.randn(M, N, device=dev); t1a = timeit(lambda: _fwd(x, w, b, res, None, None, False, False)); t1b = timeit(lambda: _fwd(x, w, b, None, x[:, :K] if False else None, res, False, False)); del res
    wp, _ = pack_weight(w); t1p = timeit(lambda: _fwd_wp(x, wp, N, b, None, None, None, False, True)); tpk = timeit(lambda: pack_weight(w, True, True))
    assert torch.equal(_fwd_wp(x, wp, N, b, None, None, None, False, True), _fwd(x, w, b, None, None, None, False, True))
    w3, _ = pack_weight_x3(w); t1x = timeit(lambda: _fwd_x3(x, w3, N, b, None, None, None, False, True))
    ref64 = torch.relu(x[:4096].double() @ w.double().t() + b.double())
    e32 = (_fwd(x[:4096].contiguous(), w, b, None, None, None, False, True).double() - ref64).abs().max().item() / ref64.abs().max().item()
    ex3 = (_fwd_x3(x[:4096].contiguous(), w3, N, b, None, None, None, False, True).double() - ref64).abs().max().item() / ref64.abs().max().item()
    t2 = timeit(lambda: F.relu(F.linear(x, w, b)))
    from nsdp_amd._lib import lib
    lib().nsdp_debug_set(5, 0); t3 = timeit(lambda: _wgrad(dy, x, None, False, True))
    lib().nsdp_debug_set(5, 1); t3b = timeit(lambda: _wgrad(dy, x, None, False, True))
    t4 = timeit(lambda: (dy.t() @ x, dy.sum(0)))
    fl = 2.0 * M * N * K
    print(f"M={M} K={K} N={N}: hip fwd {t1:.3f} ms {fl/t1/1e9:.1f} TF | packed-W {fl/t1p/1e9:.1f} TF (pack {tpk*1e3:.1f} us) | bf16x3 {t1x:.3f} ms {fl/t1x/1e9:.1f} TF err fp32 {e32:.1e} x3 {ex3:.1e} | +residual {fl/t1a/1e9:.1f} | +out_mask {fl/t1b/1e9:.1f} | torch fwd {t2:.3f} ms {fl/t2/1e9:.1f} TF | wgrad dword {fl/t3/1e9:.1f} TF vec4 {t3b:.3f} ms {fl/t3b/1e9:.1f} TF | torch wgrad {t4:.3f} ms {fl/t4/1e9:.1f} TF")
