"""K = 4 tail (hip_linear.K4Tail): the dX GEMM of a position-encoding MLP's second layer with the first layer's weight gradient in
its epilogue, against the two launches it replaces (plain dX GEMM + nsdp_linear_wgrad_k4_remask_f32).  ReLU-sparse dY like the step's."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nsdp_amd import hip_linear as hl
dev = torch.device('cuda:0')


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return min(ts) * 1e3


for (M, d) in [(1835008, 200), (320000, 256)]:
    torch.manual_seed(0)
    lin0 = torch.nn.Linear(3, d).to(dev); lin2 = torch.nn.Linear(d, d).to(dev)
    x4 = torch.nn.functional.pad(torch.randn(M, 3, device=dev), (0, 1)).contiguous()
    dy = torch.relu(torch.randn(M, d, device=dev))
    _, wpt = hl.pack_weight_x3(lin2.weight.detach(), True, True)
    link = hl.K4Tail(); link.w_param, link.b_param, link.k_orig = lin0.weight, lin0.bias, 3
    tail = hl._k4tail_fn(link, wpt, d)
    remask = hl._wgrad_k4_remask(hl._padded_w4(lin0.weight), lin0.bias.detach(), 3)

    def two():
        dh = hl._fwd_x3(dy, wpt, d, None, None, None, None, False, False)
        return remask(dh, x4, None, False, True)
    a = tail(dy, x4, None, False, True); b = two()
    err = max(float((p - q).abs().max() / q.abs().max()) for p, q in zip(a, b))
    t_tail = timeit(lambda: tail(dy, x4, None, False, True))
    t_dx = timeit(lambda: hl._fwd_x3(dy, wpt, d, None, None, None, None, False, False))
    dh = hl._fwd_x3(dy, wpt, d, None, None, None, None, False, False)
    t_k4 = timeit(lambda: remask(dh, x4, None, False, True))
    print(f"M={M} d={d}: tail {t_tail:.0f} us | dX {t_dx:.0f} + k4 wgrad {t_k4:.0f} = {t_dx + t_k4:.0f} us | rel diff {err:.1e}")
    if os.environ.get("K4TAIL_PHASES"):
        from nsdp_amd._lib import lib
        lib().nsdp_debug_set(6, 8)       # no epilogue at all (timing only)
        t_tail0 = timeit(lambda: tail(dy, x4, None, False, True)); t_dx0 = timeit(lambda: hl._fwd_x3(dy, wpt, d, None, None, None, None, False, False))
        lib().nsdp_debug_set(6, 0)
        print(f"   phases: GEMM without epilogue {t_dx0:.0f} us; store epilogue {t_dx - t_dx0:.0f}; tail epilogue {t_tail - t_tail0:.0f}; tail reduce launches {t_tail0 - t_dx0:.0f}")
