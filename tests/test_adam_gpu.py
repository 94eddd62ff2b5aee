"""csrc/adam.hip (one launch for every parameter tensor) against torch.optim.Adam -- the optimizer of the reference's train
step (/root/reference/model/__init__.py:10-41) -- on the CPU in fp32, operation by operation."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1,), (3,), (120, 3), (200, 200), (4095,), (4096,), (4097,), (256, 256), (3, 128), (17, 5, 1), (70001,)]


def _params(seed, device, misaligned=False):
    g = torch.Generator().manual_seed(seed)
    out = []
    for i, shp in enumerate(SHAPES):
        n = 1
        for s in shp:
            n *= s
        if misaligned and i % 2 == 1:      # a view that starts 4 bytes into an allocation: the scalar path of the kernel
            flat = torch.zeros(n + 1, device=device)
            flat[1:] = torch.randn(n, generator=g).to(device)
            p = flat[1:].view(shp)
        else:
            p = torch.randn(shp, generator=g).to(device)
        out.append(p.requires_grad_(True))
    return out


def _grads(seed, step, scale=1.0):
    g = torch.Generator().manual_seed(1000 * seed + step)
    return [scale * torch.randn(shp, generator=g) * (10.0 ** ((i % 5) - 3)) for i, shp in enumerate(SHAPES)]


def _run(opt_cls, params, device, steps, skip, lr_tensor=False, **kw):
    opt = opt_cls(params, **kw)
    if lr_tensor:
        opt.param_groups[0]["lr"] = torch.tensor(kw["lr"], dtype=torch.float32, device=device)
    for t in range(steps):
        for i, (p, g) in enumerate(zip(params, _grads(7, t))):
            p.grad = None if i in skip else g.to(device)
        opt.step()
    return opt


@pytest.mark.parametrize("weight_decay,misaligned,lr_tensor", [(0.0, False, False), (0.01, True, True)])
def test_hip_adam_matches_torch_cpu_adam(weight_decay, misaligned, lr_tensor):
    from nsdp_amd.hip_adam import HipAdam
    dev = torch.device("cuda:0")
    skip = {2, 6}          # parameters without a gradient are left alone (no state, no update)
    kw = dict(lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
    cpu = [p.detach().cpu().clone().requires_grad_(True) for p in _params(3, "cpu")]
    gpu = _params(3, dev, misaligned)
    ref = _run(lambda ps, **k: torch.optim.Adam(ps, foreach=False, **k), cpu, "cpu", 12, skip, **kw)
    opt = _run(HipAdam, gpu, dev, 12, skip, lr_tensor=lr_tensor, **kw)
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(cpu, gpu)):
        b = b.detach().cpu()
        if i in skip:
            assert len(opt.state[gpu[i]]) == 0 and torch.equal(a.detach(), b)
            continue
        sa, sb = ref.state[a], opt.state[gpu[i]]
        assert float(sb["step"]) == 12.0 and sb["step"].is_cuda
        # One rounding per operation here; the CPU's vectorised lerp_ / addcmul_ use fused multiply-adds where the ISA has
        # them, so the moments agree to a few units in the last place OF THE LARGEST TERM (m = m + w (g - m) cancels).
        for name in ("exp_avg", "exp_avg_sq"):
            ra = sa[name]
            torch.testing.assert_close(sb[name].cpu(), ra, rtol=1e-6, atol=4e-7 * float(ra.abs().max()))
        # parameters: at most one unit in the last place (|p| < 8) against updates of 5e-4 per step
        torch.testing.assert_close(b, a.detach(), rtol=0, atol=6e-7)


def test_hip_adam_state_dict_travels_both_ways():
    from nsdp_amd.hip_adam import HipAdam
    dev = torch.device("cuda:0")
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    a = _params(5, dev)
    opt = _run(HipAdam, a, dev, 3, set(), **kw)
    sd = copy.deepcopy(opt.state_dict())
    # into PyTorch's own Adam (a checkpoint written here, read by the reference) ...
    b = [p.detach().clone().requires_grad_(True) for p in a]
    tor = torch.optim.Adam(b, foreach=False, capturable=True, **kw)
    tor.load_state_dict(sd)
    # ... and back (a checkpoint written by the reference's non-capturable Adam: step counters on the host)
    c = [p.detach().clone().requires_grad_(True) for p in a]
    plain = torch.optim.Adam([p.detach().cpu().clone().requires_grad_(True) for p in a], **kw)
    plain.load_state_dict(copy.deepcopy(sd))
    for st in plain.state.values():
        st["step"] = torch.tensor(float(st["step"]))
    back = HipAdam(c, **kw)
    back.load_state_dict(plain.state_dict())
    for t in range(3, 6):
        for ps, o in ((a, opt), (b, tor), (c, back)):
            for p, g in zip(ps, _grads(7, t)):
                p.grad = g.to(dev)
            o.step()
    torch.cuda.synchronize()
    for pa, pb, pc in zip(a, b, c):
        assert torch.equal(pa, pc)                                   # HipAdam resumed from the file == HipAdam
        torch.testing.assert_close(pa, pb, rtol=0, atol=6e-7)        # PyTorch's capturable kernels resumed from it
        assert float(opt.state[pa]["step"]) == float(back.state[pc]["step"]) == 6.0


def test_hip_adam_replayed_step_equals_eager():
    """The update captured once and replayed (tables rebuilt inside the capture, step counters advanced on the device,
    learning rate changed between replays) is the eager sequence bit for bit."""
    from nsdp_amd.graph_step import GraphedStep, set_lr
    from nsdp_amd.hip_adam import HipAdam
    dev = torch.device("cuda:0")
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    a, b = _params(9, dev), _params(9, dev)
    static = [torch.zeros_like(p) for p in b]
    oa, ob = HipAdam(a, **kw), HipAdam(b, **kw)
    for o in (oa, ob):
        o.param_groups[0]["lr"] = torch.tensor(1e-3, dtype=torch.float32, device=dev)

    def feed(t):
        gs = _grads(11, t)
        for p, g in zip(a, gs):
            p.grad = g.to(dev)
        for s, g in zip(static, gs):
            s.copy_(g)
    feed(0)
    for p, s in zip(b, static):
        p.grad = s.clone()
    oa.step()
    ob.step()                       # (first step eagerly: creates the state and the spare pinned table)

    def captured():
        for p, s in zip(b, static):
            p.grad = s * 1.0        # gradients born inside the graph's pool: the tables are rebuilt in the capture
        ob.step()
    step = GraphedStep(captured).capture(warmup=0)
    for t in range(1, 6):
        if t == 3:
            set_lr(oa, 2e-4)
            set_lr(ob, 2e-4)
        feed(t)
        oa.step()
        step()
    torch.cuda.synchronize()
    step.close()
    for pa, pb in zip(a, b):
        assert torch.equal(pa, pb)
        assert float(ob.state[pb]["step"]) == 6.0
