"""Channels-last BatchNorm kernels (with fused residual-add prologue / ReLU epilogue) vs nn.BatchNorm1d."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("B,n,C,training,addend,relu", [
    (2, 37, 120, True, False, False), (3, 100, 256, True, True, False), (2, 500, 120, True, False, True),
    (1, 7, 8, True, True, True), (2, 64, 256, False, False, False), (2, 64, 120, False, True, True),
    (32, 2048, 120, True, False, False),
    # the one-launch slab kernels (csrc/batchnorm.hip): every (rows, threads, vector) variant and its edges
    (32, 100, 256, True, True, True), (32, 500, 120, True, False, True), (32, 500, 256, True, True, False),
    (1, 4096, 120, True, True, True), (1, 4097, 256, True, False, False), (1, 8192, 120, True, True, True),
    (1, 8193, 256, True, True, True), (1, 16384, 8, True, False, True), (1, 16385, 120, True, True, False),
    (32, 500, 256, False, True, True)])
def test_batch_norm_matches_torch(B, n, C, training, addend, relu):
    from nsdp_amd.hip_batchnorm import batch_norm
    g = torch.Generator().manual_seed(B * 100 + n + C)
    x = (torch.randn(B, n, C, generator=g) * 2 + 0.5).to(DEV)
    a = torch.randn(B, n, C, generator=g).to(DEV) if addend else None
    bn = torch.nn.BatchNorm1d(C).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    ref_bn = torch.nn.BatchNorm1d(C).to(DEV).double()
    ref_bn.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    bn.train(training)
    ref_bn.train(training)
    x1 = x.clone().requires_grad_(True)
    a1 = a.clone().requires_grad_(True) if addend else None
    y = batch_norm(x1, bn, addend=a1, relu=relu)
    x2 = x.double().requires_grad_(True)
    a2 = a.double().requires_grad_(True) if addend else None
    inp = x2 + a2 if addend else x2
    yr = ref_bn(inp.permute(0, 2, 1)).permute(0, 2, 1)
    yr = F.relu(yr) if relu else yr
    assert float((y.double() - yr).abs().max()) < 2e-5
    go = torch.randn(B, n, C, generator=g).to(DEV)
    ins1 = [t for t in (x1, a1, bn.weight, bn.bias) if t is not None]
    ins2 = [t for t in (x2, a2, ref_bn.weight, ref_bn.bias) if t is not None]
    g1 = torch.autograd.grad(y, ins1, go)
    g2 = torch.autograd.grad(yr, ins2, go.double())
    for u, v in zip(g1, g2):
        assert float((u.double() - v).abs().max()) <= 2e-5 * (float(v.abs().max()) + 1.0), u.shape
    if training:
        assert float((bn.running_mean.double() - ref_bn.running_mean).abs().max()) < 1e-6
        assert float((bn.running_var.double() - ref_bn.running_var).abs().max()) < 1e-5
        assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1


@pytest.mark.parametrize("R", [3200, 16000, 65536])
def test_running_updates_equals_repeated_forward_passes(R):
    """hip_batchnorm.running_updates(n): ONE norm over a batch leaves in the running statistics and the batch counter what
    n forward passes of nn.BatchNorm1d over that same batch leave (FlowArbitrary's reference encodes one cloud twice per
    step, model/flow_arbitrary.py:19-20; this library encodes it once) -- one-launch and three-launch forms."""
    from nsdp_amd import hip_batchnorm as hbn
    C = 120
    g = torch.Generator().manual_seed(R)
    x = (torch.randn(R, C, generator=g) * 1.5 - 0.3).to(DEV)
    bn = torch.nn.BatchNorm1d(C).to(DEV)
    ref = torch.nn.BatchNorm1d(C).to(DEV).double()
    with torch.no_grad():
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    with hbn.running_updates(2):
        y = hbn.batch_norm(x, bn)
    y1 = ref(x.double())
    y2 = ref(x.double())
    assert torch.equal(y1, y2)
    assert float((y.double() - y2).abs().max()) < 2e-5
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 2
    assert float((bn.running_mean.double() - ref.running_mean).abs().max()) < 1e-6
    assert float((bn.running_var.double() - ref.running_var).abs().max()) < 1e-5
    hbn.batch_norm(x, bn)                       # (the context is over: one update again)
    ref(x.double())
    assert int(bn.num_batches_tracked) == 3
    assert float((bn.running_var.double() - ref.running_var).abs().max()) < 1e-5


def test_slab_and_three_launch_forms_agree_and_are_selected():
    """NSDP_BN_SLAB knob (nsdp_debug_set(11, v)): both forms of the same norm, and the kernel-variant trace showing which ran."""
    import ctypes
    from nsdp_amd import _lib
    from nsdp_amd.hip_batchnorm import batch_norm
    L = _lib.lib()
    outs = {}
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(16000, 256, generator=g).to(DEV)
    go = torch.randn(16000, 256, generator=g).to(DEV)
    for slab in (2, 0):                      # (2: slab kernels up to 16384 rows -- the default stops at 4096)
        L.nsdp_debug_set(11, slab)
        try:
            L.nsdp_trace_enable(1)
            bn = torch.nn.BatchNorm1d(256).to(DEV)
            x = x0.clone().requires_grad_(True)
            y = batch_norm(x, bn, relu=True)
            y.backward(go)
            L.nsdp_trace_enable(0)
            n = L.nsdp_trace_read(None, 0)
            buf = ctypes.create_string_buffer(n)
            L.nsdp_trace_read(buf, n)
            names = set(buf.value.decode().split("\n"))
            outs[slab] = (y.detach(), x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone(), names)
        finally:
            L.nsdp_debug_set(11, 1)
    assert {"bn_slab_fwd<4,512>", "bn_slab_bwd<2,512>"} <= outs[2][6], outs[2][6]
    assert not any(n.startswith("bn_slab") for n in outs[0][6]), outs[0][6]
    for a, b in zip(outs[2][:6], outs[0][:6]):
        assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1.0)
