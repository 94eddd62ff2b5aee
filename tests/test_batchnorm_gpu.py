"""Channels-last BatchNorm kernels (with fused residual-add prologue / ReLU epilogue) vs nn.BatchNorm1d."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("B,n,C,training,addend,relu", [
    (2, 37, 120, True, False, False), (3, 100, 256, True, True, False), (2, 500, 120, True, False, True),
    (1, 7, 8, True, True, True), (2, 64, 256, False, False, False), (2, 64, 120, False, True, True),
    (32, 2048, 120, True, False, False)])
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
