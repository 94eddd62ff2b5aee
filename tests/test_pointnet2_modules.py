"""pointnet2_modules (SA / SA-MSG / FP) on the HIP ops against the reference's modules run on CPU with the emulated
`_ext` (tests/golden/pointnet2_modules.npz, oracle/make_golden_modules.py)."""
import os

import numpy as np
import pytest
import torch

from nsdp_amd import synth

FX = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "pointnet2_modules.npz")))
C = 16
CASES = {
    "msg": lambda M: M.PointnetSAModuleMSG(64, [0.15, 0.3], [8, 16], [[C, 32, 48], [C, 32, 64]]),
    "sa_all": lambda M: M.PointnetSAModule([C, 64, 96]),
    "fp": lambda M: M.PointnetFPModule([C + 8, 64, 32]),
}


def _module(tag):
    from nsdp_amd import pointnet2_modules as M
    mod = CASES[tag](M)
    state = synth.procedural_state_dict(mod.state_dict(), int(FX[tag + "/seed"]))
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    return mod


@pytest.mark.parametrize("tag", list(CASES))
def test_state_dict_keys_match_reference(tag):
    mod = _module(tag)
    ref_params = sorted(k[len(tag + "/grad/"):] for k in FX if k.startswith(tag + "/grad/"))
    assert sorted(k for k, _ in mod.named_parameters()) == ref_params
    for k, p in mod.named_parameters():
        assert tuple(p.shape) == FX[tag + "/grad/" + k].shape, k


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_forward_backward_match_reference(tag):
    dev = torch.device("cuda:0")
    mod = _module(tag).to(dev).train()
    xyz = torch.from_numpy(FX["xyz"]).to(dev)
    feats = torch.from_numpy(FX["feats"]).to(dev).requires_grad_(True)
    if tag == "fp":
        out = mod(torch.from_numpy(FX["fp/unknown"]).to(dev), xyz, torch.from_numpy(FX["fp/ufe"]).to(dev), feats)
    else:
        new_xyz, out = mod(xyz, feats)
        if new_xyz is not None:
            assert np.array_equal(new_xyz.cpu().numpy(), FX[tag + "/new_xyz"])       # FPS: bit-exact centres
    ref = FX[tag + "/out"]
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    out.backward(torch.from_numpy(FX[tag + "/go"]).to(dev))
    g = FX[tag + "/dfeats"]
    np.testing.assert_allclose(feats.grad.cpu().numpy(), g, rtol=0, atol=2e-4 * np.abs(g).max() + 1e-6)
    for k, p in mod.named_parameters():
        r = FX[tag + "/grad/" + k]
        # (conv biases do not exist with bn=True; BN affine gradients are well conditioned)
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=0, atol=3e-4 * np.abs(r).max() + 2e-5, err_msg=k)
    for k, v in mod.state_dict().items():
        if k.endswith(("running_mean", "running_var")):
            np.testing.assert_allclose(v.cpu().numpy(), FX[tag + "/bn/" + k], rtol=1e-4, atol=1e-6, err_msg=k)
