"""Pins the CPU oracle (oracle/tdnet_ref.py + oracle/pointnet2_ref.c) against fixtures produced by
the IMPORTED REFERENCE (oracle/make_golden.py).  CPU only."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from nsdp_amd import synth
from oracle import pointnet2_ref, tdnet_ref


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False))


def _cfg(mtype, npl):
    cfg = copy.deepcopy(tdnet_ref.DEFAULT_MODEL_CFG)
    cfg["type"] = mtype
    cfg["encoder_kwargs"]["npoints_per_layer"] = [int(x) for x in npl]
    return cfg


def _template(golden_dir, mtype):
    def one(t):
        with open(os.path.join(golden_dir, f"state_template_{t}.json")) as f:
            return {k: np.empty(v, dtype=np.float32) for k, v in json.load(f).items()}
    if mtype != "arbitrary":
        return one(mtype)
    out = {"model_canonicalize." + k: v for k, v in one("backward").items()}
    out.update({"model_deform." + k: v for k, v in one("forward").items()})
    return out


def _setup(golden_dir, name, mtype, requires_grad=False):
    fx = _load(golden_dir, name)
    seed, b, ns, nq = (int(fx[k]) for k in ("meta_seed", "meta_batch", "meta_ns", "meta_nq"))
    cfg = _cfg(mtype, fx["meta_npl"])
    state = synth.procedural_state_dict(_template(golden_dir, mtype), seed)
    sd = tdnet_ref.to_torch_state(state, requires_grad=requires_grad)
    data = {k: torch.from_numpy(v) for k, v in synth.make_batch(seed, b, ns, nq).items()}
    return fx, cfg, sd, data


def _sample_flat(t, n):
    f = t.detach().reshape(-1)
    if f.numel() <= n:
        return f.numpy()
    return f[torch.linspace(0, f.numel() - 1, n).long().clamp_(max=f.numel() - 1)].numpy()


@pytest.mark.parametrize("mtype", ["forward", "backward", "arbitrary"])
def test_eval_forward_matches_reference(golden_dir, mtype):
    fx, cfg, sd, data = _setup(golden_dir, "tiny_" + mtype, mtype)
    tape = {}
    with torch.no_grad():
        out = tdnet_ref.model_forward(sd, cfg, data, training=False, tape=tape)
    np.testing.assert_allclose(out.numpy(), fx["eval_out"], rtol=0, atol=2e-5)
    n_checked = 0
    for key, ref in fx.items():
        if not key.startswith("eval_tap/"):
            continue
        mine = tape[key[len("eval_tap/"):]]
        mine = mine.numpy() if mine.numel() == ref.size and mine.dim() > 1 else _sample_flat(mine, 64)
        np.testing.assert_allclose(mine.reshape(ref.shape), ref, rtol=0, atol=5e-5, err_msg=key)
        n_checked += 1
    assert n_checked >= 15


def test_full_shape_forward_matches_reference(golden_dir):
    """BASELINE configs[0] geometry: forward.yaml architecture, B=1, 2048 surface + 8192 queries."""
    fx, cfg, sd, data = _setup(golden_dir, "full_forward", "forward")
    with torch.no_grad():
        out = tdnet_ref.model_forward(sd, cfg, data, training=False)
    l2 = float(np.sqrt(((out.numpy() - fx["eval_out"]) ** 2).sum(-1).mean()))
    assert l2 <= 1e-5, l2


@pytest.mark.parametrize("mtype", ["forward", "backward"])
def test_geometry_matches_reference(golden_dir, mtype):
    """FPS indices and kNN index sets (the reference's own square_distance + argsort) are exact."""
    fx, cfg, sd, data = _setup(golden_dir, "tiny_" + mtype, mtype)
    npl = cfg["encoder_kwargs"]["npoints_per_layer"]
    xyz0 = data["surface_samples_inputs"][:, :, :3].contiguous().numpy()
    fps1 = pointnet2_ref.furthest_point_sampling(xyz0, npl[1])
    xyz1 = np.take_along_axis(xyz0, fps1[..., None].astype(np.int64), axis=1)
    fps2 = pointnet2_ref.furthest_point_sampling(xyz1, npl[2])
    xyz2 = np.take_along_axis(xyz1, fps2[..., None].astype(np.int64), axis=1)
    np.testing.assert_array_equal(fps1, fx["geo/fps1"])
    np.testing.assert_array_equal(fps2, fx["geo/fps2"])
    q = data["space_samples_src"].numpy()
    sites = {"begin": (xyz0, xyz0, 10), "tsa0": (xyz1, xyz0, 16), "down0": (xyz1, xyz1, 16),
             "tsa1": (xyz2, xyz1, 16), "down1": (xyz2, xyz2, 16), "dec": (q, xyz2, 7)}
    for name, (a, b, k) in sites.items():
        idx, d2 = pointnet2_ref.knn(a, b, k, return_dist=True)
        np.testing.assert_array_equal(idx, fx["geo/knn_" + name], err_msg=name)
    # distance bit pattern: ((dx*dx)+(dy*dy))+(dz*dz), products rounded separately (no FMA)
    _, d2 = pointnet2_ref.knn(xyz1[:, :8], xyz0[:, :64], 64, return_dist=True)
    ref_sorted = np.sort(fx["geo/sqdist_sample"], axis=-1)
    assert np.array_equal(d2.view(np.uint32), ref_sorted.view(np.uint32))


@pytest.mark.parametrize("mtype", ["forward", "backward", "arbitrary"])
def test_train_step_matches_reference(golden_dir, mtype):
    """loss, per-parameter gradients, None-grad list, post-Adam deltas, BN running stats."""
    fx, cfg, sd, data = _setup(golden_dir, "tiny_" + mtype, mtype, requires_grad=True)
    names = tdnet_ref.trainable(sd)
    before = {k: sd[k].detach().clone() for k in names}
    opt = torch.optim.Adam([{"params": [sd[k] for k in names], "lr": 5e-4, "weight_decay": 0.0}])
    loss = tdnet_ref.train_step(sd, cfg, data, opt)
    assert abs(loss - float(fx["train_loss"])) <= 1e-5 * max(1.0, abs(loss))
    none = sorted(k for k in names if sd[k].grad is None)
    assert none == sorted(str(s) for s in fx["none_grads"])
    for k in names:
        if sd[k].grad is None:
            continue
        gn = float(fx["grad_norm/" + k])
        mine = float(sd[k].grad.double().norm())
        assert abs(mine - gn) <= 2e-4 * gn + 1e-7, (k, mine, gn)
        np.testing.assert_allclose(_sample_flat(sd[k].grad, 16), fx["grad_sample/" + k],
                                   rtol=2e-3, atol=2e-5 * max(gn, 1e-3), err_msg=k)
    for key, ref in fx.items():
        if key.startswith("bn_after/"):
            mine = sd[key[len("bn_after/"):]]
            mine = mine.numpy() if key.endswith("num_batches_tracked") else _sample_flat(mine, 16)
            np.testing.assert_allclose(mine, ref, rtol=1e-4, atol=1e-6, err_msg=key)
    # Adam's first step is +-lr*sign(g) up to eps: compare only where |g| is well above eps
    for k in names:
        if sd[k].grad is None:
            continue
        g = fx["grad_sample/" + k]
        d = _sample_flat(sd[k].detach() - before[k], 16)
        sel = np.abs(g) > 1e-6
        np.testing.assert_allclose(d[sel], fx["delta_sample/" + k][sel], rtol=1e-3, atol=1e-7, err_msg=k)
