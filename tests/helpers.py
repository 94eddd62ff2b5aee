"""Shared helpers for the parity tests."""
import contextlib
import copy
import json
import os

import numpy as np
import torch

from nsdp_amd import synth
from oracle import tdnet_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def model_cfg(mtype, npl):
    cfg = copy.deepcopy(tdnet_ref.DEFAULT_MODEL_CFG)
    cfg["type"] = mtype
    cfg["encoder_kwargs"]["npoints_per_layer"] = [int(x) for x in npl]
    return {"model": cfg}


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def fixture_setup(name, mtype):
    fx = load_fixture(name)
    seed, b, ns, nq = (int(fx[k]) for k in ("meta_seed", "meta_batch", "meta_ns", "meta_nq"))
    cfg = model_cfg(mtype, fx["meta_npl"])
    data = synth.make_batch(seed, b, ns, nq)
    return fx, cfg, seed, data


def build_product(cfg, seed, device):
    """The HIP product model with procedural weights (same generator the fixtures were made with)."""
    from nsdp_amd.model import build_model
    model, train_fn, val_fn, test_fn = build_model(cfg, device="cpu")
    state = synth.procedural_state_dict(model.state_dict(), seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    return model.to(device), train_fn, state


def to_dev(data, device):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in data.items()}


def run_forward(model, cfg, data):
    if cfg["model"]["type"] == "arbitrary":
        s = data["surface_samples_inputs"]
        return model(data["space_samples_src"], s[:, :, 0:3], s[:, :, 3:6], s[:, :, 6:7])
    return model(data["space_samples_src"], data["surface_samples_inputs"])


def sample_flat(t, n):
    f = t.detach().reshape(-1).cpu()
    if f.numel() <= n:
        return f.numpy()
    return f[torch.linspace(0, f.numel() - 1, n).long().clamp_(max=f.numel() - 1)].numpy()   # (as oracle/make_golden.py)


def l2_err(a, b):
    """max over shapes of sqrt(mean_n ||delta||^2)  (SURVEY.md section 8c parity metric)."""
    d = np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)
    return float(np.sqrt((d ** 2).sum(-1).mean(-1)).max())


def snapshot_model(model):
    """Clones of every parameter and buffer (weights, BatchNorm running statistics and counters)."""
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def restore_model(model, snap, optimizer=None):
    """Copies a snapshot_model() back IN PLACE (addresses stay what a captured step recorded) and, with ``optimizer``,
    returns its existing state to that of a fresh optimizer (moments and step counters zeroed in place): the next step
    is then step 1 from the snapshot's weights, whether it runs eagerly or as a replay."""
    from nsdp_amd import hip_linear
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(snap[k])
        if optimizer is not None:
            for st in optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
    hip_linear.invalidate_weight_packs()      # (weights rewritten behind the optimizer's back)


def nondeterministic_knobs():
    """Variant knobs of the library that put floating-point atomics back into the train step (A/B forms kept for comparison):
    under them two runs of a step differ in rounding, and the tests that hold the replay EQUAL to the eager step do not apply."""
    from nsdp_amd import hip_attention, pointnet2_utils
    out = []
    if not hip_attention.ONEHOT_SCATTER_F32:
        out.append("NSDP_ONEHOT_SCATTER_F32=0")
    if not hip_attention.ONEHOT_SCATTER:      # (bf16 storage: the decoder's anchor tables by register-table atomics)
        out.append("NSDP_ONEHOT_SCATTER=0")
    if hip_attention.INVERSE_LISTS == "0":
        out.append("NSDP_INVERSE_LISTS=0")
    if not pointnet2_utils._SCATTER_INVERSE:
        out.append("NSDP_SCATTER_ROWS=atomic")
    from nsdp_amd.model import ops
    if not ops.FUSE_DPOS:           # (d(pos) summed by the atomic form of attn_pre_bwd)
        out.append("NSDP_FUSE_DPOS=0")
    if not pointnet2_utils._SCATTER_DETERMINISTIC:      # (small / narrow scatters through the atomic kernel)
        out.append("NSDP_SCATTER_DETERMINISTIC=0")
    return out


@contextlib.contextmanager
def batchnorm_three_launch():
    """The three-launch BatchNorm forms for every row count (NSDP_BN_SLAB=0) inside the block: the SAME arithmetic as the default
    one-launch slab kernels with the batch mean summed in another order (1e-7 relative) -- the reference variant of the tests
    whose quantity is ill-conditioned enough to turn that into percents."""
    from nsdp_amd import hip_linear
    L = hip_linear.lib()
    L.nsdp_debug_set(11, 0)
    try:
        yield
    finally:
        L.nsdp_debug_set(11, int(os.environ.get("NSDP_BN_SLAB", "1")))
