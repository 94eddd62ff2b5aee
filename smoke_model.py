"""One tiny forward+backward of the TDNet hot path on the GPU, checked against the CPU oracle
(called from __graft_entry__.smoke(); imports the oracle as the checker only).  Lives at the repo root, NOT inside the
product package: nothing under nsdp_amd/ may import oracle/ (tests/test_no_oracle_in_product.py)."""
from __future__ import annotations

import copy

import numpy as np
import torch


def run(device) -> None:
    from nsdp_amd import synth
    from nsdp_amd.model import build_model, optimizer_factory
    from oracle import tdnet_ref

    from nsdp_amd.config import default_config
    cfg = default_config("forward")
    cfg["model"]["encoder_kwargs"]["npoints_per_layer"] = [256, 64, 16]
    model, train_fn, _, _ = build_model(cfg, device="cpu")
    state = synth.procedural_state_dict(model.state_dict(), 99)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.to(device)
    data = synth.make_batch(99, 2, 256, 128)
    dev = {k: torch.from_numpy(v).to(device) for k, v in data.items()}
    cpu = {k: torch.from_numpy(v) for k, v in data.items()}

    model.eval()
    with torch.no_grad():
        out = model(dev["space_samples_src"], dev["surface_samples_inputs"]).cpu().numpy()
        ref = tdnet_ref.model_forward(tdnet_ref.to_torch_state(state), cfg["model"], cpu).numpy()
    l2 = float(np.sqrt(((out - ref) ** 2).sum(-1).mean(-1)).max())
    assert l2 <= 1e-4, f"forward parity {l2}"

    model.train()
    _, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-4}, model.parameters())
    loss = train_fn(model, opt, dev, cfg)
    sd = tdnet_ref.to_torch_state(state, requires_grad=True)
    names = tdnet_ref.trainable(sd)
    ref_loss = tdnet_ref.train_step(sd, cfg["model"], cpu, torch.optim.Adam([sd[k] for k in names], lr=5e-4))
    assert abs(loss - ref_loss) <= 2e-5 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    print(f"smoke model: fwd L2 {l2:.2e}, train loss {loss:.6f} (oracle {ref_loss:.6f})")
