"""tests/golden/tiny_alternates.npz: the registry alternates (PointNet++ encoder + interpolation decoder, SURVEY.md
section 8 a20) run by the IMPORTED REFERENCE on CPU (same stub and procedural weights as oracle/make_golden.py).
Run in the build container only:  python oracle/make_golden_alternates.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nsdp_amd import synth  # noqa: E402
from oracle import make_golden, pointnet2_ref  # noqa: E402

ALT_CFG = {"model": {"type": "forward", "use_normals": False, "encoder": "pointnet++", "decoder": "interp",
                     "encoder_kwargs": {"npoints_per_layer": [256, 64, 16], "nneighbor": 16, "d_transformer": 256,
                                        "nfinal_transformers": 3},
                     "decoder_kwargs": {"dim_inp": 256, "dim": 200, "hidden_dim": 128, "out_dim": 3}}}


def main():
    pointnet2_ref.build()
    ref_model, _ = make_golden.import_reference()
    torch.set_num_threads(8)
    seed, B, ns, nq = 4321, 2, 256, 128
    model, train_on_batch, _, _ = ref_model.build_model(ALT_CFG, device="cpu")
    state = synth.procedural_state_dict(model.state_dict(), seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    data = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.make_batch(seed, B, ns, nq).items()}
    fx = {"meta_seed": np.int64(seed), "meta_batch": np.int64(B), "meta_ns": np.int64(ns), "meta_nq": np.int64(nq)}
    model.eval()
    with torch.no_grad():
        fx["eval_out"] = model(data["space_samples_src"], data["surface_samples_inputs"]).numpy()
    model.train()
    _, opt = ref_model.optimizer_factory({"optimizer": "Adam", "lr": 5e-4, "lr_step": 200, "lr_decay": 0.1,
                                          "weight_decay": 0.0}, model.parameters())
    fx["train_loss"] = np.float64(train_on_batch(model, opt, data, ALT_CFG))
    for k, p in model.named_parameters():
        if p.grad is not None:
            fx["grad_norm/" + k] = np.float64(p.grad.double().norm().item())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tiny_alternates.npz"), **fx)
    print("wrote tiny_alternates.npz: loss", fx["train_loss"], "params with grad", sum(k.startswith("grad_norm/") for k in fx))


if __name__ == "__main__":
    main()
