"""Model configuration of the hot path: the `model:` block every shipped NSDP YAML shares
(/root/reference/config/deform4d/forward.yaml:23-41) as a plain dict; reference YAML files load too."""
from __future__ import annotations

import copy

_MODEL = {
    "type": "forward",
    "use_normals": False,
    "encoder": "pointransformer",
    "encoder_kwargs": {
        "npoints_per_layer": [5000, 500, 100],
        "nneighbor": 16,
        "nneighbor_reduced": 10,
        "nfinal_transformers": 3,
        "d_transformer": 256,
        "d_reduced": 120,
        "full_SA": True,
    },
    "decoder": "crossatten",
    "decoder_kwargs": {"dim_inp": 256, "dim": 200, "nneigh": 7, "hidden_dim": 128, "out_dim": 3},
}

_TRAINING = {"optimizer": "Adam", "lr": 0.0005, "lr_step": 200, "lr_decay": 0.1, "weight_decay": 0.0,
             "batch_size": 16, "epochs": 600}


def default_config(model_type: str = "forward") -> dict:
    """{'model': ..., 'training': ...} with type in {'forward', 'backward', 'arbitrary'}."""
    cfg = {"model": copy.deepcopy(_MODEL), "training": copy.deepcopy(_TRAINING)}
    cfg["model"]["type"] = model_type
    return cfg


def load_config(path: str) -> dict:
    """Reads a reference-style experiment YAML (utils/training_utils.py:14-17)."""
    import yaml
    with open(path) as f:
        return yaml.safe_load(f)
