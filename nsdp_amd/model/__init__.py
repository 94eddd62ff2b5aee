"""Model factory of the MI355X TDNet hot path -- same API as the reference's model/__init__.py:
``build_model(config, weight_file, weight_forward_file, weight_backward_file, device)`` returning
``(model, train_on_batch, validate_on_batch, test_on_batch)`` and ``optimizer_factory(cfg, params)``."""
from __future__ import annotations

import os

import torch

from .deformation_networks import (Deformation_Networks, test_on_batch_with_cano, train_on_batch_with_cano,
                                   validate_on_batch_with_cano)
from .flow_arbitrary import (FlowArbitrary, test_on_batch_with_arbitrary, train_on_batch_with_arbitrary,
                             validate_on_batch_with_arbitrary)
from .learningrate import StepLearningRateSchedule


def optimizer_factory(config, parameters):
    """reference model/__init__.py:10-41 (Adam / SGD, one param group, step LR schedule)."""
    schedule = StepLearningRateSchedule({"type": "step", "initial": config.get("lr", 1e-3),
                                         "interval": config.get("lr_step", 100),
                                         "factor": config.get("lr_decay", 0.1)})
    name = config.get("optimizer", "Adam")
    parameters = list(parameters)
    group = {"params": parameters, "lr": schedule.get_learning_rate(0),
             "weight_decay": config.get("weight_decay", 0.0)}
    if name == "SGD":
        group["momentum"] = config.get("momentum", 0.9)
        return schedule, torch.optim.SGD([group])
    if name == "Adam":
        # GPU parameters: the whole update as one launch of csrc/adam.hip behind torch.optim.Adam's own container
        # (hip_adam.HipAdam; NSDP_HIP_ADAM=0 keeps PyTorch's kernels for an A/B).  CPU parameters (host-side tests, the
        # gloo harness): torch's Adam, as the reference.
        if (parameters and all(p.is_cuda and p.dtype == torch.float32 for p in parameters)
                and os.environ.get("NSDP_HIP_ADAM", "1") != "0"):
            from ..hip_adam import HipAdam
            return schedule, HipAdam([group])
        return schedule, torch.optim.Adam([group])
    raise NotImplementedError(name)


def _load(module, path, device):
    state = torch.load(path, map_location=device)
    try:
        module.load_state_dict(state)
    except Exception:
        module.load_state_dict(state["model_state_dict"])


def build_model(config, weight_file=None, weight_forward_file=None, weight_backward_file=None, device="cpu"):
    """reference model/__init__.py:43-118."""
    model_type = config["model"]["type"]
    if model_type in ("forward", "backward"):
        fns = (train_on_batch_with_cano, validate_on_batch_with_cano, test_on_batch_with_cano)
        model = Deformation_Networks(config, no_input_corr=(model_type == "backward"))
    elif model_type == "arbitrary":
        fns = (train_on_batch_with_arbitrary, validate_on_batch_with_arbitrary, test_on_batch_with_arbitrary)
        model_canonicalize = Deformation_Networks(config, no_input_corr=True)
        model_deform = Deformation_Networks(config, no_input_corr=False)
        model = FlowArbitrary(config, model_canonicalize, model_deform)
        if weight_forward_file is not None:
            _load(model_deform, weight_forward_file, device)
        if weight_backward_file is not None:
            _load(model_canonicalize, weight_backward_file, device)
    else:
        raise NotImplementedError(model_type)
    if weight_file is not None:
        _load(model, weight_file, device)
    model.to(device)
    return (model,) + fns
