"""Two-network composition source -> canonical -> target
(mirror of the reference's model/flow_arbitrary.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import precision
from .utils import compute_l2_error


class FlowArbitrary(nn.Module):
    """reference model/flow_arbitrary.py:8-27."""

    def __init__(self, cfg, model_canonicalize, model_deform):
        super().__init__()
        self.model_canonicalize = model_canonicalize
        self.model_deform = model_deform

    def forward(self, space_samples_src, surface_samples_src, surface_samples_tgt, cano_handle_sample_mask):
        if precision.is_bf16() and precision.canonicalize_f32():
            # mixed storage (see nsdp_amd/precision.py): the network whose output points feed the second network's geometry
            # runs in fp32 storage; inputs and outputs are fp32 coordinates in either mode, so there is nothing to cast
            with precision.storage(torch.float32):
                space_src2cano = self.model_canonicalize(space_samples_src, surface_samples_src)
                surf_src2cano = self.model_canonicalize(surface_samples_src, surface_samples_src)
        else:
            space_src2cano = self.model_canonicalize(space_samples_src, surface_samples_src)
            surf_src2cano = self.model_canonicalize(surface_samples_src, surface_samples_src)
        deform_in = torch.cat([surf_src2cano, surface_samples_tgt, cano_handle_sample_mask], dim=-1).contiguous()
        return self.model_deform(space_src2cano, deform_in)


def _split(data_dict):
    s = data_dict["surface_samples_inputs"]
    return s[:, :, 0:3], s[:, :, 3:6], s[:, :, 6:7]


def _loss_with_arbitrary(model, data_dict, config):
    """forward + l2 loss of train_on_batch_with_arbitrary (reference :33-43), as a tensor."""
    src, tgt, mask = _split(data_dict)
    pred = model(data_dict["space_samples_src"], src, tgt, mask)
    return compute_l2_error(pred, data_dict["space_samples_tgt"])


def _train_step_with_arbitrary(model, optimizer, data_dict, config):
    """train_on_batch_with_arbitrary without the host read-back of the loss (what nsdp_amd.graph_step captures)."""
    optimizer.zero_grad()
    loss = _loss_with_arbitrary(model, data_dict, config)
    loss.backward()
    optimizer.step()
    return loss


def train_on_batch_with_arbitrary(model, optimizer, data_dict, config):
    """reference model/flow_arbitrary.py:30-48."""
    return _train_step_with_arbitrary(model, optimizer, data_dict, config).item()


train_on_batch_with_arbitrary.tensor_step = _train_step_with_arbitrary
train_on_batch_with_arbitrary.loss_fn = _loss_with_arbitrary


@torch.no_grad()
def validate_on_batch_with_arbitrary(model, data_dict, config):
    """reference model/flow_arbitrary.py:51-63."""
    src, tgt, mask = _split(data_dict)
    pred = model(data_dict["space_samples_src"], src, tgt, mask)
    return compute_l2_error(pred, data_dict["space_samples_tgt"]).item()


@torch.no_grad()
def test_on_batch_with_arbitrary(model, data_dict, config, compute_loss=False):
    """reference model/flow_arbitrary.py:65-85."""
    src, tgt, mask = _split(data_dict)
    data_dict["surface_samples_tgt_pred"] = model(src, src, tgt, mask)
    deformed_verts = model(data_dict["verts_src"], src, tgt, mask)
    data_dict["verts_tgt_pred"] = deformed_verts
    if compute_loss:
        loss = compute_l2_error(deformed_verts, data_dict["verts_tgt"])
    else:
        loss = torch.zeros((1), dtype=torch.float32)
    return loss.item(), data_dict
