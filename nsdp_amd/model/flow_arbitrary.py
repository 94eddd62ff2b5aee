"""Two-network composition source -> canonical -> target
(mirror of the reference's model/flow_arbitrary.py)."""
from __future__ import annotations

import contextlib

import torch
import torch.nn as nn

from .. import hip_batchnorm, precision
from . import deformation_networks
from .utils import compute_l2_error


class FlowArbitrary(nn.Module):
    """reference model/flow_arbitrary.py:8-27."""

    def __init__(self, cfg, model_canonicalize, model_deform):
        super().__init__()
        self.model_canonicalize = model_canonicalize
        self.model_deform = model_deform

    def _canonicalize_storage(self):
        # mixed storage (see nsdp_amd/precision.py): the network whose output points feed the second network's geometry
        # runs in fp32 storage; inputs and outputs are fp32 coordinates in either mode, so there is nothing to cast
        if precision.is_bf16() and precision.canonicalize_f32():
            return precision.storage(torch.float32)
        return contextlib.nullcontext()

    def canonicalize(self, query_sets, surface_samples_src):
        """model_canonicalize applied to several query sets against the SAME source cloud: ONE encoder pass, ONE decoder pass
        over the concatenated queries (the decoder has no BatchNorm and treats query points independently).  The reference
        runs the whole network once per set (model/flow_arbitrary.py:19-20); in training mode the two encoder passes see the
        same batch, produce the same tensors and differ only in what they leave in the BatchNorm buffers -- two momentum
        updates from the same batch statistics, num_batches_tracked += 2 -- which hip_batchnorm.running_updates reproduces.
        Autograd sums the decoder paths into one encoder backward."""
        net = self.model_canonicalize
        with self._canonicalize_storage():
            if not deformation_networks.ENCODE_ONCE:
                return [net(q, surface_samples_src) for q in query_sets]
            queries = query_sets[0] if len(query_sets) == 1 else torch.cat(list(query_sets), dim=1)
            queries = queries if queries.is_contiguous() else queries.contiguous()
            with hip_batchnorm.running_updates(len(query_sets)):      # (inert on eval-mode norms)
                encoding = net.encode(surface_samples_src, queries=queries)
            if precision.is_bf16() and precision.canonicalize_decoder_f32():
                # the middle point of the mixed storage (precision.py): bf16 encoder, fp32 decoder -- the encoding (one latent
                # code and 100 anchor features per shape) is cast once, the per-point chain runs in fp32 storage
                with precision.storage(torch.float32):
                    enc32 = {k: (v.float() if torch.is_tensor(v) and v.dtype is torch.bfloat16 else v) for k, v in encoding.items()}
                    out = net.decode(queries, enc32)
            else:
                out = net.decode(queries, encoding)
            if len(query_sets) == 1:
                return [out]
            return list(torch.split(out, [q.shape[1] for q in query_sets], dim=1))

    def deform_input(self, surf_src2cano, surface_samples_tgt, cano_handle_sample_mask):
        return torch.cat([surf_src2cano, surface_samples_tgt, cano_handle_sample_mask], dim=-1).contiguous()

    def forward(self, space_samples_src, surface_samples_src, surface_samples_tgt, cano_handle_sample_mask):
        space_src2cano, surf_src2cano = self.canonicalize([space_samples_src, surface_samples_src], surface_samples_src)
        deform_in = self.deform_input(surf_src2cano, surface_samples_tgt, cano_handle_sample_mask)
        return self.model_deform(space_src2cano.contiguous(), deform_in)


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
    if deformation_networks.ENCODE_ONCE:
        # the reference's two model() calls (:71, :76) run six encoder passes over two distinct clouds: the source cloud
        # (four times) and the canonicalised surface + target + mask (twice).  Two passes here.
        surf2cano, verts2cano = model.canonicalize([src, data_dict["verts_src"]], src)
        encoding = model.model_deform.encode(model.deform_input(surf2cano, tgt, mask))
        data_dict["surface_samples_tgt_pred"] = model.model_deform.decode(surf2cano.contiguous(), encoding)
        deformed_verts = model.model_deform.decode(verts2cano.contiguous(), encoding)
    else:
        data_dict["surface_samples_tgt_pred"] = model(src, src, tgt, mask)
        deformed_verts = model(data_dict["verts_src"], src, tgt, mask)
    data_dict["verts_tgt_pred"] = deformed_verts
    if compute_loss:
        loss = compute_l2_error(deformed_verts, data_dict["verts_tgt"])
    else:
        loss = torch.zeros((1), dtype=torch.float32)
    return loss.item(), data_dict
