"""Deformation network = encoder + decoder, and its step functions
(mirror of the reference's model/deformation_networks.py)."""
from __future__ import annotations

import inspect
import os

import torch
import torch.nn as nn

from .decoder import decoder_dict
from .encoder import encoder_dict
from .utils import compute_l2_error

# One encoder pass per distinct surface cloud (NSDP_ENCODE_ONCE=0: one per module call, the reference's op sequence -- A/B)
ENCODE_ONCE = os.environ.get("NSDP_ENCODE_ONCE", "1") != "0"
# NSDP_DECODER_PREFETCH=1: the decoder's anchor-only work (anchor search, relative coordinates, position-encoding MLP: ~1.8 ms
# at B = 32) on its own stream beside the encoder's forward chain instead of behind it.  OFF by default -- measured on one box,
# interleaved: B = 32 41.30 / 41.38 ms against 40.91 / 41.03 (41.06-41.18 with the prefetch GEMM confined to half the chip), bf16
# +0.1 ms, FlowArbitrary +1 ms; B = 8 14.73 / 14.76 against 14.86 / 14.90 (the only win).  The encoder's forward is not idle
# under it: its 500- and 100-point levels are 256 k- and 320 k-row GEMMs that fill the chip, so the two chains time-share, and
# the early launch gives up the q - k + pos epilogue of the position-encoding GEMM (docs/EXPERIMENTS.md, round 5).
DECODER_PREFETCH = os.environ.get("NSDP_DECODER_PREFETCH", "0") == "1"


class Deformation_Networks(nn.Module):
    """reference model/deformation_networks.py:12-60 (input-channel logic :17-30)."""

    def __init__(self, cfg, no_input_corr=False):
        super().__init__()
        self.no_input_corr = no_input_corr
        use_normals = cfg["model"]["use_normals"]
        if no_input_corr:
            has_features, inp_feat_dim = (True, 3) if use_normals else (False, 0)
        else:
            has_features, inp_feat_dim = (True, 7) if use_normals else (True, 4)
        self.encoder = encoder_dict[cfg["model"]["encoder"]](
            has_features=has_features, inp_feat_dim=inp_feat_dim, **cfg["model"]["encoder_kwargs"])
        self.decoder = decoder_dict[cfg["model"]["decoder"]](**cfg["model"]["decoder_kwargs"])

    @torch.no_grad()
    def geometry(self, points, surface_samples_inputs, training=None):
        """Every index set forward(points, surface_samples_inputs) derives from its two coordinate inputs alone (the encoder's
        sampling / grouping pyramid and neighbour sets, the queries' anchor neighbours, the inverse lists of the backward pass
        when ``training``), on the current stream.  forward(geometry=) takes it instead of searching: a host that knows the next
        batch computes this beside the current step (nsdp_amd.graph_step.PipelinedGeometry) -- FPS is a chain of ~600 dependent
        iterations that no batch size shortens.  ``points`` must be the very tensor forward() is then called with."""
        x = surface_samples_inputs[:, :, 0:3].contiguous() if self.no_input_corr else surface_samples_inputs
        if not (hasattr(self.encoder, "geometry") and hasattr(self.decoder, "geometry")):
            raise NotImplementedError("geometry(): this encoder / decoder pair searches inside its forward pass only")
        g = {"encoder": self.encoder.geometry(x, training)}
        g.update(self.decoder.geometry(points, g["encoder"]["anchors"]))
        g["query_points"] = points
        return g

    def encode(self, surface_samples_inputs, queries=None, geometry=None):
        """The encoding {'z', 'anchors', 'anchor_feats'} of a surface cloud -- the half of forward() that does not depend on
        the query points.  Callers that decode several query sets against ONE cloud (FlowArbitrary, the dense-inference step
        functions) encode once and call decode() per set; the reference re-runs the whole module each time.
        ``queries``: the points decode() will be called with next -- the decoder's anchor-only work is then launched beside the
        encoder's forward chain (CrossTransformerDecoder.prefetch, NSDP_DECODER_PREFETCH=0 switches it off)."""
        x = surface_samples_inputs[:, :, 0:3].contiguous() if self.no_input_corr else surface_samples_inputs
        if geometry is not None:
            enc = self.encoder(x, geometry=geometry["encoder"])
            enc["query_idx"], enc["query_points"] = geometry["query_idx"], geometry["query_points"]
            return enc
        if (DECODER_PREFETCH and queries is not None and queries.is_cuda and hasattr(self.decoder, "prefetch")
                and "on_anchors" in inspect.signature(self.encoder.forward).parameters):
            return self.encoder(x, on_anchors=lambda anchors, after: self.decoder.prefetch(queries, anchors, after))
        return self.encoder(x)

    def decode(self, points, encoding):
        return self.decoder(points, encoding)

    def forward(self, points, surface_samples_inputs, geometry=None):
        points = points if points.is_contiguous() else points.contiguous()
        return self.decoder(points, self.encode(surface_samples_inputs, queries=points, geometry=geometry))


def _loss_with_cano(model, data_dict, config, geometry=None):
    """forward + l2 loss of train_on_batch_with_cano (reference :66-72), as a tensor.  ``geometry``: model.geometry() of this
    batch's inputs, computed ahead of the step."""
    if geometry is not None:
        pred = model(data_dict["space_samples_src"], data_dict["surface_samples_inputs"], geometry=geometry)
    else:
        pred = model(data_dict["space_samples_src"], data_dict["surface_samples_inputs"])
    return compute_l2_error(pred, data_dict["space_samples_tgt"])


def _train_step_with_cano(model, optimizer, data_dict, config, geometry=None):
    """The step of train_on_batch_with_cano up to (not including) the host read-back of the loss: everything that is
    enqueued on the GPU.  This is what nsdp_amd.graph_step captures and replays."""
    optimizer.zero_grad()
    loss = _loss_with_cano(model, data_dict, config, geometry)
    loss.backward()
    optimizer.step()
    return loss


def train_on_batch_with_cano(model, optimizer, data_dict, config):
    """reference model/deformation_networks.py:63-77."""
    return _train_step_with_cano(model, optimizer, data_dict, config).item()


train_on_batch_with_cano.tensor_step = _train_step_with_cano
train_on_batch_with_cano.loss_fn = _loss_with_cano      # (data-parallel replay: two graphs around the gradient exchange)


@torch.no_grad()
def validate_on_batch_with_cano(model, data_dict, config):
    """reference model/deformation_networks.py:80-88."""
    pred = model(data_dict["space_samples_src"], data_dict["surface_samples_inputs"])
    return compute_l2_error(pred, data_dict["space_samples_tgt"]).item()


@torch.no_grad()
def test_on_batch_with_cano(model, data_dict, config, compute_loss=False):
    """Dense inference: surface samples, then all mesh vertices (reference :90-109)."""
    inputs = data_dict["surface_samples_inputs"]
    if ENCODE_ONCE:
        # the reference runs the module twice on the same surface input (:96, :101): same encoding both times
        encoding = model.encode(inputs)
        data_dict["surface_samples_tgt_pred"] = model.decode(data_dict["surface_samples_src"], encoding)
        deformed_verts = model.decode(data_dict["verts_src"], encoding)
    else:
        data_dict["surface_samples_tgt_pred"] = model(data_dict["surface_samples_src"], inputs)
        deformed_verts = model(data_dict["verts_src"], inputs)
    data_dict["verts_tgt_pred"] = deformed_verts
    if compute_loss:
        loss = compute_l2_error(deformed_verts, data_dict["verts_tgt"])
    else:
        loss = torch.zeros((1), dtype=torch.float32)
    return loss.item(), data_dict
