"""PointNet++ set-abstraction / feature-propagation modules on the MI355X ops -- same classes, constructor arguments,
channel-major ``(B, C, N)`` tensor contract and state_dict keys as the reference's
pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py, so code written against
``pointnet2_ops.pointnet2_modules`` runs unchanged with ``import nsdp_amd.pointnet2_modules``.

Sampling, ball query, grouping, 3-NN and interpolation are the hand-written HIP kernels behind
``nsdp_amd.pointnet2_utils``; the shared MLPs (1x1 Conv2d + BatchNorm2d + ReLU) run as channels-last rows through the
MFMA linear and the HIP batch-norm kernels (the modules keep nn.Conv2d / nn.BatchNorm2d objects only as parameter
holders, which is what makes the checkpoints interchangeable).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import hip_batchnorm, hip_linear, pointnet2_utils


def build_shared_mlp(mlp_spec: List[int], bn: bool = True) -> nn.Sequential:
    """pointnet2_modules.py:9-19: [Conv2d(1x1, bias = not bn), BatchNorm2d?, ReLU] per layer."""
    layers = []
    for i in range(1, len(mlp_spec)):
        layers.append(nn.Conv2d(mlp_spec[i - 1], mlp_spec[i], kernel_size=1, bias=not bn))
        if bn:
            layers.append(nn.BatchNorm2d(mlp_spec[i]))
        layers.append(nn.ReLU(True))
    return nn.Sequential(*layers)


def _run_shared_mlp(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """x: channels-last (..., C_in) -> (..., C_out) through the Conv2d/BN2d/ReLU parameter holders of `seq`."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        conv = mods[i]
        i += 1
        bn = None
        if i < len(mods) and isinstance(mods[i], nn.BatchNorm2d):
            bn = mods[i]
            i += 1
        relu = i < len(mods) and isinstance(mods[i], nn.ReLU)
        if relu:
            i += 1
        w = conv.weight.view(conv.out_channels, conv.in_channels)
        if bn is None:
            x = hip_linear.linear(x, w, conv.bias, relu_out=relu)
        else:
            x = hip_batchnorm.batch_norm(hip_linear.linear(x, w, conv.bias), bn, relu=relu)
    return x


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        """xyz (B,N,3), features (B,C,N) -> new_xyz (B,npoint,3), new_features (B, sum_k mlps[k][-1], npoint)
        (pointnet2_modules.py:29-76)."""
        new_features_list = []
        xyz_flipped = xyz.transpose(1, 2).contiguous()
        new_xyz = (pointnet2_utils.gather_operation(xyz_flipped, pointnet2_utils.furthest_point_sample(xyz, self.npoint))
                   .transpose(1, 2).contiguous() if self.npoint is not None else None)
        for grouper, mlp in zip(self.groupers, self.mlps):
            grouped = grouper(xyz, new_xyz, features)                               # (B, C, npoint, nsample)
            rows = _run_shared_mlp(mlp, grouped.permute(0, 2, 3, 1).contiguous())   # (B, npoint, nsample, C')
            new_features_list.append(rows.max(dim=2)[0].permute(0, 2, 1))           # max over the group
        return new_xyz, torch.cat(new_features_list, dim=1).contiguous()


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Set abstraction with multi-scale grouping (pointnet2_modules.py:79-118)."""

    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3                         # (in place, like the reference: the caller's list changes)
            self.mlps.append(build_shared_mlp(mlp_spec, bn))


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction (pointnet2_modules.py:121-152)."""

    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz)


class PointnetFPModule(nn.Module):
    """Feature propagation: inverse-distance 3-NN interpolation + shared MLP (pointnet2_modules.py:155-209)."""

    def __init__(self, mlp, bn=True):
        super().__init__()
        self.mlp = build_shared_mlp(mlp, bn=bn)

    def forward(self, unknown, known, unknow_feats, known_feats):
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        new_features = torch.cat([interpolated, unknow_feats], dim=1) if unknow_feats is not None else interpolated
        rows = _run_shared_mlp(self.mlp, new_features.permute(0, 2, 1).contiguous())   # (B, n, C')
        return rows.permute(0, 2, 1).contiguous()
