"""Per-batch tensor contract of the reference's data loader, on the device.

The reference prepares every sample with numpy on the host (dataset/dataset_deform4d_flow.py:174-264 +
dataset/utils.py:38-83): random sub-sampling of the surface / space samples, the bounding-box handle mask, the masked
flow, optional noise and the ``[N,7]`` packing ``[src xyz | mask * tgt xyz | mask]``.  Here the raw samples of a
whole batch live on the GPU and one call produces the ``data_dict`` the model consumes -- index gathers go through
the HIP row-gather kernel, the rest is a handful of elementwise ops; nothing returns to the host.
The partial-shape branch (``partial_shape_ratio < 1``, dataset/utils.py:79-101: holes carved around random non-handle seeds)
runs on the device too (``create_partial_src``); no shipped configuration enables it, and the reference's own call site cannot
run (dataset_deform4d_flow.py:222 indexes the 2-D ``[N,7]`` array with three subscripts: IndexError) -- the function is pinned
against the imported reference FUNCTION, the call site does what that line evidently means (rows ``remain_idx`` of every array).
"""
from __future__ import annotations

import numpy as np
import torch

from . import pointnet2_utils


def load_npz_surface_flow(path):
    """dataset/utils.py:8-12 (the files hold fp16 / fp32 ``points`` and ``normals``)."""
    d = np.load(path)
    return d["points"].astype(np.float32), d["normals"].astype(np.float32)


def load_npz_space_flow(path):
    """dataset/utils.py:14-17."""
    return np.load(path)["points"].astype(np.float32)


def fix_coord_system(points: torch.Tensor) -> torch.Tensor:
    """dataset/utils.py:29-32: (x, y, z) -> (x, -z, y)."""
    return torch.stack([points[..., 0], -points[..., 2], points[..., 1]], dim=-1).contiguous()


def random_subset(batch: int, n_full: int, n_keep: int, device, generator=None) -> torch.Tensor:
    """Per-sample random permutation prefix (dataset/utils.py:41): int32 (batch, min(n_keep, n_full))."""
    keys = torch.rand(batch, n_full, device=device, generator=generator)
    return keys.argsort(dim=1)[:, :min(n_keep, n_full)].to(torch.int32).contiguous()


def _take(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    return pointnet2_utils.gather_rows(points.contiguous().float(), idx)


def cano_sample_handle_mask(partial_range: float, cano: torch.Tensor, bbox_min: torch.Tensor, bbox_max: torch.Tensor):
    """dataset/utils.py:56-62 for a batch: cano (B,n,3), bbox_* (B,3) -> bool (B,n)."""
    head = cano[..., 1] < bbox_min[:, None, 1] + partial_range
    tail = cano[..., 1] > bbox_max[:, None, 1] - partial_range
    foot = cano[..., 2] < bbox_min[:, None, 2] + partial_range
    return head | tail | foot


def create_partial_src(partial_shape_ratio: float, surface_samples_src: torch.Tensor, handle_sample_idx: torch.Tensor,
                       num_seeds: int = 5, seed_choice=None, generator=None) -> torch.Tensor:
    """dataset/utils.py:79-101 for a batch, on the device: bool (B,n), True where a surface sample REMAINS.
    surface_samples_src (B,n,3); handle_sample_idx bool (B,n).  Per sample: ``num_seeds`` seeds among the non-handle samples
    -- a random permutation prefix, or the given positions ``seed_choice`` (B,num_seeds) WITHIN the non-handle subsequence, as
    the reference's ``np.random.permutation(#non-handle)[:num_seeds]`` --, and around every seed its
    int(hole_ratio * n // num_seeds) nearest surface samples (handles included) are removed.  The reference asks a KD-tree for
    those k (81 at n = 2048, ratio 0.8: beyond nsdp_knn's k <= 64); here 5 x n squared distances in fp64 (what scipy computes in)
    and a top-k -- the same sets wherever distances are distinct.  ``remain_idx`` of sample b = ``mask[b].nonzero()`` (ascending,
    like the reference's)."""
    B, n, _ = surface_samples_src.shape
    dev = surface_samples_src.device
    if partial_shape_ratio >= 1.0:
        return torch.ones(B, n, dtype=torch.bool, device=dev)
    hole_ratio = 1.0 - partial_shape_ratio
    per_hole = int(hole_ratio * n // num_seeds)
    nonhandle = ~handle_sample_idx
    counts = nonhandle.sum(dim=1)
    if int(counts.min()) < num_seeds:
        raise ValueError("create_partial_src: fewer non-handle samples than hole seeds")
    if seed_choice is None:
        # a uniformly random num_seeds-subset (in random order) of each sample's non-handle positions
        keys = torch.rand(B, n, device=dev, generator=generator) + handle_sample_idx.to(torch.float32) * 2.0
        seed_abs = keys.argsort(dim=1)[:, :num_seeds]
    else:
        # position j within the non-handle subsequence -> absolute index: the first i with cumsum(nonhandle)[i] == j + 1
        rank = torch.cumsum(nonhandle.to(torch.int64), dim=1)
        want = torch.as_tensor(seed_choice, device=dev, dtype=torch.int64).reshape(B, num_seeds) + 1
        seed_abs = torch.searchsorted(rank, want)
    pts = surface_samples_src.to(torch.float64)
    seeds = torch.gather(pts, 1, seed_abs[..., None].expand(B, num_seeds, 3))
    d2 = ((seeds[:, :, None, :] - pts[:, None, :, :]) ** 2).sum(-1)                       # (B, num_seeds, n)
    remove = torch.topk(d2, per_hole, dim=2, largest=False).indices.reshape(B, -1)
    keep = torch.ones(B, n, dtype=torch.bool, device=dev)
    keep.scatter_(1, remove, False)
    return keep


def prepare_batch(cfg_data: dict, cano: dict, src: dict, tgt: dict, surf_idx=None, space_idx=None, noise=None,
                  generator=None, partial_seed_choice=None) -> dict:
    """``cano`` / ``src`` / ``tgt``: dicts of device tensors ``surface_samples`` (B,Nf,3), ``surface_normals`` (B,Nf,3),
    ``space_samples`` (B,Mf,3).  Returns the reference's data_dict entries (dataset_deform4d_flow.py:226-246),
    batched, on the device.  ``surf_idx`` / ``space_idx`` (int32) and ``noise`` may be supplied for reproducibility."""
    if not cfg_data["arbitrary"] and cfg_data["inverse"]:
        src, tgt = tgt, src                                                      # backward network: arbitrary -> canonical
    cano_full = cano["surface_samples"].float()
    B, nf, _ = cano_full.shape
    dev = cano_full.device
    bbox_min, bbox_max = cano_full.min(dim=1).values, cano_full.max(dim=1).values      # before sub-sampling (:209)
    if surf_idx is None:
        surf_idx = random_subset(B, nf, cfg_data["num_surf_samples"], dev, generator)
    out = {}
    for key in ("surface_samples", "surface_normals"):
        for name, d in (("cano", cano), ("src", src), ("tgt", tgt)):
            out[f"{key}_{name}"] = _take(d[key], surf_idx)
    mask = cano_sample_handle_mask(cfg_data["partial_range"], out["surface_samples_cano"], bbox_min, bbox_max)
    maskf = mask[..., None].float()
    if cfg_data["noise_level"] > 0.0:
        if noise is None:
            noise = torch.randn(out["surface_samples_src"].shape, device=dev, generator=generator)
        out["surface_samples_src"] = out["surface_samples_src"] + cfg_data["noise_level"] * noise
    out["cano_handle_sample_idx"] = mask[..., None]
    out["surface_samples_inputs"] = torch.cat([out["surface_samples_src"], out["surface_samples_tgt"] * maskf, maskf],
                                              dim=-1).contiguous()
    if cfg_data.get("partial_shape_ratio", 1.0) < 1.0:
        # dataset_deform4d_flow.py:220-226: every per-sample array keeps the rows that remain.  Holes of different samples overlap
        # differently, so the remaining counts differ: as in the reference (whose default collate cannot stack ragged samples)
        # this is a batch-of-one path unless the counts happen to agree
        keep = create_partial_src(cfg_data["partial_shape_ratio"], out["surface_samples_src"], mask, seed_choice=partial_seed_choice,
                                  generator=generator)
        kept = keep.sum(dim=1)
        if int(kept.min()) != int(kept.max()):
            raise ValueError("partial shapes: the samples of this batch keep different numbers of points "
                             f"({int(kept.min())} .. {int(kept.max())}); prepare them one at a time")
        rows = keep.nonzero()[:, 1].reshape(B, -1).to(torch.int32).contiguous()                  # ascending per sample
        for key in ("surface_samples", "surface_normals"):
            for name in ("cano", "src", "tgt"):
                out[f"{key}_{name}"] = _take(out[f"{key}_{name}"], rows)
        out["surface_samples_inputs"] = _take(out["surface_samples_inputs"], rows)
        out["cano_handle_sample_idx"] = torch.gather(mask, 1, rows.long())[..., None]
        out["partial_remain_idx"] = rows
    mf = cano["space_samples"].shape[1]
    if mf > cfg_data["num_space_samples"]:
        if space_idx is None:
            space_idx = random_subset(B, mf, cfg_data["num_space_samples"], dev, generator)
        for name, d in (("cano", cano), ("src", src), ("tgt", tgt)):
            out[f"space_samples_{name}"] = _take(d["space_samples"], space_idx)
    else:
        for name, d in (("cano", cano), ("src", src), ("tgt", tgt)):
            out[f"space_samples_{name}"] = d["space_samples"].float().contiguous()
    return out
