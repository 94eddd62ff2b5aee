"""TEST INFRASTRUCTURE ONLY (oracle): CPU restatement of the reference's per-sample tensor contract,
dataset/dataset_deform4d_flow.py:174-264 + dataset/utils.py:38-83 of tangjiapeng/NSDP (sub-sampling, handle mask,
masked flow, noise, [N,7] packing).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

Pinned: oracle/make_golden_dataset.py imports the reference's dataset/utils.py itself (`trimesh` stubbed: unused by
these functions) and stores its outputs in tests/golden/dataset_contract.npz; tests/test_harness_cpu.py checks this
restatement against them.
"""
import numpy as np


def subsample_surface_flow(num_surf_samples, cano, src, tgt, idxs=None):
    """dataset/utils.py:38-45 (random permutation prefix unless the indices are given)."""
    if idxs is None:
        idxs = np.random.permutation(cano.shape[0])[:num_surf_samples]
    return cano[idxs, :], src[idxs, :], tgt[idxs, :], idxs


def subsample_space_flow(num_space_samples, cano, src, tgt):
    """dataset/utils.py:47-54 (only when there are more samples than requested)."""
    if cano.shape[0] > num_space_samples:
        idxs = np.random.permutation(cano.shape[0])[:num_space_samples]
        cano, src, tgt = cano[idxs, :], src[idxs, :], tgt[idxs, :]
    return cano, src, tgt


def cano_sample_handle_mask(partial_range, cano, bbox_min, bbox_max):
    """dataset/utils.py:56-62: head (low y) | tail (high y) | feet (low z) slabs of the canonical bounding box."""
    head = cano[:, 1] < bbox_min[1] + partial_range
    tail = cano[:, 1] > bbox_max[1] - partial_range
    foot = cano[:, 2] < bbox_min[2] + partial_range
    return head | tail | foot


def create_partial_src(partial_shape_ratio, surface_samples_src, handle_sample_idx, num_seeds=5, seed_choice=None):
    """dataset/utils.py:79-101: carve `num_seeds` holes into the non-handle region -- seeds = a random permutation prefix of the
    non-handle samples (``seed_choice``: those positions, when given), each hole = the seed's int(hole_ratio * n // num_seeds)
    nearest surface samples (handles included; the reference asks scipy's KDTree) -- and return the indices that remain,
    ascending (the reference builds them through a Python set of small ints, which iterates in ascending order)."""
    n = len(surface_samples_src)
    if partial_shape_ratio >= 1.0:
        return np.arange(n)
    hole_ratio = 1.0 - partial_shape_ratio
    per_hole = int(hole_ratio * n // num_seeds)
    nonhandle = surface_samples_src[~handle_sample_idx]
    if seed_choice is None:
        seed_choice = np.random.permutation(nonhandle.shape[0])[:num_seeds]
    seeds = nonhandle[seed_choice].astype(np.float64)
    d2 = ((seeds[:, None, :] - surface_samples_src[None, :, :].astype(np.float64)) ** 2).sum(-1)
    remove = np.argsort(d2, axis=1, kind="stable")[:, :per_hole].reshape(-1)
    keep = np.ones(n, dtype=bool)
    keep[remove] = False
    return np.nonzero(keep)[0]


def sample_contract(cfg_data, data_cano, data_src, data_tgt, surf_idxs=None, noise=None):
    """dataset_deform4d_flow.py:190-246 for one sample (without the partial-shape branch, which no shipped config
    enables): returns the data_dict entries that feed the model."""
    if not cfg_data["arbitrary"] and cfg_data["inverse"]:
        data_src, data_tgt = data_tgt, data_src                                  # :195-199
    cano_full = data_cano["surface_samples"]
    bbox_min, bbox_max = cano_full.min(axis=0), cano_full.max(axis=0)            # :209 (before sub-sampling)
    cano, src, tgt, idxs = subsample_surface_flow(cfg_data["num_surf_samples"], cano_full,
                                                  data_src["surface_samples"], data_tgt["surface_samples"], surf_idxs)
    ncano, nsrc, ntgt, _ = subsample_surface_flow(cfg_data["num_surf_samples"], data_cano["surface_normals"],
                                                  data_src["surface_normals"], data_tgt["surface_normals"], idxs)
    mask = cano_sample_handle_mask(cfg_data["partial_range"], cano, bbox_min, bbox_max)
    tgt_masked = tgt * mask[:, None]                                             # :217
    if cfg_data["noise_level"] > 0.0:                                            # :219-220, utils.py:73-78
        if noise is None:
            noise = np.random.randn(*src.shape).astype(np.float32)
        src = src + cfg_data["noise_level"] * noise
    inputs = np.concatenate([src, tgt_masked, mask[:, None]], axis=1).astype(np.float32)   # :222-223
    out = {"surface_samples_cano": cano, "surface_samples_src": src, "surface_samples_tgt": tgt,
           "surface_normals_cano": ncano, "surface_normals_src": nsrc, "surface_normals_tgt": ntgt,
           "cano_handle_sample_idx": mask[:, None], "surface_samples_inputs": inputs}
    sc, ss, st = subsample_space_flow(cfg_data["num_space_samples"], data_cano["space_samples"],
                                      data_src["space_samples"], data_tgt["space_samples"])
    out.update({"space_samples_cano": sc, "space_samples_src": ss, "space_samples_tgt": st})
    return out
