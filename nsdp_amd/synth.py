"""Procedural (torch-RNG-independent) synthetic inputs and weights for the TDNet hot path.

Everything is derived from a counter hash (splitmix64) evaluated in numpy uint64 arithmetic, so the
same (seed, name) gives bit-identical arrays on every machine and numpy version.  Used by
``bench.py`` (synthetic data, there is no dataset on the GPU box), by ``oracle/make_golden.py`` and
by the tests, which must all regenerate exactly the tensors the golden fixtures were made from.

Input layout follows the reference dataset contract:
``surface_samples_inputs[B,NS,7] = src xyz | mask * tgt xyz | mask``
(/root/reference/dataset/dataset_deform4d_flow.py:217-222), queries ``space_samples_src[B,NQ,3]``,
targets ``space_samples_tgt[B,NQ,3]``.
"""
from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _stream_key(seed: int, name: str) -> np.uint64:
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    k = np.array([(int(seed) & 0xFFFFFFFF) << 32 | h], dtype=np.uint64)
    return _splitmix64(k)[0]


def uniform01(seed: int, name: str, shape) -> np.ndarray:
    """float64 uniforms in [0,1) with 53 random bits, deterministic in (seed, name, flat index)."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    key = _stream_key(seed, name)
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + key) & _M64
    bits = _splitmix64(ctr) >> np.uint64(11)
    return (bits.astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(shape)


def uniform(seed: int, name: str, shape, lo: float, hi: float) -> np.ndarray:
    return (lo + (hi - lo) * uniform01(seed, name, shape)).astype(np.float32)


def normal(seed: int, name: str, shape) -> np.ndarray:
    """Box-Muller on two independent uniform streams (float32 result)."""
    u1 = uniform01(seed, name + "#u1", shape)
    u2 = uniform01(seed, name + "#u2", shape)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def make_batch(seed: int, batch: int, n_surf: int, n_query: int, handle_ratio: float = 0.3):
    """Synthetic batch in the reference's data_dict layout (numpy float32 arrays).

    surf xyz ~ U[-0.5,0.5)^3 (GAPS-normalised meshes live roughly in that cube,
    /root/reference/preprocess/others/process_mesh_local.sh:62-63); handle mask m = (u < ratio);
    inputs = [xyz, m*(xyz + 0.05 n1), m]; queries ~ U[-0.5,0.5)^3; targets = q + 0.01 n2.
    """
    xyz = uniform(seed, "surf_xyz", (batch, n_surf, 3), -0.5, 0.5)
    mask = (uniform01(seed, "handle_mask", (batch, n_surf, 1)) < handle_ratio).astype(np.float32)
    tgt = (xyz + 0.05 * normal(seed, "surf_disp", (batch, n_surf, 3))).astype(np.float32)
    inputs = np.concatenate([xyz, (mask * tgt).astype(np.float32), mask], axis=-1)
    q = uniform(seed, "space_src", (batch, n_query, 3), -0.5, 0.5)
    t = (q + 0.01 * normal(seed, "space_disp", (batch, n_query, 3))).astype(np.float32)
    return {
        "surface_samples_inputs": np.ascontiguousarray(inputs, dtype=np.float32),
        "space_samples_src": q,
        "space_samples_tgt": t,
    }


def procedural_state_dict(template: dict, seed: int) -> dict:
    """Deterministic values for every entry of a ``state_dict`` (keys + shapes from ``template``).

    Every weight is randomised -- including ``ResnetBlockFC.fc_1.weight`` that the reference
    zero-initialises (/root/reference/model/decoder/blocks.py:131) and the BatchNorm running
    statistics -- so that no gradient or eval-mode path is trivially zero.
    Returns numpy arrays (int64 scalar for ``num_batches_tracked``).
    """
    out = {}
    for key, ref in template.items():
        shape = tuple(ref.shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = np.zeros(shape, dtype=np.int64)
        elif leaf == "running_mean":
            out[key] = uniform(seed, key, shape, -0.1, 0.1)
        elif leaf == "running_var":
            out[key] = uniform(seed, key, shape, 0.5, 1.5)
        elif leaf == "weight" and len(shape) == 1:  # BatchNorm gamma
            out[key] = uniform(seed, key, shape, 0.5, 1.5)
        elif leaf == "bias":
            out[key] = uniform(seed, key, shape, -0.1, 0.1)
        elif leaf == "weight":
            fan_in = int(np.prod(shape[1:]))
            bound = float(np.sqrt(1.0 / max(fan_in, 1)))  # PyTorch-default-like scale: outputs O(1)
            out[key] = uniform(seed, key, shape, -bound, bound)
        else:
            raise KeyError(f"unhandled state_dict entry {key} {shape}")
    return out
