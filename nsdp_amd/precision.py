"""Storage precision of the activations of the TDNet path.

``float32`` (default): every tensor fp32; the large dense layers run as error-compensated bf16x3 products (fp32 accuracy).
``bfloat16`` (BASELINE config 3): activations and everything saved for the backward pass are bf16 tensors, all
arithmetic accumulates in fp32 (MFMA accumulators, softmax, BatchNorm statistics), parameters / their gradients / the
optimizer state stay fp32 ("master weights"), coordinates, indices and the network output stay fp32.

    with precision.storage(torch.bfloat16):
        loss = train_on_batch(model, optimizer, data, config)
"""
from __future__ import annotations

import os

import torch

_NAMES = {"f32": torch.float32, "fp32": torch.float32, "float32": torch.float32,
          "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}
_storage = _NAMES[os.environ.get("NSDP_STORAGE", "f32")]


def storage_dtype() -> torch.dtype:
    return _storage


def is_bf16() -> bool:
    return _storage is torch.bfloat16


def set_storage(dtype):
    global _storage
    dtype = _NAMES[dtype] if isinstance(dtype, str) else dtype
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError(f"storage precision must be float32 or bfloat16, got {dtype}")
    _storage = dtype


class storage:
    """Context manager: ``with precision.storage(torch.bfloat16): ...``"""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self._was = _storage
        set_storage(self.dtype)
        return self

    def __exit__(self, *exc):
        set_storage(self._was)
        return False


# bf16 storage of FlowArbitrary: the first network's output POINTS are what the second network samples, groups and queries
# at -- its ~9e-3 output error in bf16 storage is amplified ~15x by those discrete selections (tools/bf16_bisect.py,
# profiles/r4_bf16_bisect.txt: 1.3e-1 against the reference, the same as the fp32 model's own response to bf16-rounded input
# coordinates; 1.0e-2 with network 1 in fp32 storage).  "f32" keeps network 1 (model_canonicalize) in fp32 storage (its
# dense layers on the bf16x3 kernels) and only network 2 in bf16: NSDP_BF16_NET1=f32 / set_canonicalize_f32(True).
# "dec32" is the middle point: network 1's ENCODER in bf16 storage (its output is one latent code + 100 anchor features per shape: a
# smooth function of the cloud), its DECODER -- whose per-point outputs are the coordinates network 2 searches -- in fp32 storage.
_net1_mode = os.environ.get("NSDP_BF16_NET1", "bf16")
if _net1_mode not in ("bf16", "f32", "dec32"):
    raise ValueError("NSDP_BF16_NET1 must be bf16, f32 or dec32")


def canonicalize_f32() -> bool:
    return _net1_mode == "f32"


def canonicalize_decoder_f32() -> bool:
    return _net1_mode == "dec32"


def canonicalize_mode() -> str:
    return _net1_mode


def set_canonicalize_f32(flag: bool):
    set_canonicalize_mode("f32" if flag else "bf16")


def set_canonicalize_mode(mode: str):
    global _net1_mode
    if mode not in ("bf16", "f32", "dec32"):
        raise ValueError("canonicalize mode must be bf16, f32 or dec32")
    _net1_mode = mode


def to_storage(t: torch.Tensor) -> torch.Tensor:
    """Cast a feature tensor to the storage precision (no-op when it already is)."""
    return t if t.dtype is _storage else t.to(_storage)
