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


def to_storage(t: torch.Tensor) -> torch.Tensor:
    """Cast a feature tensor to the storage precision (no-op when it already is)."""
    return t if t.dtype is _storage else t.to(_storage)
