"""ctypes binding of libnsdp_hip.so (the C-ABI declared in include/nsdp_hip.h).

No torch types cross the boundary: tensors are passed as raw device pointers + sizes and the current
HIP stream handle.  There is NO fallback: if the shared library is missing the import of any op that
needs it raises, and every op refuses non-GPU tensors (the reference asserts the same way on CPU
tensors, _ext-src/src/sampling.cpp:82-84).
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "libnsdp_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "nsdp_hip.h")

_lib = None


class NsdpHipError(RuntimeError):
    pass


def declared_symbols() -> list[str]:
    """Every function include/nsdp_hip.h declares (used by the CPU test that checks the exports)."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(nsdp_[a-z0-9_]+)\s*\(", text)))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise NsdpHipError(
                f"{SO_PATH} is missing: build it with `python -m nsdp_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the hot path.")
        _lib = ctypes.CDLL(SO_PATH)
        _lib.nsdp_last_error.restype = ctypes.c_char_p
        for key, env in ((1, "NSDP_WGRAD_PIPE"), (3, "NSDP_NT_VARIANT"), (6, "NSDP_X3_DBG"), (7, "NSDP_WG16_DBG"), (8, "NSDP_LIN16_DBG"), (10, "NSDP_KNN_QUEUE"), (11, "NSDP_BN_SLAB"), (12, "NSDP_SEARCH_QUAD")):  # kernel-variant knobs for A/B runs
            if os.environ.get(env) is not None:
                _lib.nsdp_debug_set(key, int(os.environ[env]))
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().nsdp_last_error().decode("utf-8", "replace")
        raise NsdpHipError(f"{what} failed (status {rc}): {msg}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_NO_GUARD = contextlib.nullcontext()


def stream_ptr() -> ctypes.c_void_p:
    """The current HIP stream of the current device as a raw handle (one C call: the step launches ~500 kernels,
    torch.cuda.current_stream() alone was 15 % of the host time of a small-batch step)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require(t: torch.Tensor, name: str, dtype):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise NsdpHipError(f"{name} must be a GPU tensor (CPU not supported, no fallback)")
    if t.dtype != dtype:
        raise NsdpHipError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise NsdpHipError(f"{name} must be a contiguous tensor")


def fptr(t: torch.Tensor, name: str = "tensor") -> ctypes.c_void_p:
    _require(t, name, torch.float32)
    return ctypes.c_void_p(t.data_ptr())


def hptr(t: torch.Tensor, name: str = "tensor") -> ctypes.c_void_p:
    """bf16 tensor (the storage precision of BASELINE config 3)."""
    _require(t, name, torch.bfloat16)
    return ctypes.c_void_p(t.data_ptr())


def opthptr(t, name: str = "tensor"):
    return ctypes.c_void_p(0) if t is None else hptr(t, name)


def iptr(t: torch.Tensor, name: str = "tensor") -> ctypes.c_void_p:
    _require(t, name, torch.int32)
    return ctypes.c_void_p(t.data_ptr())


def optptr(t):
    return ctypes.c_void_p(0) if t is None else ctypes.c_void_p(t.data_ptr())


def on_device(t: torch.Tensor):
    """Context manager selecting the tensor's GPU; refuses CPU tensors (no fallback)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NsdpHipError("expected a GPU tensor (CPU not supported, no fallback)")
    if t.device.index == torch.cuda.current_device():      # one process per GPU: the usual case, no guard needed
        return _NO_GUARD
    return torch.cuda.device(t.device)
