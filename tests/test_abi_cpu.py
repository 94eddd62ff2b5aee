"""CPU-only checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/nsdp_hip.h declares; the Python mirror refuses CPU tensors loudly (no fallback)."""
import ctypes
import os

import pytest
import torch

from nsdp_amd import _lib


@pytest.fixture(scope="module")
def so():
    if not os.path.exists(_lib.SO_PATH):
        from nsdp_amd import build
        build.build()
    return ctypes.CDLL(_lib.SO_PATH)


def test_header_declares_expected_entry_points():
    names = _lib.declared_symbols()
    for must in ["nsdp_furthest_point_sampling", "nsdp_gather_points", "nsdp_gather_points_grad",
                 "nsdp_group_points", "nsdp_group_points_grad", "nsdp_ball_query", "nsdp_three_nn",
                 "nsdp_three_interpolate", "nsdp_three_interpolate_grad", "nsdp_knn"]:
        assert must in names


def test_library_exports_every_declared_symbol(so):
    missing = [n for n in _lib.declared_symbols() if not hasattr(so, n)]
    assert not missing, missing
    assert so.nsdp_abi_version() >= 1


def test_bad_arguments_return_status_not_exit(so):
    so.nsdp_last_error.restype = ctypes.c_char_p
    rc = so.nsdp_knn(None, None, 1, 4, 4, 2, None, None, None)
    assert rc == -1
    assert b"null" in so.nsdp_last_error()
    rc = so.nsdp_furthest_point_sampling(None, 0, 16, 4, None, None, None)
    assert rc == 0  # empty batch is a no-op, like the reference's zero-size launch


def test_python_mirror_refuses_cpu_tensors():
    from nsdp_amd import pointnet2_utils as pu
    xyz = torch.zeros(1, 16, 3)
    with pytest.raises(RuntimeError):
        pu.furthest_point_sample(xyz, 4)
    with pytest.raises(RuntimeError):
        pu.knn(xyz, xyz, 4)
