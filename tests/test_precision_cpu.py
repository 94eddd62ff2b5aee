"""Host-side logic of the storage-precision switch and of the scatter-strategy selection (no GPU)."""
import pytest
import torch


def test_storage_switch_and_context_manager():
    from nsdp_amd import precision
    assert precision.storage_dtype() is torch.float32 and not precision.is_bf16()
    with precision.storage(torch.bfloat16):
        assert precision.is_bf16() and precision.storage_dtype() is torch.bfloat16
        with precision.storage("f32"):
            assert not precision.is_bf16()
        assert precision.is_bf16()
        x = torch.ones(3)
        assert precision.to_storage(x).dtype is torch.bfloat16
    assert not precision.is_bf16()
    with pytest.raises(ValueError):
        precision.set_storage(torch.float16)
    with pytest.raises(KeyError):
        precision.set_storage("fp8")
    assert not precision.is_bf16()


def test_context_manager_restores_on_exceptions():
    from nsdp_amd import precision
    with pytest.raises(RuntimeError):
        with precision.storage(torch.bfloat16):
            raise RuntimeError("boom")
    assert not precision.is_bf16()


def test_scatter_strategy_per_level():
    """Which scatter the attention backward uses at the levels of the forward.yaml encoder (csrc/attention.hip's
    lds_table_fits mirrored in hip_attention._use_inverse): inverse lists where the table fits neither LDS nor registers."""
    from nsdp_amd import hip_attention as ha
    f32, bf = torch.float32, torch.bfloat16
    # (n centres, N sources, d): begin block, set abstractions, down blocks
    assert ha._use_inverse(f32, False, 2048, 2048, 120)
    assert ha._use_inverse(bf, False, 500, 2048, 120)
    assert ha._use_inverse(f32, False, 500, 500, 120)
    assert ha._use_inverse(f32, False, 100, 500, 256)
    assert ha._use_inverse(f32, False, 100, 100, 256)            # n < 4 N: the LDS-table kernel does not apply
    assert not ha._use_inverse(f32, False, 8192, 100, 200)       # a 100-row table with many centres: LDS table
    assert not ha._use_inverse(f32, True, 8192, 100, 200)        # the decoder (one query vector per shape): register table
    # the decoder's anchor tables: scatter as a GEMM in both storage types (fp32: three bf16 planes x the exact one-hot operand)
    assert ha._onehot_ok(bf, True, 100, 200) and ha._onehot_ok(f32, True, 100, 200) == ha.ONEHOT_SCATTER_F32
    assert not ha._onehot_ok(bf, True, 130, 200) and not ha._onehot_ok(bf, False, 100, 200)
    assert not ha._onehot_ok(f32, True, 100, 256) and not ha._onehot_ok(f32, False, 100, 200)
