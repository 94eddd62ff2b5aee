"""Data-parallel gradient exchange on CPU: world_size 2, gloo (the RCCL path uses the same code)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _Toy(torch.nn.Module):
    """decoder.* parameters first in the bucket, one parameter that never receives a gradient."""

    def __init__(self):
        super().__init__()
        self.encoder = torch.nn.ModuleDict({"lin": torch.nn.Linear(5, 4), "unused": torch.nn.Linear(3, 3, bias=False)})
        self.decoder = torch.nn.ModuleDict({"lin": torch.nn.Linear(4, 2)})

    def forward(self, x):
        return self.decoder["lin"](torch.tanh(self.encoder["lin"](x)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nsdp_amd.parallel import GradAllReducer
        torch.manual_seed(0)
        model = _Toy()
        reducer = GradAllReducer(model, world)
        names = [n for n, _ in reducer.named]
        assert names[0].startswith("decoder.")                      # decoder bucket goes first
        assert reducer.split == sum(p.numel() for n, p in model.named_parameters() if n.startswith("decoder."))
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        g = torch.Generator().manual_seed(100)
        x_all = torch.randn(world * 6, 5, generator=g)
        y_all = torch.randn(world * 6, 2, generator=g)
        x, y = x_all[rank * 6:(rank + 1) * 6], y_all[rank * 6:(rank + 1) * 6]
        for _ in range(3):
            reducer.zero_grad()
            loss = ((model(x) - y) ** 2).mean()
            loss.backward()
            for (_, p), v in zip(reducer.named, reducer.views):
                assert p.grad.data_ptr() == v.data_ptr()              # grads accumulate in the flat buffer
            reducer.all_reduce_mean()
            opt.step()
        # single-process reference on the full batch
        torch.manual_seed(0)
        ref = _Toy()
        ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
        for _ in range(3):
            ropt.zero_grad()
            ((ref(x_all) - y_all) ** 2).mean().backward()
            ropt.step()
        err = max(float((a - b).abs().max()) for (_, a), (_, b) in
                  zip(sorted(model.named_parameters()), sorted(ref.named_parameters())))
        unused_zero = bool((model.encoder["unused"].weight.grad == 0).all())
        out.put((rank, err, unused_zero, reducer.nbytes))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_grad_allreduce_matches_single_process_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, err, unused_zero, nbytes in res:
        assert err < 1e-6, (rank, err)      # mean of per-rank mean-loss grads == full-batch grad
        assert unused_zero                   # parameters without gradient keep a zero slice on every rank
        assert nbytes == 4 * (5 * 4 + 4 + 9 + 4 * 2 + 2)


def test_flat_bucket_layout_for_tdnet():
    """Forward TDNet: 4 492 267 fp32 parameters = 17.97 MB exchanged per step, decoder first."""
    from helpers import model_cfg
    from nsdp_amd.model import build_model
    from nsdp_amd.parallel import GradAllReducer
    model, *_ = build_model(model_cfg("forward", [2048, 500, 100]))
    red = GradAllReducer(model, 1)
    assert red.nbytes == 4492267 * 4
    assert red.named[0][0].startswith("decoder.") and red.named[-1][0].startswith("encoder.")
    assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for (_, p), v in zip(red.named, red.views))
