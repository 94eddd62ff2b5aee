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


class _ToyDecoder(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(4, 2)
        self.q = torch.nn.Linear(3, 2, bias=False)

    def forward(self, points, encoding):
        return self.lin(encoding["anchor_feats"]) * encoding["z"] + self.q(points)


class _ToyNet(torch.nn.Module):
    """The TDNet's shape: an encoder whose outputs (a dict) and the query points are the decoder's inputs."""

    def __init__(self):
        super().__init__()
        self.encoder = torch.nn.Linear(5, 4)
        self.mid = torch.nn.Linear(4, 2)
        self.decoder = _ToyDecoder()

    def forward(self, points, x):
        f = torch.tanh(self.encoder(x))
        return self.decoder(points, {"z": self.mid(f), "anchor_feats": f, "anchors": x.detach()})


def _worker_overlap(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nsdp_amd.parallel import GradAllReducer
        g = torch.Generator().manual_seed(100)
        x_all, p_all, y_all = (torch.randn(world * 6, n, generator=g) for n in (5, 3, 2))
        sl = slice(rank * 6, (rank + 1) * 6)
        torch.manual_seed(rank)                   # ranks start DIFFERENT ...
        model = _ToyNet()
        for p in model.parameters():              # ... and are made equal by a broadcast, as the harness does
            dist.broadcast(p.data, src=0)
        reducer = GradAllReducer(model, world)
        assert reducer.named[0][0].startswith("decoder.") and 0 < reducer.split < reducer.flat.numel()
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        order = []
        orig_start = reducer.start
        reducer.start = lambda i: (order.append(("start", i)), orig_start(i))[1]
        orig_tail = reducer.backward_tail
        reducer.backward_tail = lambda: (order.append(("tail",)), orig_tail())[1]
        # reference on this rank: the same model stepped with ONE backward pass and the whole exchange after it
        ref = _ToyNet()
        ref.load_state_dict(model.state_dict())
        ref_red = GradAllReducer(ref, world)
        ref_opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
        for _ in range(3):
            reducer.zero_grad(two_pass=True)
            loss = ((model(p_all[sl], x_all[sl]) - y_all[sl]) ** 2).mean()
            reducer.backward(loss)
            assert reducer._pending[0] is not None                   # bucket 0 is in flight when backward() returns
            reducer.finish()
            opt.step()
            ref_red.zero_grad()
            ((ref(p_all[sl], x_all[sl]) - y_all[sl]) ** 2).mean().backward()
            ref_red.all_reduce_mean()
            ref_opt.step()
        # bucket 0's collective was enqueued BEFORE the encoder's backward pass, every step
        # (finish() calls start() again for both buckets: no-ops once started)
        tails = [i for i, e in enumerate(order) if e == ("tail",)]
        assert len(tails) == 3 and all(order[i - 1] == ("start", 0) and order[i + 1] == ("start", 1) for i in tails), order
        assert reducer.enqueued_before_backward_returned == 3
        same_as_one_pass = all(torch.equal(a, b) for a, b in zip(model.parameters(), ref.parameters()))
        w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        lo, hi = w.clone(), w.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        # single-process run on the full batch
        full = _ToyNet()
        torch.manual_seed(0)
        full = _ToyNet()
        fopt = torch.optim.Adam(full.parameters(), lr=1e-2)
        for _ in range(3):
            fopt.zero_grad()
            ((full(p_all, x_all) - y_all) ** 2).mean().backward()
            fopt.step()
        err = max(float((a - b).abs().max()) for a, b in zip(model.parameters(), full.parameters()))
        out.put((rank, same_as_one_pass, bool(torch.equal(lo, hi)), err))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_pass_backward_issues_the_decoder_bucket_before_the_encoder_backward_world2():
    """GradAllReducer.backward: d(loss) / d(decoder inputs) first, bucket 0's all-reduce enqueued, then the encoder's backward,
    then bucket 1 -- asserted by the ORDER of the calls, every step; gradients and weights equal the one-pass step's bit for bit,
    ranks end bit-identical, and the trajectory is the single-process full-batch one."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker_overlap, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, same_as_one_pass, in_sync, err in res:
        assert same_as_one_pass, rank
        assert in_sync, rank
        assert err < 1e-6, (rank, err)


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


# ---------------------------------------------------------------------------------------------------------
# The product's dense layers do NOT return weight gradients through autograd: hip_linear publishes them to
# `param.grad` itself -- immediately inside backward (small batches) or from an end-of-backward engine callback after the
# side stream joined (large batches): `param.grad = g` when there is none, `param.grad.add_(g)` otherwise.  The mock
# below reproduces exactly that publication protocol on CPU (same statements as hip_linear._LinearFn.backward /
# hip_linear._publish), so the interplay with the flat bucket is tested at world size 2 without a GPU.
# ---------------------------------------------------------------------------------------------------------
class _DirectLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, w_param, b_param, deferred):
        ctx.save_for_backward(x, w)
        ctx.w_param, ctx.b_param, ctx.deferred = w_param, b_param, deferred
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        gw, gb = dy.t() @ x, dy.sum(0)

        def publish():
            with torch.no_grad():
                for prm, g in ((ctx.w_param, gw), (ctx.b_param, gb)):
                    prm.grad = g if prm.grad is None else prm.grad.add_(g)
        if ctx.deferred:
            torch.autograd.Variable._execution_engine.queue_callback(publish)
        else:
            publish()
        return dy @ w, None, None, None, None, None


class _DirectToy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = torch.nn.ModuleDict({"lin": torch.nn.Linear(5, 4)})
        self.decoder = torch.nn.ModuleDict({"lin": torch.nn.Linear(4, 2)})

    def forward(self, x, direct=True):
        e, d = self.encoder["lin"], self.decoder["lin"]
        if not direct:
            return d(torch.tanh(e(x)))
        h = torch.tanh(_DirectLinearFn.apply(x, e.weight.detach(), e.bias.detach(), e.weight, e.bias, True))
        return _DirectLinearFn.apply(h, d.weight.detach(), d.bias.detach(), d.weight, d.bias, False)


def _train_on_batch(model, optimizer, data, config):
    """Shape of the reference's train_on_batch_with_cano: its own optimizer.zero_grad() (set_to_none) included."""
    optimizer.zero_grad()
    loss = ((model(data["x"]) - data["y"]) ** 2).mean()
    loss.backward()
    optimizer.step()
    return loss.item()


def _worker_direct(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nsdp_amd.parallel import GradAllReducer, data_parallel_step
        g = torch.Generator().manual_seed(7)
        x_all, y_all = torch.randn(world * 6, 5, generator=g), torch.randn(world * 6, 2, generator=g)
        data = {"x": x_all[rank * 6:(rank + 1) * 6], "y": y_all[rank * 6:(rank + 1) * 6]}
        torch.manual_seed(0)
        ref = _DirectToy()
        ropt = torch.optim.Adam(ref.parameters(), lr=1e-2)
        for _ in range(3):
            ropt.zero_grad()
            ((ref(x_all, direct=False) - y_all) ** 2).mean().backward()
            ropt.step()
        errs = {}
        # (a) reducer.zero_grad() protocol: direct publication adds into the attached flat views
        torch.manual_seed(0)
        model = _DirectToy()
        red = GradAllReducer(model, world)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        for _ in range(3):
            red.zero_grad()
            ((model(data["x"]) - data["y"]) ** 2).mean().backward()
            assert red.adopt_grads() == 0                 # nothing strayed from the flat buffer
            red.all_reduce_mean()
            opt.step()
        errs["views"] = max(float((a - b).abs().max()) for a, b in zip(model.parameters(), ref.parameters()))
        # (b) the shipped step functions call optimizer.zero_grad() (grads -> None): wrapped, they must still reduce
        torch.manual_seed(0)
        model = _DirectToy()
        red = GradAllReducer(model, world)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        step = data_parallel_step(_train_on_batch, red)
        for _ in range(3):
            step(model, opt, data, None)
        errs["wrapped"] = max(float((a - b).abs().max()) for a, b in zip(model.parameters(), ref.parameters()))
        assert not opt._optimizer_step_pre_hooks           # the wrapper removes its hook
        # (c) stale-buffer hazard: without re-attachment the flat buffer would still hold the previous step
        opt.zero_grad()
        ((model(data["x"]) - data["y"]) ** 2).mean().backward()
        assert all(p.grad.data_ptr() != v.data_ptr() for (_, p), v in zip(red.named, red.views))
        assert red.adopt_grads() == len(red.named)
        assert all(p.grad.data_ptr() == v.data_ptr() for (_, p), v in zip(red.named, red.views))
        out.put((rank, errs))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_direct_grad_publication_with_flat_bucket_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker_direct, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, errs in res:
        assert errs["views"] < 1e-6 and errs["wrapped"] < 1e-6, (rank, errs)


# ------------------------------------------------------------------------------------------------
# the train harness as a data-parallel job (nsdp_amd.train.fit(dp=...)), world 2 over gloo
# ------------------------------------------------------------------------------------------------
class _BnToy(torch.nn.Module):
    """A model with a BatchNorm (per-rank running statistics) in front of a decoder."""

    def __init__(self):
        super().__init__()
        self.encoder = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.BatchNorm1d(4))
        self.decoder = torch.nn.Linear(4, 2)

    def forward(self, x):
        return self.decoder(torch.tanh(self.encoder(x)))


def _toy_fns():
    def train_on_batch(model, optimizer, sample, config):
        optimizer.zero_grad()
        loss = ((model(sample["x"]) - sample["y"]) ** 2).mean()
        loss.backward()
        optimizer.step()
        return loss.item()

    @torch.no_grad()
    def validate_on_batch(model, sample, config):
        return ((model(sample["x"]) - sample["y"]) ** 2).mean().item()
    return train_on_batch, validate_on_batch


def _toy_batches(n, seed):
    g = torch.Generator().manual_seed(seed)
    return [{"x": torch.randn(6, 5, generator=g), "y": torch.randn(6, 2, generator=g)} for _ in range(n)]


_FIT_CFG = {"training": {"epochs": 4, "save_frequency": 2, "optimizer": "Adam", "lr": 1e-2, "lr_step": 3, "lr_decay": 0.5,
                         "weight_decay": 0.0},
            "validation": {"frequency": 2}}


class _GeneratedBatches:
    """A loader WITHOUT a shard() method that builds its batches as it is walked and counts how many are alive at once: the
    every-world-th fallback of DataParallel.shard must hold one batch at a time, not an epoch of them."""

    def __init__(self, n, seed):
        self.n, self.seed = n, seed

    def __iter__(self):
        for b in _toy_batches(self.n, self.seed):
            yield b


def _fit_worker_n(rank, world, port, directory, n_batches, out):
    import argparse
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nsdp_amd import train
        from nsdp_amd.model import optimizer_factory
        from nsdp_amd.parallel import DataParallel
        torch.manual_seed(1000 + rank)            # DIFFERENT initial weights per rank: the broadcast must fix that
        model = _BnToy()
        sched, opt = optimizer_factory(_FIT_CFG["training"], model.parameters())
        dp = DataParallel(model, rank, world)
        import types
        assert isinstance(dp.shard(_GeneratedBatches(n_batches, 7)), types.GeneratorType)      # lazy: nothing materialised
        args = argparse.Namespace(continue_from_epoch=0, best_val_loss=float("inf"))
        lines = []
        hist = train.fit(model, _toy_fns(), sched, opt, _GeneratedBatches(n_batches, 7), _toy_batches(world, 8), _FIT_CFG,
                         directory, args, "cpu", log=lines.append, dp=dp)
        sync = dp.in_sync(model)
        out.put((rank, hist, sync, len(lines), {k: v.numpy().copy() for k, v in model.state_dict().items()}, args.best_val_loss))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("world", [2, 4])
def test_fit_as_a_data_parallel_job_keeps_ranks_identical_and_writes_one_set_of_files(tmp_path, world):
    """train.fit(dp=DataParallel) as a 2- and a 4-rank job over gloo: per-rank batches taken lazily from a plain iterable (2 *
    world + 1 batches -> 2 steps per epoch and rank, the odd one dropped), gradient mean before every optimizer step, rank-0
    files only, rank-0 BatchNorm buffers in the checkpoints and on every rank at validation time -- and the weights equal a
    single-process run over all ranks' batches with averaged gradients."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    d = str(tmp_path)
    n_batches = 2 * world + 1
    procs = [ctx.Process(target=_fit_worker_n, args=(r, world, port, d, n_batches, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=200) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, h0, s0, n0, sd0, b0) = res[0]
    assert all(r[2] for r in res)                            # bit-identical parameters on all ranks
    assert all(r[1] == h0 and r[5] == b0 for r in res)       # same (rank-averaged) losses, same best-model decisions
    assert [h[0] for h in h0] == ["train", "train", "train", "val", "train"]
    assert n0 > 0 and all(r[3] == 0 for r in res[1:])        # only rank 0 logs
    for r in res[1:]:
        for k in sd0:
            if "running" in k or "num_batches" in k:
                continue                                     # per-rank statistics between broadcasts
            assert (sd0[k] == r[4][k]).all(), k
    files = sorted(os.listdir(d))
    assert [f for f in files if f.startswith("model_")] == ["model_00000", "model_00002"]
    assert [f for f in files if f.startswith("opt_")] == ["opt_00000", "opt_00002"]
    assert len([f for f in files if f.startswith("modelbest_")]) == 1
    ck = torch.load(os.path.join(d, "model_00002"))
    # single-process reference: rank 0's initial weights, every rank's batches, mean of the `world` gradients per step
    import copy
    from nsdp_amd.model import optimizer_factory
    from nsdp_amd.model.learningrate import adjust_learning_rate
    torch.manual_seed(1000)
    ref = _BnToy()
    sched, opt = optimizer_factory(_FIT_CFG["training"], ref.parameters())
    batches = _toy_batches(n_batches, 7)
    twins = [ref] + [copy.deepcopy(ref) for _ in range(world - 1)]      # the ranks' replicas (own BatchNorm statistics)
    # a bias in front of a BatchNorm has an analytically zero gradient: what arrives is rounding noise of ~1e-9, which Adam
    # normalises into steps of +-lr whose signs follow the summation order of the all-reduce -- not comparable, not meaningful
    noise = ("encoder.0.bias", "encoder.1.running_mean")      # (the batch mean carries that bias)
    for epoch in range(4):
        adjust_learning_rate(sched, opt, epoch)
        for g in range(2):
            grads = []
            for r, m in enumerate(twins):
                b = batches[world * g + r]
                m.zero_grad()
                ((m(b["x"]) - b["y"]) ** 2).mean().backward()
                grads.append([p.grad.clone() for p in m.parameters()])
            for i, p in enumerate(ref.parameters()):
                p.grad = sum(gr[i] for gr in grads) / world
            opt.step()
            with torch.no_grad():
                for m in twins[1:]:
                    for p, q in zip(ref.parameters(), m.parameters()):
                        q.copy_(p)
        if epoch == 2:
            for k, v in ref.state_dict().items():
                if k not in noise:
                    assert torch.allclose(ck[k].to(v.dtype), v, rtol=1e-5, atol=1e-6), k
    for (k, v) in ref.named_parameters():
        if k not in noise:
            assert torch.allclose(torch.from_numpy(sd0[k]), v.detach(), rtol=1e-5, atol=1e-6), k
