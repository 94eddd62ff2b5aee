"""One real RCCL data point on the single GPU of the test box: a world-size-1 `nccl` (= RCCL on ROCm) communicator, the
flat-bucket gradient exchange of the real TDNet (nsdp_amd/parallel.py) run IN PLACE on the two bucket halves the 8-GPU
job will use.  The sum over one rank is the identity, so the gradients must stay bit-equal to what the backward pass
left in the bucket -- what this test proves is that librccl loads, a communicator comes up on the device, and the
collective accepts exactly these buffers (views of one flat tensor written by the weight-gradient kernels on the side
stream).  The reference has no counterpart (train.py:74-75 is single-device)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from helpers import build_product, model_cfg, to_dev
from nsdp_amd import synth

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def rccl_world1():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    saved = {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(DEV)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    try:
        yield
    finally:
        dist.destroy_process_group()
        for k, v in saved.items():      # (later tests start subprocesses: they must not inherit this rendezvous)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_rccl_all_reduce_of_the_flat_gradient_bucket_world1(rccl_world1):
    from nsdp_amd.model.utils import compute_l2_error
    from nsdp_amd.parallel import GradAllReducer
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    cfg = model_cfg("forward", [2048, 500, 100])
    data = to_dev(synth.make_batch(21, 16, 2048, 8192), DEV)      # 131 072 output rows: weight gradients on the side stream
    model, _, _ = build_product(cfg, 21, DEV)
    model.train()
    red = GradAllReducer(model, 1, always_exchange=True)
    assert 0 < red.split < red.flat.numel()                        # two buckets: decoder first, then the encoder
    red.zero_grad()
    compute_l2_error(model(data["space_samples_src"], data["surface_samples_inputs"]), data["space_samples_tgt"]).backward()
    before = red.flat.clone()
    assert float(before.abs().sum()) > 0
    red.all_reduce_mean()                                          # ncclAllReduce x2, in place, async + wait
    torch.cuda.synchronize()
    assert torch.equal(red.flat, before)
    for (_, p), v in zip(red.named, red.views):
        assert p.grad.data_ptr() == v.data_ptr()
    # a raw collective on the same buffer with a non-trivial reduction, to see the kernel really ran on the device:
    # MAX over one rank of (x) is x; PRODUCT as well; AVG divides by 1 -- run them all on the bucket halves
    for op in (dist.ReduceOp.MAX, dist.ReduceOp.AVG):
        dist.all_reduce(red.flat[:red.split], op=op)
        dist.all_reduce(red.flat[red.split:], op=op)
    torch.cuda.synchronize()
    assert torch.equal(red.flat, before)
    # the collective's result feeds the optimizer exactly like the plain path's gradients
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    w0 = next(model.decoder.parameters()).detach().clone()
    opt.step()
    assert not torch.equal(w0, next(model.decoder.parameters()).detach())


def test_rccl_broadcast_and_barrier_world1(rccl_world1):
    """The other two collectives a data-parallel job issues (initial weight broadcast, barrier around timed regions)."""
    t = torch.arange(1 << 20, dtype=torch.float32, device=DEV)
    dist.broadcast(t, src=0)
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    assert float(t[-1]) == float((1 << 20) - 1)


def test_two_graph_step_around_the_exchange_with_pipelined_geometry_equals_the_eager_loop(rccl_world1):
    """GraphedTrainOnBatch(reducer=, pipeline_geometry=): [zero the bucket, next batch's index sets beside forward + loss +
    backward, hand-over] and [optimizer step] as two replayed graphs around the one-rank RCCL all-reduce -- the data-parallel
    form of the pipelined step.  Losses equal the plain eager loop's, step for step (the exchange over one rank is the identity)."""
    from helpers import nondeterministic_knobs
    if nondeterministic_knobs():
        pytest.skip("the step is not bit-reproducible under " + ", ".join(nondeterministic_knobs()))
    from nsdp_amd import train
    from nsdp_amd.graph_step import GraphedTrainOnBatch, capturable_adam
    from nsdp_amd.model import build_model, optimizer_factory
    from nsdp_amd.parallel import GradAllReducer
    cfg = model_cfg("forward", [256, 64, 16])
    cfg["training"] = {"optimizer": "Adam", "lr": 5e-4, "lr_step": 100, "lr_decay": 0.1, "weight_decay": 0.0}
    batches = [{k: v.to(DEV) for k, v in b.items()} for b in train.SyntheticLoader(5, 3, 2, n_surf=256, n_query=128)]
    seq = [batches[i % 3] for i in range(7)]
    losses = []
    for mode in ("eager", "piped", "piped3"):      # piped3: the backward cut at the decoder's inputs, head / tail / update graphs
        train.seed_everything(31)
        model, train_fn, _, _ = build_model(cfg, device=DEV)
        model.train()
        _, opt = optimizer_factory(cfg["training"], model.parameters())
        capturable_adam(opt)
        fn = train_fn
        if mode != "eager":
            red = GradAllReducer(model, 1, always_exchange=True)
            fn = GraphedTrainOnBatch(train_fn, reducer=red, overlap=(mode == "piped3"),
                                     pipeline_geometry=lambda d: (d["space_samples_src"], d["surface_samples_inputs"]))
        out = []
        for i, b in enumerate(seq):
            if mode != "eager":
                out.append(fn(model, opt, b, cfg, next_data_dict=seq[i + 1] if i + 1 < len(seq) else None))
            else:
                out.append(fn(model, opt, b, cfg))
        losses.append(out)
        if mode != "eager":
            assert fn.replays == len(seq) - 1 and fn._pipe is not None and getattr(fn, "unannounced", 0) == 0
            assert (fn._tail is not None) == (mode == "piped3")
    assert losses[0] == losses[1] == losses[2], losses


@pytest.mark.parametrize("mtype,B,npl,ns,nq", [("forward", 2, [256, 64, 16], 256, 128), ("forward", 16, [2048, 500, 100], 2048, 8192),
                                             ("arbitrary", 2, [256, 64, 16], 256, 128)])
def test_two_pass_backward_equals_the_one_pass_backward_and_issues_bucket_0_in_between(rccl_world1, mtype, B, npl, ns, nq):
    """GradAllReducer.backward on the real networks: the forward pass cut at the (last) decoder's inputs, loss.backward() down to
    the cut, bucket 0's RCCL all-reduce issued, the encoder's backward from the cut, bucket 1 -- every gradient bit-equal to the
    one-pass backward's (the cut changes no arithmetic), the early bucket = exactly that decoder's parameters, and its
    collective in flight before the second pass starts.  FlowArbitrary: the cut also takes the query points, which are the
    first network's predictions (reference model/flow_arbitrary.py:19-27)."""
    from helpers import nondeterministic_knobs
    if nondeterministic_knobs():
        pytest.skip("the step is not bit-reproducible under " + ", ".join(nondeterministic_knobs()))
    from nsdp_amd.parallel import GradAllReducer
    cfg = model_cfg(mtype, npl)
    data = to_dev(synth.make_batch(23, B, ns, nq), DEV)
    model, train_fn, _ = build_product(cfg, 23, DEV)
    model.train()
    red = GradAllReducer(model, 1, always_exchange=True)
    prefix = "model_deform.decoder." if mtype == "arbitrary" else "decoder."
    assert all(n.startswith(prefix) for n, _ in red.named[:3]) and 0 < red.split < red.flat.numel()
    assert red.split == sum(p.numel() for n, p in model.named_parameters() if n.startswith(prefix))
    snap = {k: v.detach().clone() for k, v in model.state_dict().items()}      # (train-mode BatchNorm moves its buffers)

    def grads(two_pass):
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(snap[k])
        red.zero_grad(two_pass=two_pass)
        loss = train_fn.loss_fn(model, data, cfg)
        order = []
        if two_pass:
            assert red.backward_head(loss)
            red.start(0)
            order.append(red._pending[0] is not None)
            red.backward_tail()
            red.start(1)
        else:
            loss.backward()
        red.finish()
        torch.cuda.synchronize()
        return loss.detach().clone(), red.flat.clone(), order
    l1, g1, _ = grads(False)
    l2, g2, order = grads(True)
    assert order == [True]
    assert torch.equal(l1, l2)
    assert float(g1[:red.split].abs().sum()) > 0 and float(g1[red.split:].abs().sum()) > 0
    assert torch.equal(g1, g2), float((g1 - g2).abs().max())
