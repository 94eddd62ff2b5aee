"""Training harness around the MI355X TDNet path with the loop semantics of the reference's train.py (SURVEY.md
section 8 next-1): seeding, epoch-wise step learning rate, checkpoint cadence, validation cadence and best-model
tracking are the reference's; the model, its train/validate functions and the optimizer come from
``nsdp_amd.model``.  The data loader is whatever yields the reference's ``data_dict`` batches; ``--synthetic`` uses
procedural batches (the dataset readers are outside this round's scope).

    python -m nsdp_amd.train config.yaml experiment_dir [--synthetic 4] [--epochs 2]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

from .checkpoints import load_best_checkpoints, load_checkpoints, save_best_checkpoints, save_checkpoints
from .model import build_model, optimizer_factory
from .model.learningrate import adjust_learning_rate


def seed_everything(seed: int):
    """train.py:68-72: numpy seeds torch, which seeds every GPU."""
    np.random.seed(seed)
    torch.manual_seed(np.random.randint(np.iinfo(np.int32).max))
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(np.random.randint(np.iinfo(np.int32).max))


class _Average:
    """Running mean of the batch losses of an epoch (utils/logger.py AverageAggregator)."""

    def __init__(self):
        self.total, self.count = 0.0, 0

    def add(self, v):
        self.total += float(v)
        self.count += 1

    @property
    def value(self):
        return self.total / max(self.count, 1)


def fit(model, fns, lr_scheduler, optimizer, train_loader, val_loader, config, experiment_directory, args, device,
        log=print, dp=None):
    """train.py:150-225.  ``fns`` = (train_on_batch, validate_on_batch) from build_model; ``args`` carries
    ``continue_from_epoch`` and ``best_val_loss`` (both updated by the checkpoint loaders and by this loop).
    ``dp``: a nsdp_amd.parallel.DataParallel -- this process is one rank of a data-parallel job (the reference is
    single-process, train.py:74-75): batches are sharded over the ranks, ``train_on_batch`` gets the gradient exchange in
    front of its optimizer step (unless it already carries one: GraphedTrainOnBatch(reducer=...)), epoch / validation
    losses are means over ranks, rank 0 writes the files.  The policy (BatchNorm buffers included) is DataParallel's
    docstring."""
    train_on_batch, validate_on_batch = fns
    load_best_checkpoints(model, experiment_directory, args, device)   # best first, then the latest (train.py:153-156)
    load_checkpoints(model, optimizer, experiment_directory, args, device)
    main = dp is None or dp.is_main
    if dp is not None:
        dp.broadcast_model(model)
        if not getattr(train_on_batch, "exchanges_gradients", False):
            train_on_batch = dp.wrap(train_on_batch)
    epochs = config["training"].get("epochs", 1000)
    save_every = config["training"].get("save_frequency", 20)
    val_every = config["validation"].get("frequency", 10)
    history = []
    for i in range(args.continue_from_epoch, epochs):
        adjust_learning_rate(lr_scheduler, optimizer, i)
        model.train()
        avg = _Average()
        if getattr(train_on_batch, "accepts_next_batch", False):
            # one batch of look-ahead: the step computes the NEXT batch's index sets beside itself (graph_step.PipelinedGeometry)
            nxt = None
            for sample in _with_next(train_loader if dp is None else dp.shard(train_loader), device):
                cur, nxt = sample
                avg.add(train_on_batch(model, optimizer, cur, config, next_data_dict=nxt))
        else:
            for b, sample in enumerate(train_loader if dp is None else dp.shard(train_loader)):
                sample = {k: v.to(device) for k, v in sample.items()}
                avg.add(train_on_batch(model, optimizer, sample, config))
        epoch_loss = avg.value if dp is None else dp.mean(avg.value, device)
        if main:
            log("epoch: {} - batches: {} - loss: {:.5f}".format(i + 1, avg.count, epoch_loss))
        history.append(("train", i, epoch_loss))
        want_val = i % val_every == 0 and i > 0
        if dp is not None and ((i % save_every) == 0 or want_val):
            dp.broadcast_buffers(model)          # rank 0's BatchNorm statistics are the model's
        if (i % save_every) == 0 and main:
            save_checkpoints(i, model, optimizer, experiment_directory)
        if want_val:
            model.eval()
            vavg = _Average()
            for b, sample in enumerate(val_loader if dp is None else dp.shard(val_loader)):
                sample = {k: v.to(device) for k, v in sample.items()}
                vavg.add(validate_on_batch(model, sample, config))
            val_loss = vavg.value if dp is None else dp.mean(vavg.value, device)
            if main:
                log("validation epoch: {} - loss: {:.5f}".format(i + 1, val_loss))
            history.append(("val", i, val_loss))
            if val_loss < args.best_val_loss:      # (the same number on every rank: same decision everywhere)
                if main:
                    save_best_checkpoints(i, model, experiment_directory, val_loss)
                args.best_val_loss = val_loss
    return history


def _with_next(loader, device):
    """(batch, next batch or None) pairs of ``loader``, both on ``device``."""
    it = iter(loader)
    try:
        cur = {k: v.to(device) for k, v in next(it).items()}
    except StopIteration:
        return
    for sample in it:
        nxt = {k: v.to(device) for k, v in sample.items()}
        yield cur, nxt
        cur = nxt
    yield cur, None


class SyntheticLoader:
    """``n_batches`` procedural data_dict batches per epoch (same generator as bench.py and the parity fixtures)."""

    def __init__(self, seed, n_batches, batch, n_surf=2048, n_query=8192):
        from . import synth
        self.batches = [{k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in
                         synth.make_batch(seed + i, batch, n_surf, n_query).items()} for i in range(n_batches)]

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)

    def shard(self, rank, world_size):
        """This rank's share of an epoch (DataParallel.shard asks for it): of every ``world_size`` consecutive batches the
        rank-th; a trailing partial group is dropped so that all ranks take the same number of steps."""
        groups = len(self.batches) // world_size
        return (self.batches[g * world_size + rank] for g in range(groups))


def initial_weight_files(config, args=None):
    """train.py:140-146: the three weight files that initialise the networks come from the config's ``training``
    section (``weight_file``, ``weight_forward_file``, ``weight_backward_file``); a command-line value, where given,
    overrides the config's.  Returns (weight_file, weight_forward_file, weight_backward_file)."""
    out = []
    for key in ("weight_file", "weight_forward_file", "weight_backward_file"):
        v = config.get("training", {}).get(key, None)
        cli = getattr(args, key, None) if args is not None else None
        out.append(cli if cli is not None else v)
    return tuple(out)


def launch_ranks(n, argv):
    """``python -m nsdp_amd.train ... --gpus N`` from a bare shell: start the N ranks through torch.distributed.run on
    127.0.0.1 and a free port (rank r -> GPU r); the children see RANK / WORLD_SIZE and take the worker path of main()."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, 16 // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "nsdp_amd.train"] + list(argv)
    return subprocess.call(cmd, env=env)


def main(argv=None):
    import yaml
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser(description="Train a deformation network on MI355X")
    ap.add_argument("config_file")
    ap.add_argument("experiment_directory")
    ap.add_argument("--weight_file", default=None, help="overrides training.weight_file of the config")
    ap.add_argument("--weight_forward_file", default=None, help="overrides training.weight_forward_file")
    ap.add_argument("--weight_backward_file", default=None, help="overrides training.weight_backward_file")
    ap.add_argument("--continue_from_epoch", default=0, type=int)
    ap.add_argument("--seed", type=int, default=27)
    ap.add_argument("--synthetic", type=int, default=4, help="procedural batches per epoch (no dataset readers yet)")
    ap.add_argument("--batch", type=int, default=8, help="shapes per batch = per GPU and step (the global batch is this x --gpus)")
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--graph", action="store_true",
                    help="capture the train step once and replay it (nsdp_amd.graph_step.GraphedTrainOnBatch): batches of "
                         "one fixed shape replay, others run eagerly; the sequence of optimizer steps is the eager loop's")
    ap.add_argument("--gpus", type=int, default=None,
                    help="data-parallel ranks, one process per GPU (nsdp_amd.parallel.DataParallel: batches sharded over the "
                         "ranks, one RCCL all-reduce of the flat gradient per step, rank-0 checkpoints).  From a bare shell "
                         "the ranks are launched here; under torchrun (RANK / WORLD_SIZE set) this process is one of them")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the data-parallel job (nccl = RCCL)")
    args = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus is not None and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, argv)
    if args.gpus is not None and args.gpus != world:
        sys.exit(f"nsdp_amd.train: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}")
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    args.best_val_loss = float("inf")
    seed_everything(args.seed)           # (the same seed on every rank: identical initial weights even before the broadcast)
    device = torch.device("cuda", 0)
    if world > 1:
        import torch.distributed as dist
        from .parallel import rank_device
        index, cpus = rank_device(local_rank, world, args.backend)       # rank r -> GPU r, pinned to its CPU slice
        device = torch.device("cuda", index)
        print(f"nsdp_amd.train: rank {rank} -> GPU {index}, CPUs {cpus}", file=sys.stderr)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    with open(args.config_file) as f:
        config = yaml.safe_load(f)
    if args.epochs is not None:
        config["training"]["epochs"] = args.epochs
    os.makedirs(args.experiment_directory, exist_ok=True)
    weights = initial_weight_files(config, args)
    say = print if rank == 0 else (lambda *a, **k: None)
    for key, path in zip(("weight_file", "weight_forward_file", "weight_backward_file"), weights):
        if path is not None:
            say("initialising from {} = {}".format(key, path))
    if config["model"]["type"] == "arbitrary" and weights[1] is None and weights[2] is None and weights[0] is None:
        say("WARNING: FlowArbitrary starts from random weights (no weight_forward_file / weight_backward_file in the "
            "config's training section or on the command line)")
    model, train_fn, val_fn, _ = build_model(config, *weights, device=device)
    lr_scheduler, optimizer = optimizer_factory(config["training"], model.parameters())
    dp = None
    if world > 1:
        from .parallel import DataParallel
        dp = DataParallel(model, rank, world)
    if args.graph:
        from .graph_step import GraphedTrainOnBatch
        train_fn = GraphedTrainOnBatch(train_fn, reducer=dp.reducer if dp is not None else None)
    # (one loader of world x synthetic batches: every rank builds the same list and takes its share, DataParallel.shard)
    train = SyntheticLoader(args.seed, args.synthetic * world, args.batch)
    val = SyntheticLoader(args.seed + 10000, max(1, args.synthetic // 4) * world, args.batch)
    fit(model, (train_fn, val_fn), lr_scheduler, optimizer, train, val, config, args.experiment_directory, args, device,
        log=say, dp=dp)
    if dp is not None:
        import torch.distributed as dist
        ok = dp.in_sync(model)
        say("ranks in sync after training: {}".format(ok))
        dist.destroy_process_group()
        if not ok:
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
