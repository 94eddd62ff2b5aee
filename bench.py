#!/usr/bin/env python
"""bench.py -- query-points/s of the TDNet hot path (forward + loss + backward + Adam) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B]

One process per GPU (torch.distributed / RCCL for N > 1).  Started under torch.distributed.run (RANK / WORLD_SIZE in
the environment) it is one rank of the job; started from a bare shell with --gpus N > 1 it launches the N ranks itself
(re-exec through `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` on a free
port, rank r bound to GPU r).  A "step"
is one pass of the reference's ``train_on_batch`` sequence (model/deformation_networks.py:63-77) over
one synthetic batch of B shapes per GPU, 2048 surface + 8192 query points each, forward.yaml
architecture, fp32, inputs resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SURF, N_QUERY = 2048, 8192
FLOP_PER_QUERY_FWD_BWD = 14.3e6  # 117 GFLOP fwd+bwd per 8192-query shape


def model_config():
    from nsdp_amd.config import default_config
    cfg = default_config("forward")
    cfg["model"]["encoder_kwargs"]["npoints_per_layer"] = [N_SURF, 500, 100]
    return cfg


def cpu_baseline(seconds_budget=20.0):
    """Reference algorithm on the host cores: the oracle port (oracle/tdnet_ref.py, pinned against the
    imported reference) running the same train step at B=1 -- a bounded sample, reported beside the GPU
    number, never part of it."""
    import torch
    from nsdp_amd import synth
    from oracle import tdnet_ref
    from nsdp_amd.model import build_model
    cfg = model_config()
    model, *_ = build_model(cfg, device="cpu")
    state = synth.procedural_state_dict(model.state_dict(), 2048)
    del model
    sd = tdnet_ref.to_torch_state(state, requires_grad=True)
    names = tdnet_ref.trainable(sd)
    opt = torch.optim.Adam([{"params": [sd[k] for k in names], "lr": 5e-4}])
    data = {k: torch.from_numpy(v) for k, v in synth.make_batch(2048, 1, N_SURF, N_QUERY).items()}
    tdnet_ref.train_step(sd, cfg["model"], data, opt)  # warm-up
    # small per-op tensors do not scale to 256 hardware threads: pick the fastest of a few thread counts
    best = None
    from nsdp_amd.cpu_budget import cpu_budget
    for nt in (8, 16, 32, 64):
        if nt > cpu_budget():        # (affinity and cgroup quota, not os.cpu_count())
            break
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        tdnet_ref.train_step(sd, cfg["model"], data, opt)
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (nt, dt)
    torch.set_num_threads(best[0])
    t0 = time.perf_counter()
    n = 0
    while True:
        tdnet_ref.train_step(sd, cfg["model"], data, opt)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 12:
            break
    # `cores` (the contract's field) = the THREADS the port really used -- not cores of the host: the box has
    # `host_logical_cpus` hardware threads of which this container may use `cpu_quota` (affinity and cgroup cpu.max)
    return {"value": round(n * N_QUERY / el, 1), "unit": "query-points/s", "cores": torch.get_num_threads(),
            "threads": torch.get_num_threads(), "host_logical_cpus": os.cpu_count(), "cpu_quota": cpu_budget(),
            "kind": "port", "sample": f"{n} train steps at B=1 (2048 surf / 8192 query), fp32, oracle/tdnet_ref.py "
                                      f"on {torch.get_num_threads()} host threads (fastest of 8/16/32/64 within the "
                                      f"container's {cpu_budget()}-CPU budget; {os.cpu_count()} logical CPUs on the host)"}


def bf16_parity(workload, device):
    """The accuracy cost of `--dtype bf16`, next to the throughput it buys: eval-mode L2 of the bf16-storage product against
    the REFERENCE's fp32 output (tests/golden/full_forward.npz / full_arbitrary.npz: the imported reference on CPU at 2048
    surface + 8192 query points, same procedural weights and seeded inputs), outside the timed region.  None when the
    fixture is not in the tree."""
    import numpy as np
    import torch
    from nsdp_amd import precision, synth
    from nsdp_amd.model import build_model
    name, mtype = ("full_forward", "forward") if workload == "forward_train" else ("full_arbitrary", "arbitrary")
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    if not os.path.exists(path):
        return None
    fx = np.load(path, allow_pickle=False)
    seed, b, ns, nq = (int(fx[k]) for k in ("meta_seed", "meta_batch", "meta_ns", "meta_nq"))
    cfg = model_config()
    cfg["model"]["type"] = mtype
    model, *_ = build_model(cfg, device="cpu")
    state = synth.procedural_state_dict(model.state_dict(), seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.to(device).eval()
    d = {k: torch.from_numpy(v).to(device) for k, v in synth.make_batch(seed, b, ns, nq).items()}
    out = {}

    def forward(mode, scale=1.0, round_inputs=False):
        with precision.storage(mode), torch.no_grad():
            s_in, q = d["surface_samples_inputs"] * scale, d["space_samples_src"] * scale
            if round_inputs:
                s_in, q = s_in.to(torch.bfloat16).float(), q.to(torch.bfloat16).float()
            mask = d["surface_samples_inputs"][:, :, 6:7]
            if mtype == "arbitrary":
                o = model(q, s_in[:, :, 0:3].contiguous(), s_in[:, :, 3:6].contiguous(), mask.contiguous())
            else:
                o = model(q, torch.cat([s_in[:, :, 0:6], mask], dim=-1).contiguous())
        return o.float().cpu().numpy().astype(np.float64) / scale
    for mode in ("f32", "bf16"):
        out[mode] = forward(mode)
    # chaos floor: what a bf16-SIZED perturbation of the inputs alone does to the fp32 model's output (neighbour sets flip,
    # FPS picks other centres) -- the yardstick for the bf16 figure (tools/bf16_bisect.py, profiles/r4_bf16_bisect.txt)
    floor_scaled = forward("f32", 1.0 + 2.0 ** -8)
    floor_rounded = forward("f32", 1.0, True)
    stride = int(fx["meta_eval_stride"]) if "meta_eval_stride" in fx else 1
    ref = fx["eval_out"].astype(np.float64)
    # (the two worst queries per shape are left out: where a query's k-th and (k+1)-th anchor distances are bit-equal the
    # reference's unstable argsort picks an arbitrary neighbour set -- 1 of 8192 queries in full_forward, see
    # tests/test_model_gpu.py::test_full_shape_forward_matches_golden)
    def l2(o):
        err = ((o[:, ::stride] - ref) ** 2).sum(-1)
        return float(np.sqrt(np.sort(err, axis=1)[:, :-2].mean(-1)).max())
    def l2_pair(a, b_):
        err = ((a[:, ::stride] - b_[:, ::stride]) ** 2).sum(-1)
        return float(np.sqrt(np.sort(err, axis=1)[:, :-2].mean(-1)).max())
    return {"bf16": round(l2(out["bf16"]), 6), "f32": round(l2(out["f32"]), 8), "fixture": f"tests/golden/{name}.npz",
            "canonicalize_f32": precision.canonicalize_f32() if mtype == "arbitrary" else None,
            "canonicalize_mode": precision.canonicalize_mode() if mtype == "arbitrary" else None,
            "chaos_floor_fp32_inputs_x_1p2m8": round(l2_pair(floor_scaled, out["f32"]), 6),
            "chaos_floor_fp32_inputs_rounded_to_bf16": round(l2_pair(floor_rounded, out["f32"]), 6),
            "metric": "max over shapes of sqrt(mean_q |pred - reference|^2) without the 2 worst queries, eval forward, B=%d" % b}


def stub_main(args, rank, world):
    """The multi-rank plumbing of this file on a box without GPUs: same rendezvous, barrier-bracketed timing, MAX over
    ranks, flat-bucket all-reduce (nsdp_amd.parallel.GradAllReducer) and JSON line, around a tiny CPU model.  Used by
    tests/test_bench_launch_cpu.py; the line is marked as a stub and is not a measurement."""
    import torch
    import torch.distributed as dist
    from nsdp_amd.parallel import GradAllReducer
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if args.backend == "nccl" and not torch.cuda.is_available() else args.backend,
                                rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.ModuleDict({"encoder": torch.nn.Linear(16, 16), "decoder": torch.nn.Linear(16, 3)})
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    reducer = GradAllReducer(model, world) if world > 1 else None
    g = torch.Generator().manual_seed(1000 + rank)
    x = torch.randn(args.batch * 64, 16, generator=g)
    y = torch.randn(args.batch * 64, 3, generator=g)

    def step():
        reducer.zero_grad() if reducer is not None else opt.zero_grad()
        loss = ((model["decoder"](torch.relu(model["encoder"](x))) - y) ** 2).mean()
        loss.backward()
        if reducer is not None:
            reducer.all_reduce_mean()
        opt.step()
        return loss

    def fence():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank must hold the same weights after the same number of averaged steps
        w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        lo, hi = w.clone(), w.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool(torch.equal(lo, hi))
    else:
        in_sync = True
    if rank == 0:
        total = world * args.batch * 64 * args.steps
        print(json.dumps({"metric": "stub rows/s (plumbing test, NOT a measurement)", "value": round(total / elapsed, 1),
                          "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "stub", "global_batch": world * args.batch, "parallelism": f"dp{world}"},
                          "per_gpu": round(total / elapsed / world, 1),
                          "comm": {"backend": dist.get_backend() if world > 1 else None,
                                   "world_size": dist.get_world_size() if world > 1 else 1,
                                   "grad_bytes_per_step": reducer.nbytes if reducer is not None else None},
                          "ranks_in_sync": in_sync, "final_loss": round(float(loss), 6), "stub": True}))
    if world > 1:
        dist.destroy_process_group()


def self_launch(n, argv):
    """`python bench.py --gpus N` from a bare shell (no RANK / WORLD_SIZE): start the N ranks ourselves through
    torch.distributed.run on 127.0.0.1 and a free port; the children take the worker path below.  Returns the exit
    code of the launcher (rank 0's JSON line passes through on stdout)."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, 16 // n)))      # (the ranks share the box's CPU budget: nsdp_amd/cpu_budget.py)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


DEFAULT_BATCH = {"forward_train": 32, "arbitrary_train": 32, "forward_eval": 8, "dense_inference": 4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None,
                    help="shapes per GPU (weak scaling); default 32 for the train steps (BASELINE configs 3/4 per GPU), "
                         "8 for forward_eval (config 2), 4 for dense_inference (config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reps", type=int, default=3,
                    help="timed repetitions of the K-step region (each bracketed by barrier + synchronize); the line reports "
                         "the MEDIAN repetition as value / ms_per_step and min / max beside it")
    ap.add_argument("--workload", default="forward_train",
                    choices=["forward_train", "arbitrary_train", "forward_eval", "dense_inference"],
                    help="forward_train (default, the headline metric) | arbitrary_train (BASELINE config 3) | "
                         "forward_eval (BASELINE config 2: eval forward, 8192 queries per shape) | dense_inference "
                         "(BASELINE config 5: eval, 100k queries per shape)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only "
                    "to exercise the multi-rank code path on a single GPU)")
    ap.add_argument("--force-reducer", action="store_true",
                    help="use the flat-bucket gradient path even at world size 1 (exercises the DP code on one GPU)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="storage precision of the activations: f32 (default; dense layers as error-compensated bf16x3 "
                         "products, fp32 accuracy) | bf16 (BASELINE config 3: bf16 activations and saved tensors, fp32 "
                         "accumulation, fp32 master weights)")
    ap.add_argument("--dp-overlap", choices=["auto", "on", "off"], default=os.environ.get("NSDP_DP_OVERLAP", "auto"),
                    help="data-parallel steps: 'on' = the backward pass in two autograd passes cut at the decoder's inputs, bucket 0's "
                         "all-reduce issued between them (a captured step is then THREE graphs: head / tail / update); 'off' = one "
                         "backward pass, the whole exchange behind it (two graphs); auto (default): on for eager steps, off for "
                         "captured ones -- every graph boundary joins the executor's streams, and the head / tail boundary costs the "
                         "overlap of the decoder's weight gradients with the encoder's backward chain: measured at ONE rank, B = 32: "
                         "plain 38.2-38.3 ms, two graphs 38.75-38.9, three graphs 39.5-39.7, against an exchange of 0.05-0.2 ms")
    ap.add_argument("--canonicalize-decoder-f32", action="store_true",
                    help="with --dtype bf16 --workload arbitrary_train: the middle point -- FlowArbitrary's first network with a bf16 "
                         "ENCODER and an fp32-storage DECODER (its per-point outputs are the coordinates the second network searches)")
    ap.add_argument("--canonicalize-f32", action="store_true",
                    help="with --dtype bf16 --workload arbitrary_train: FlowArbitrary's first network in fp32 storage (its output "
                         "points are the second network's geometry: eval L2 against the reference 1.0e-2 instead of 1.3e-1)")
    ap.add_argument("--eager", action="store_true",
                    help="enqueue every launch of every step from Python (the reference's way).  Default: the step "
                         "(train or inference) is captured once and replayed through the multi-stream graph executor "
                         "(nsdp_amd/graph_step.py: one C call per step instead of ~1000 Python-enqueued launches; same "
                         "kernels, same stream schedule, same numbers step for step)")
    ap.add_argument("--graph", action="store_true", help="(the default; kept for symmetry with --eager)")
    ap.add_argument("--geometry", default="inline", choices=["pipelined", "inline"],
                    help="inline (default): the step searches inside its forward pass, like the reference.  pipelined (replayed "
                         "TDNet steps): every step computes the NEXT batch's index sets (FPS, kNN, inverse lists: functions of "
                         "the batch, not of the weights) on a stream of its own beside its forward pass and takes its own from "
                         "the previous step (nsdp_amd.graph_step.PipelinedGeometry) -- the same searches once per step, results "
                         "bit-identical.  Measured: eval B = 8 4.42 -> 4.24 ms, train B = 8 13.62 -> 13.37, B = 32 39.12 -> 38.95 "
                         "(the searches still cost their chip time; only the sampling chain's latency leaves the critical path)")
    ap.add_argument("--stub-step", action="store_true",
                    help="replace the TDNet step by a tiny CPU model (tests of the launch / rendezvous / all-reduce / "
                         "timing / JSON plumbing on a box without GPUs; the line says so and is not a measurement)")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = DEFAULT_BATCH[args.workload]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    # CPU thread pools within the cgroup quota (nsdp_amd/cpu_budget.py: 256 spinning OpenMP workers under a 16-CPU quota
    # throttle the thread that enqueues the GPU work -- steps of 60-90 ms instead of 45)
    from nsdp_amd.cpu_budget import cap_thread_pools, cpu_budget
    cap_thread_pools(16 // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))) or 1)
    import torch
    cap_thread_pools(16 // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))) or 1)
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}")
    if args.stub_step:
        return stub_main(args, rank, world)
    if (args.backend == "nccl" and torch.cuda.device_count() < world // max(1, int(os.environ.get("NNODES", "1")))
            and not (torch.cuda.device_count() == 1 and os.environ.get("HIP_VISIBLE_DEVICES", "").count(",") == 0
                     and os.environ.get("HIP_VISIBLE_DEVICES", "") != "")):
        sys.exit(f"bench.py: --gpus {args.gpus} needs {world} visible GPUs, found {torch.cuda.device_count()}")
    # LOCAL_RANK -> GPU: with a per-rank HIP_VISIBLE_DEVICES (some launchers narrow it to ONE device per process) the only
    # visible device is 0; otherwise rank r takes device r
    one_visible = torch.cuda.device_count() == 1 and world > 1 and args.backend == "nccl" and \
        len([d for d in os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")).split(",") if d]) == 1
    dev_index = 0 if one_visible else (local_rank % torch.cuda.device_count() if args.backend != "nccl" else local_rank)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cpu_mask = None
    if local_world > 1:
        from nsdp_amd.cpu_budget import pin_rank
        props = torch.cuda.get_device_properties(dev_index)
        bdf = (f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
               if hasattr(props, "pci_bus_id") else None)
        cpu_mask = pin_rank(local_rank, local_world, bdf)
        print(f"bench.py: rank {rank} -> GPU {dev_index} ({bdf}), CPUs {cpu_mask}", file=sys.stderr)
    exchange_world1 = world == 1 and args.force_reducer and args.backend == "nccl"
    if world > 1 or exchange_world1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if exchange_world1:      # (a one-rank RCCL communicator on this GPU: always a fresh port of its own -- an inherited
            #                       MASTER_PORT belongs to whoever exported it)
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from nsdp_amd import precision, profiling, synth
    from nsdp_amd.model import build_model, optimizer_factory
    precision.set_storage(args.dtype)
    if args.canonicalize_f32:
        precision.set_canonicalize_f32(True)
    if args.canonicalize_decoder_f32:
        precision.set_canonicalize_mode("dec32")
    from nsdp_amd.parallel import DataParallel
    from nsdp_amd.model.utils import compute_l2_error

    cfg = model_config()
    n_query = N_QUERY
    if args.workload == "arbitrary_train":
        cfg["model"]["type"] = "arbitrary"
    elif args.workload == "dense_inference":
        n_query = 100000
    is_eval = args.workload in ("forward_eval", "dense_inference")
    model, _train_on_batch, _, _ = build_model(cfg, device="cpu")
    state = synth.procedural_state_dict(model.state_dict(), 2048)  # identical weights on every rank
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model.to(device).train(not is_eval)
    _, optimizer = optimizer_factory({"optimizer": "Adam", "lr": 5e-4, "lr_step": 200, "lr_decay": 0.1,
                                      "weight_decay": 0.0}, model.parameters())
    dp = DataParallel(model, rank, world, always_exchange=exchange_world1) if (world > 1 or args.force_reducer) else None
    reducer = dp.reducer if dp is not None else None
    if dp is not None:
        dp.broadcast_model(model)      # rank 0's weights and buffers everywhere (they are procedural, i.e. equal already: this is
        #                                the job's start-up protocol, exercised; `ranks_in_sync` below is its check)
    comm_events = []                   # (start, end) HIP event pairs around each gradient exchange of the timed region

    def exchange():
        # the EXPOSED part of the gradient exchange: bucket 0 (the decoder's gradients) has been travelling since the decoder's
        # backward pass ended (reducer.backward / the head graph); here bucket 1 is issued and the stream waits for both
        if len(comm_events) < 4096:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reducer.finish()
            e1.record()
            comm_events.append((e0, e1))
        else:
            reducer.finish()
    data = {k: torch.from_numpy(v).to(device)
            for k, v in synth.make_batch(1000 + rank, args.batch, N_SURF, n_query).items()}

    # The next batch's geometry beside the current step (see --geometry).  The synthetic "next batch" is a second static copy of
    # the same tensors: every step runs the whole search for it, nothing is carried over but the index sets the previous step
    # computed for THIS one.
    pipe = data_next = None
    PIPE_STREAMS = int(os.environ["NSDP_PIPE_STREAMS"]) if "NSDP_PIPE_STREAMS" in os.environ else None      # (A/B: executor streams of a pipelined step)
    if (args.geometry == "pipelined" and not args.eager and args.workload != "arbitrary_train" and hasattr(model, "geometry")
            and not args.stub_step):
        from nsdp_amd.graph_step import PipelinedGeometry
        data_next = {k: v.clone() for k, v in data.items()}
        pipe = PipelinedGeometry(model, lambda d: (d["space_samples_src"], d["surface_samples_inputs"]))
        pipe.prime(data, training=not is_eval)

    def forward():
        if args.workload == "arbitrary_train":
            s_in = data["surface_samples_inputs"]
            return model(data["space_samples_src"], s_in[:, :, 0:3], s_in[:, :, 3:6], s_in[:, :, 6:7])
        if pipe is not None:
            return model(data["space_samples_src"], data["surface_samples_inputs"], geometry=pipe.current)
        return model(data["space_samples_src"], data["surface_samples_inputs"])

    def infer_step():
        if pipe is not None:
            pipe.prefetch(data_next)
        with torch.no_grad():
            out = forward().sum()
        if pipe is not None:
            pipe.rotate()
        return out

    def step():
        # train_on_batch_with_cano (reference model/deformation_networks.py:63-77); the loss scalar is
        # read back after the timed region instead of per step (loss.item() is a pure host sync).
        if pipe is not None:
            pipe.prefetch(data_next)
        if reducer is not None:
            reducer.zero_grad(two_pass=overlap_eager)
        else:
            optimizer.zero_grad(set_to_none=True)
        pred = forward()
        loss = compute_l2_error(pred, data["space_samples_tgt"])
        if reducer is not None:
            reducer.backward(loss)      # two autograd passes, bucket 0's all-reduce issued between them (nsdp_amd/parallel.py)
            exchange()
        else:
            loss.backward()
        optimizer.step()
        if pipe is not None:
            pipe.rotate()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    overlap_eager = args.dp_overlap != "off"          # reducer.backward falls back to one pass when the forward was not cut
    overlap_graph = args.dp_overlap == "on"

    run = infer_step if is_eval else step
    graph = None
    timed_step = None
    graph_note = "eager (Python enqueues every launch)"
    eager_run = run
    if not args.eager:
        # the step captured once, replayed from C on real HIP streams (plain hipGraphLaunch serialises the branches of the
        # captured graph on this ROCm and is slower than eager).  A collective cannot be captured:
        # with a gradient exchange the step is one graph per side of each collective -- [zero_grad, forward, loss, the
        # decoder's backward] | all-reduce(bucket 0), asynchronous | [the encoder's backward] | all-reduce(bucket 1), wait |
        # [optimizer.step] -- and the all-reduces stay eager RCCL calls.
        from nsdp_amd.graph_step import GraphedStep, capturable_adam
        try:
            if not is_eval:
                capturable_adam(optimizer)
            if is_eval or reducer is None:
                graph = GraphedStep(run, weights_change=not is_eval, max_streams=PIPE_STREAMS if pipe is not None else None).capture(warmup=3)      # (inference: frozen weights)
                run = graph
                graph_note = "graph replay, multi-stream executor: " + json.dumps(graph.info)
            else:
                for _ in range(3):
                    step()

                def head():
                    # (the search of the next batch's index sets forks a stream that its hand-over joins: both in ONE graph)
                    if pipe is not None and not (overlap_graph and reducer._can_cut):
                        pipe.prefetch(data_next)
                    reducer.zero_grad(two_pass=overlap_graph)
                    loss = compute_l2_error(forward(), data["space_samples_tgt"])
                    if not reducer.backward_head(loss):      # (no cut: the whole backward in this graph, no tail graph)
                        loss.backward()
                        if pipe is not None:
                            pipe.rotate()
                    return loss

                def tail():
                    if pipe is not None:
                        pipe.prefetch(data_next)
                    reducer.backward_tail()
                    if pipe is not None:      # (the hand-over overwrites the index sets the backward pass reads: backward first)
                        pipe.rotate()
                    return reducer.flat
                g1 = GraphedStep(head, max_streams=PIPE_STREAMS if pipe is not None else None).capture(warmup=0)
                g_tail = None
                if reducer._root_grads is not None:      # the forward was cut: the encoder's backward is a graph of its own
                    reducer.start(0)
                    # (the tail rebuilds no weight pack -- it runs on what the head's forward saved: frozen-weights capture)
                    g_tail = GraphedStep(tail, max_streams=PIPE_STREAMS if pipe is not None else None, weights_change=False).capture(warmup=0)
                reducer.finish()
                g2 = GraphedStep(lambda: optimizer.step()).capture(warmup=0)
                graph = g1

                def run():
                    loss = g1()
                    if g_tail is not None:
                        reducer.start(0)
                        g_tail()
                    exchange()
                    g2()
                    return loss

                def timed_step(stem):      # (one whole step with event pairs around the class's launches in the head and the tail)
                    n_a, ms_a = g1.timed_replay(stem)
                    n_b, ms_b = 0, 0.0
                    if g_tail is not None:
                        reducer.start(0)
                        n_b, ms_b = g_tail.timed_replay(stem)
                    reducer.finish()
                    g2()
                    return n_a + n_b, ms_a + ms_b
                graph_note = (("graph replay, multi-stream executor, one graph per side of each all-reduce (head / tail / update): "
                               if g_tail is not None else "graph replay, multi-stream executor, two graphs around the eager all-reduce: ")
                              + " + ".join(json.dumps(g.info) for g in (g1, g_tail, g2) if g is not None))
        except Exception as exc:      # (a PyTorch / ROCm without the capture hooks: the eager step is always there)
            if os.environ.get("NSDP_BENCH_REQUIRE_GRAPH") == "1":      # (the tests: a silent fallback would hide a broken capture)
                raise
            graph, run, timed_step = None, eager_run, None
            graph_note = f"eager (graph capture unavailable: {type(exc).__name__}: {str(exc)[:120]})"
            print(f"bench.py: WARNING: the step could not be captured, timing the EAGER step instead: {type(exc).__name__}: {exc}",
                  file=sys.stderr, flush=True)
    # set-up, not part of the contract's W: the first steps of a fresh process also build the weight packs, grow the
    # caching allocator to its steady state and bring the GPU out of its idle power state (a cold first run was
    # measured 25 % slow with W = 3)
    for _ in range(3):
        run()
    fence()
    for _ in range(args.warmup):
        run()
    fence()
    # Per-kernel durations come from HIP events around each launch (csrc/api.hip).  An event pair keeps consecutive
    # kernels from overlapping head-to-tail, and timing all ~500 launches of a step was measured to cost 3 ms of a
    # 52 ms step -- so the full per-kernel table is taken in an extra, untimed pass, and inside the timed region only
    # the dominant kernel class (the one the roofline object is about) is timed.
    # That extra pass also switches the weight-gradient side stream off: with it, an event pair on one stream also
    # spans the time the kernel waits for CUs held by the other stream.
    prof_iso = None
    dominant = "linear_bf16x3_kernel" if args.dtype == "f32" else "linear_bf16_kernel"
    if not is_eval or graph is not None:      # (every rank runs it -- same program on every rank; rank 0 reports it)
        # (with graph replay this pass runs the EAGER step function: the kernels are the same, and a replay carries no events)
        from nsdp_amd import hip_linear
        one = eager_run if graph is not None else run
        was = hip_linear._OVERLAP_WGRAD
        hip_linear._OVERLAP_WGRAD = False
        one()
        torch.cuda.synchronize()
        profiling.start()
        one()
        one()
        torch.cuda.synchronize()
        prof_iso = profiling.stop()
        hip_linear._OVERLAP_WGRAD = was
        if prof_iso:
            dominant = max(prof_iso.items(), key=lambda kv: kv[1]["ms"])[0]
        run()            # back on the overlapped schedule before the clock starts
    elif is_eval:
        dominant = None      # few launches per step: time them all
    fence()
    # Python's cyclic collector is kept out of the timed region (a generation-2 pass over the objects of a model this
    # size was seen to stall the host for ~80 ms -- a whole inference step -- whenever it happened to fall inside);
    # nothing the step allocates is cyclic garbage, and the collector is switched back on right after
    import gc
    gc.collect()
    gc.disable()
    # host pace with an EMPTY launch queue: one step enqueued right after a synchronize -- what the host needs for a step
    # when nothing pushes back (inside the timed region the enqueue calls also wait for queue / kernarg slots whenever the
    # GPU is the slower side, so `host_enqueue_ms_per_step` there converges to the step time itself)
    t0 = time.perf_counter()
    run()
    host_unblocked = time.perf_counter() - t0
    fence()
    if os.environ.get("NSDP_BENCH_NO_EVENTS") != "1":    # (A/B knob: cost of the HIP events themselves)
        profiling.start(only=[dominant] if dominant else None)
    comm_events.clear()
    reps = []
    for _ in range(max(1, args.reps)):      # each repetition: EXACTLY K steps between fences, MAX over ranks
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = run()
        t_enq = time.perf_counter() - t0       # host time to ENQUEUE the K steps (no sync inside a step)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        reps.append((el, t_enq))
    prof = profiling.stop()
    order = sorted(range(len(reps)), key=lambda i: reps[i][0])
    elapsed, t_enqueued = reps[order[len(order) // 2]]        # the median repetition is the reported one
    final_loss = float(loss.item())
    comm_ms = None
    if comm_events:
        comm_ms = sum(a.elapsed_time(b) for a, b in comm_events) / len(comm_events)
    # The reference-shaped step beside the headline: train_on_batch returns the loss as a FLOAT every step (reference
    # model/deformation_networks.py:77) -- one host sync per step, behind which the next step's enqueue (3-5 ms of host time
    # for a replay, 20+ ms eager) is exposed.  Same K steps, same launcher, outside the contract's timed region.
    with_readback = None
    if not is_eval:
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            float(run().item())
        fence()
        with_readback = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([with_readback], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            with_readback = float(t.item())
    # The dominant kernel class AS REPLAYED: one more replay (outside every timed region) with a HIP event pair around each
    # of its launches inside the executor -- durations then include the time a kernel shares the chip with the other
    # stream's kernels, which the isolated pass above excludes on purpose.
    replayed = None
    if graph is not None and dominant and hasattr(graph, "timed_replay"):
        try:
            fence()
            stem = dominant.replace("_kernels", "").replace("_kernel", "") + "_kernel"
            timed = timed_step if timed_step is not None else graph.timed_replay
            n1, ms1 = timed(stem)
            n2, ms2 = timed(stem)
            if n1 and n1 == n2:
                replayed = {"launches_per_step": n1, "ms_per_step": round(0.5 * (ms1 + ms2), 3)}
        except Exception as exc:      # (measurement only)
            replayed = {"error": f"{type(exc).__name__}: {str(exc)[:100]}"}
        fence()
    gc.enable()
    in_sync = dp.in_sync(model) if dp is not None else True
    parity = None
    if rank == 0 and args.dtype == "bf16" and args.workload in ("forward_train", "arbitrary_train"):
        parity = bf16_parity(args.workload, device)

    if rank == 0:
        total_q = world * args.batch * n_query * args.steps
        value = total_q / elapsed
        names = {"forward_train": ("query-points/sec fwd+bwd (2048 surf pts, 8192 queries)",
                                   "forward.yaml TDNet train step (fwd + l2 loss + bwd + Adam)"),
                 "arbitrary_train": ("query-points/sec fwd+bwd, FlowArbitrary (2048 surf pts, 8192 queries)",
                                     "arbitrary.yaml FlowArbitrary train step (two TDNets, fwd + l2 loss + bwd + Adam)"),
                 "forward_eval": ("query-points/sec forward (2048 surf pts, 8192 queries)",
                                  "forward.yaml TDNet eval forward (no grad)"),
                 "dense_inference": ("query-points/sec forward (2048 surf pts, 100000 queries)",
                                     "forward.yaml TDNet eval forward, dense per-vertex decode")}[args.workload]
        line = {
            "metric": names[0],
            "value": round(value, 1), "unit": "query-points/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "ms_per_step_reps": {"n": len(reps), "median": round(1e3 * elapsed / args.steps, 3),
                                 "min": round(1e3 * reps[order[0]][0] / args.steps, 3),
                                 "max": round(1e3 * reps[order[-1]][0] / args.steps, 3)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{names[1]}, "
                                   f"{args.batch} shapes/GPU, {N_SURF} surface + {n_query} query points per shape, "
                                   + ("fp32 (dense layers as bf16x3 split products)" if args.dtype == "f32" else
                                      "bf16 storage / fp32 accumulate / fp32 master weights"
                                      + (" (network 1 in fp32 storage)" if args.canonicalize_f32 else "")
                                      + (" (network 1: bf16 encoder, fp32-storage decoder)" if args.canonicalize_decoder_f32 else ""))
                                   + ", procedural random-init weights",
                       "global_batch": world * args.batch, "n_surf": N_SURF, "n_query": n_query,
                       "parallelism": f"dp{world}"},
            "per_gpu": round(value / world, 1),
            "ms_per_step_with_loss_readback": (round(1e3 * with_readback / args.steps, 3) if with_readback is not None else None),
            "host_enqueue_ms_per_step": round(1e3 * t_enqueued / args.steps, 3),
            "host_enqueue_unblocked_ms": round(1e3 * host_unblocked, 3),
            "step_launch": graph_note,
            "geometry": ("pipelined: every step computes the next batch's index sets (FPS, kNN, inverse lists) beside its forward "
                         "pass and uses the ones the previous step computed for it" if pipe is not None else
                         "inline: searched inside the step's forward pass"),
            "cpu_mask": cpu_mask,
            "parity_l2_vs_fp32": parity,
            "comm": {"backend": (dist.get_backend() if dist.is_initialized() else None),
                     "world_size": (dist.get_world_size() if dist.is_initialized() else 1),
                     "grad_bytes_per_step": (reducer.nbytes if reducer is not None else None),
                     "exchange_ms_per_step": (round(comm_ms, 4) if comm_ms is not None else None),
                     "exchange": ("flat fp32 gradient, 2 in-place all-reduce buckets (mean: ncclAvg / sum + scale): bucket 0 -- the "
                                  "decoder's gradients -- issued when the decoder's backward pass has ended and travelling under the "
                                  "encoder's; exchange_ms_per_step = the exposed part (bucket 1 + the wait for both)"
                                  if reducer is not None else None),
                     "bucket0_enqueued_between_the_backward_passes": (bool(reducer.split) if reducer is not None else None)},
            "ranks_in_sync": in_sync,
            "model_tflops": round(value * FLOP_PER_QUERY_FWD_BWD / 1e12, 2) if args.workload == "forward_train" else None,
            "final_loss": round(final_loss, 6),
            "roofline": profiling.roofline(prof, prof_iso,
                                           pmc_matches=(args.workload == "forward_train" and args.batch == 32),
                                           pmc_suffix=("" if args.dtype == "f32" else "_bf16"), replayed=replayed),
            "kernels": profiling.summary(prof_iso if prof_iso else prof),
            "kernels_from": ("2 untimed steps, every launch timed, weight gradients on the main stream" if prof_iso
                             else "timed region"),
        }
        line["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1 or args.workload != "forward_train") \
            else cpu_baseline()
        # RCCL writes its version banner through C stdio, which a pipe buffers until exit: flush it first, so that the
        # JSON line is the LAST line of stdout for whoever parses it
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    if not in_sync:
        sys.exit("bench.py: the ranks do not hold identical weights after the run")


if __name__ == "__main__":
    main()
