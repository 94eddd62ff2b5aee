#!/usr/bin/env python
"""Position-encoding MLP with and without its hidden tensor (hip_linear.pos_mlp against the two-layer path), per shape class of a
B = 32 step: forward (K = 4 kernel + gather GEMM against the one H0 GEMM), and the backward's weight-gradient work.

    python tools/bench_h0.py [reps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nsdp_amd import hip_linear  # noqa: E402
from nsdp_amd.model import ops  # noqa: E402

hip_linear.H0_RECOMPUTE = 2      # (every supported shape: this is the measurement the default's shape classes come from)
DEV = torch.device("cuda:0")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timed(fn, reps=REPS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    torch.manual_seed(0)
    print(f"{'rows':>9s} {'d':>4s} {'form':>5s} | fwd two-layer us  fwd h0 us | fwd+bwd two-layer us  fwd+bwd h0 us")
    for M, d, form in ((1_048_576, 200, 2), (1_843_200, 200, 2), (262_144, 200, 2), (327_680, 256, 1), (81_920, 256, 1), (51_200, 256, 1),
                       (262_144, 128, 1), (65_536, 128, 1)):
        seq = torch.nn.Sequential(torch.nn.Linear(3, d), torch.nn.ReLU(), torch.nn.Linear(d, d)).to(DEV)
        x = torch.nn.functional.pad(torch.randn(M, 3, device=DEV), (0, 1))
        t = torch.randn(M, d, device=DEV)
        k, nsrc = 16, 100
        shapes = 32
        rps = M // shapes
        gk = torch.randn(shapes * nsrc, d, device=DEV)
        gidx = torch.randint(0, nsrc, (M,), device=DEV, dtype=torch.int32)
        gather = (None, 1, gk, gidx, rps, nsrc) if form == 2 else (torch.randn(M // k, d, device=DEV), k, gk, gidx, rps, nsrc)

        def two():
            tl = ops.k4_tail(x, seq)
            h = ops.linear(x, seq[0], relu=True, tail_src=tl)
            return ops.linear(h, seq[2], init_gather=gather, tail_dst=tl)

        def new():
            return ops.pos_mlp(x, seq, init_gather=gather)

        def train(fn):
            def step():
                seq.zero_grad(set_to_none=False)
                fn().backward(t)
            return step

        with torch.no_grad():
            f_two, f_new = timed(two), timed(new)
        b_two, b_new = timed(train(two)), timed(train(new))
        print(f"{M:9d} {d:4d} {form:5d} | {f_two:16.1f} {f_new:10.1f} | {b_two:20.1f} {b_new:14.1f}", flush=True)


if __name__ == "__main__":
    main()
