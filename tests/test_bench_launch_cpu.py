"""`python bench.py --gpus N` from a bare shell must start its N ranks itself (the driver invokes it that way as well as
under torch.distributed.run).  Exercised here on CPU: gloo, a stub step, the real launch / rendezvous / barrier /
MAX-over-ranks / flat-bucket all-reduce / one-JSON-line plumbing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines


@pytest.mark.timeout(300)
def test_bare_shell_gpus2_self_launches_two_ranks():
    p, lines = _run(["--gpus", "2", "--backend", "gloo", "--stub-step", "--steps", "3", "--warmup", "1", "--batch", "4"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout            # rank 0 prints ONE JSON line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["comm"] == {"backend": "gloo", "world_size": 2, "grad_bytes_per_step": 4 * (16 * 16 + 16 + 16 * 3 + 3)}
    assert line["ranks_in_sync"] is True        # the gradient exchange really averaged: identical weights on both ranks
    assert line["config"]["global_batch"] == 8 and line["config"]["parallelism"] == "dp2"
    assert abs(line["per_gpu"] * 2 - line["value"]) <= 0.2


@pytest.mark.timeout(300)
def test_single_rank_needs_no_launcher():
    p, lines = _run(["--gpus", "1", "--stub-step", "--steps", "2", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["comm"]["world_size"] == 1


@pytest.mark.timeout(120)
def test_world_size_mismatch_is_an_error_not_an_assert():
    p, _ = _run(["--gpus", "2", "--stub-step"], env_extra={"WORLD_SIZE": "4", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
