"""bench.py end to end on the GPU at a small batch: the one JSON line and its contract fields."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", *flags],
                         capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(os.environ, NSDP_BENCH_REQUIRE_GRAPH="1"))      # (a capture that fails must fail the test, not fall back)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("flags", [(), ("--dtype", "bf16"), ("--workload", "forward_eval")])
def test_bench_line_contract(flags):
    d = _run("--no-cpu-baseline", *flags)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "query-points/s" and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == ("bf16" if "bf16" in flags else "f32")
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = units processed / time: 2 shapes x 8192 query points per step
    assert abs(d["value"] - 2 * 8192 / (d["ms_per_step"] * 1e-3)) <= 1e-3 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_bench_cpu_baseline_leg():
    d = _run()
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "query-points/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_bench_reports_repetitions_and_the_unblocked_host_pace():
    d = _run("--no-cpu-baseline", "--reps", "3")
    r = d["ms_per_step_reps"]
    assert r["n"] == 3 and r["min"] <= r["median"] <= r["max"] and r["median"] == d["ms_per_step"]
    assert d["host_enqueue_unblocked_ms"] > 0


def test_bench_force_reducer_runs_the_exchange_over_rccl_on_one_gpu():
    """`--force-reducer` with the default backend: a one-rank RCCL communicator, the flat-bucket all-reduce on the device."""
    d = _run("--no-cpu-baseline", "--force-reducer", "--reps", "1")
    assert d["comm"]["backend"] == "nccl" and d["comm"]["world_size"] == 1 and d["comm"]["grad_bytes_per_step"] > 1e7
    assert d["final_loss"] > 0


def test_bench_bf16_line_carries_its_accuracy_cost_against_the_reference():
    d = _run("--no-cpu-baseline", "--dtype", "bf16", "--reps", "1")
    par = d["parity_l2_vs_fp32"]
    assert par is not None and par["fixture"].endswith("full_forward.npz")
    assert par["f32"] <= 1e-4 and 1e-4 < par["bf16"] < 5e-2, par


def test_bench_graph_replay_is_the_default_and_eager_is_still_there():
    d = _run("--no-cpu-baseline", "--reps", "1")
    assert d["step_launch"].startswith("graph replay") and d["final_loss"] > 0 and d["roofline"] is not None
    e = _run("--no-cpu-baseline", "--reps", "1", "--eager")
    assert e["step_launch"].startswith("eager") and e["roofline"]["in_step"]["launches"] > 0
    # (that the two ways of launching the step train alike is tests/test_graph_exec_gpu.py's subject: the runs here differ
    # in their number of set-up steps)
    f = _run("--no-cpu-baseline", "--reps", "1", "--force-reducer", "--backend", "gloo")
    forced_on = os.environ.get("NSDP_DP_OVERLAP") == "on"      # (knob matrix: the three-graph form everywhere)
    assert ("head / tail / update" if forced_on else "two graphs around the eager all-reduce") in f["step_launch"], f["step_launch"]
    # the decoder bucket's all-reduce under the encoder's backward: one graph per side of each collective
    f3 = _run("--no-cpu-baseline", "--reps", "1", "--force-reducer", "--backend", "gloo", "--dp-overlap", "on")
    assert "head / tail / update" in f3["step_launch"], f3["step_launch"]
    from helpers import nondeterministic_knobs
    tol = 5e-2 if nondeterministic_knobs() else 1e-6      # (with fp32 atomics in the step two runs differ in rounding, amplified by Adam)
    assert f3["ranks_in_sync"] and abs(f3["final_loss"] - f["final_loss"]) <= tol * max(1.0, abs(f["final_loss"])), (f3["final_loss"], f["final_loss"])
