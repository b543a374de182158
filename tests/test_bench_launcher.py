"""`python bench.py --gpus N` must run by itself: with WORLD_SIZE unset the parent starts the N ranks before it touches
the GPU, relays rank 0's JSON line and fails if any rank fails.  CPU: the launcher with a stand-in rank (gloo, world 2 and
3, sharded gather); GPU: the real bench with two ranks on one device."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

import bench

STUB = os.path.join(ROOT, "tests", "helpers", "rank_stub.py")


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 8])  # 8: the driver's scaling run (one rank per GPU of a node), rehearsed on the CPU
def test_launcher_starts_ranks_and_relays_rank0(world):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    rc, out = bench.spawn_ranks(world, [sys.executable, STUB], env=env, timeout=240)
    assert rc == 0, out
    rec = json.loads(out.strip().splitlines()[-1])
    assert rec["n"] == 2 * world and rec["ids"] == list(range(2 * world))
    assert rec["local_rank"] == "0" and rec["master"] == "127.0.0.1"


@pytest.mark.timeout(300)
def test_launcher_reports_a_failing_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    rc, _ = bench.spawn_ranks(2, [sys.executable, STUB, "--fail-rank", "1"], env=env, timeout=120)
    assert rc != 0


def test_parent_does_not_import_torch_before_spawning():
    """The launcher branch sits above the first torch import (a parent that initialised HIP could not fork ranks safely)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index("spawn_ranks(args.gpus") < src.index("import torch")
    assert "import torch" not in src[:src.index("def main()")]


def _run_bench(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_device_from_plain_python():
    d = _run_bench(["--gpus", "2", "--backend", "gloo", "--single-device", "--workload", "er-5pct-2k", "--steps", "8", "--warmup", "2",
                    "--cpu-iters", "0"])
    assert d["n_gpus"] == 2 and d["config"]["instances"] == 2 and d["scaling"] == "weak"
    assert len(d["objectives"]["max_violation_per_instance"]) == 2 and d["value"] > 0
    assert "roofline" in d and "coloring" not in d


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_instances_per_gpu_config3_shape():
    d = _run_bench(["--workload", "er-5pct-2k", "--instances-per-gpu", "4", "--steps", "8", "--warmup", "2", "--cpu-iters", "0"])
    assert d["n_gpus"] == 1 and d["config"]["instances"] == 4 and d["config"]["instances_per_gpu"] == 4
    assert len(d["objectives"]["max_violation_per_instance"]) == 4
    assert abs(d["instances_per_s"] * 8 - d["value"]) < 1e-2 * d["value"] + 1.0


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_line_carries_the_round3_fields_and_the_device_state_path():
    """One short run with the instance generated on the GPU and handed over there (--state device): the line still carries every
    contract field, the spread of the timed regions, the operand disclosure and the three colouring records."""
    d = _run_bench(["--steps", "8", "--warmup", "4", "--cpu-iters", "2", "--repeats", "3", "--state", "device", "--no-fp32-operands"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "value_min", "value_max", "timed_regions", "region_tail_ms"):
        assert k in d, k
    assert d["timed_regions"] == 3 and d["value_min"] <= d["value"] <= d["value_max"]
    assert "device-resident" in d["config"]["state"] and "operand_precision" in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert d["coloring"]["warm_start"] is False and d["coloring"]["rem"] == 0
    assert d["coloring_device_state"]["rem"] == 0 and "device-resident" in d["coloring_device_state"]["state"]
    assert d["coloring_warm_start"]["warm_start"] is True
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
