"""bench.py on the GPU box: as a plain process and as a rank under torch.distributed.run (a world of one still makes
the RCCL process group, so init, barrier and the found-record all-reduce on device tensors run on hardware)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
ARGS = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--b1", "20000", "--b2", "400000", "--curves", "256", "--no-extras",
        "--no-cpu-baseline"]


def _line(cmd):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return lines[0]


def test_bench_line_has_the_contract_s_fields():
    d = _line([sys.executable, "bench.py"] + ARGS)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "curves/s" and d["value"] > 0
    assert d["config"]["curves_per_gpu"] == 256 and d["config"]["lanes_per_curve"] == 32
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg")) <= set(d["roofline"])
    assert d["stage2"]["B2"] == 400000 and d["stage2"]["seconds"] > 0


def test_bench_as_a_rank_under_torch_distributed_run():
    d = _line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", "29541", "bench.py"] + ARGS)
    assert d["n_gpus"] == 1 and "RCCL" in d["config"]["parallelism"] and d["value"] > 0
    # the found-record all-reduce really ran, over RCCL, on a device tensor: once per step, warm-up included
    r = d["ranks"]
    assert r["process_group"] == "nccl" and r["world_size_reported_by_process_group"] == 1
    assert r["found_record_allreduces_per_rank_timed_and_warmup"] == [3]
    assert r["per_rank"][0]["global_curves"] == [0, 256] and r["per_rank"][0]["kernel_ms_avg"] > 0


def test_bench_two_ranks_with_engines_on_one_gpu():
    """`--gpus 2` from a plain process: the launcher starts two ranks; on a one-GPU box both engines sit on device 0
    (the rank's device is LOCAL_RANK modulo the visible devices) and the collectives go over gloo — RCCL does not put
    two ranks on one device.  The line counts both ranks' curves."""
    d = _line([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo"] + ARGS[2:])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    assert d["config"]["curves_per_gpu"] == 256
    assert "gloo" in d["config"]["parallelism"]
    r = d["ranks"]
    assert r["world_size_reported_by_process_group"] == 2 and r["curves_total"] == 512
    assert [x["global_curves"] for x in r["per_rank"]] == [[0, 256], [256, 512]]
    assert r["found_record_allreduces_per_rank_timed_and_warmup"] == [3, 3]
    one = _line([sys.executable, "bench.py"] + ARGS)
    assert d["config"]["curves_with_factor_last_step"] >= 0 and one["n_gpus"] == 1


def test_bench_four_ranks_rehearsal_on_one_gpu():
    """the driver's multi-GPU launch rehearsed on the box's one GPU: four ranks (the box allows six processes on the
    card), each with its own engine and its own 128 curves on device LOCAL_RANK modulo the visible devices, collectives
    over gloo.  The line adds up the ranks' curves and shows every rank's range and kernel time."""
    d = _line([sys.executable, "bench.py", "--gpus", "4", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--b1", "20000",
               "--b2", "0", "--curves", "128", "--no-extras", "--no-cpu-baseline"])
    r = d["ranks"]
    assert d["n_gpus"] == 4 and r["world_size_reported_by_process_group"] == 4 and r["curves_total"] == 512
    assert [x["global_curves"] for x in r["per_rank"]] == [[128 * k, 128 * (k + 1)] for k in range(4)]
    assert all(x["kernel_ms_avg"] > 0 for x in r["per_rank"])
    assert r["kernel_ms_avg_min_over_ranks"] <= r["kernel_ms_avg_max_over_ranks"]
    assert abs(d["value"] - 512 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
