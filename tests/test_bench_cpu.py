"""CPU checks of bench.py's bookkeeping: the algorithmic work figures it prices the roofline with are the
ones SURVEY.md §8d states, and its host-core detection is sane."""
import importlib.util
import os

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_work_per_curve_matches_survey_8d():
    b = _bench()
    # B1=1e6: A=1,980,817 adds, D=217,929 doublings (ecm.c:1849 counters)
    mul, sqr, mads15, w8 = b.work_per_curve(1980817, 217929, 15, 8)
    assert (mul, sqr) == (8577055, 4397492) and mul + sqr == 12974547
    assert w8 == 1641408616                                   # W(B1=1e6, n=8)
    assert b.work_per_curve(1980817, 217929, 30, 16)[3] == 6322861776   # W(1e6, n=16)
    assert b.work_per_curve(1980817, 217929, 23, 12)[3] == 3602129628   # W(1e6, n=12)
    assert b.work_per_curve(195448, 23269, 37, 32)[3] == 2464221376     # W(1e5, n=32)
    assert mads15 == 8577055 * 465 + 4397492 * 360
    assert abs(b.PEAK_MAD_PER_S - 39.3216e12) < 1e6


def test_host_cores_positive():
    assert 1 <= _bench().host_cores() <= 64


def test_plan_steps_keeps_request_when_it_fits_and_shrinks_otherwise():
    b = _bench()
    assert b.plan_steps(20, 5, 4.0, 400.0) == (4, 20)          # the driver's flags at 4 s per step: 96 s, fits
    assert b.plan_steps(20, 5, 25.3, 400.0) == (1, 14)         # round 1's 131072-curve step would not have
    assert b.plan_steps(20, 5, 500.0, 400.0) == (0, 1) or b.plan_steps(20, 5, 500.0, 400.0) == (1, 1)
    assert b.plan_steps(3, 0, 1.0, 100.0) == (0, 3)


def _run_bench(*args):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, capture_output=True,
                       text=True, timeout=600)
    return p, [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    """--gpus N without a torch.distributed environment: the parent (which touches no GPU) starts N ranks;
    rank 0 prints the one JSON line with n_gpus = world size.  Engine stubbed (--no-engine), gloo."""
    p, lines = _run_bench("--gpus", "2", "--backend", "gloo", "--no-engine", "--steps", "3", "--warmup", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1
    assert lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3 and lines[0]["warmup"] == 2
    assert lines[0]["config"]["curves_per_gpu"] == 4096 and lines[0]["scaling"] == "weak"
    assert "rehearsal" in lines[0]


def test_eight_ranks_rehearsal_line_checks_itself():
    """the N = 8 launch the driver makes, rehearsed on CPU (gloo, engine stubbed): the line carries the world size the
    process group reports, every rank's contiguous curve range and the number of found-record all-reduces each rank
    really ran (one per step, warm-up included)"""
    p, lines = _run_bench("--gpus", "8", "--backend", "gloo", "--no-engine", "--steps", "2", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    d = lines[0]
    r = d["ranks"]
    assert d["n_gpus"] == 8 and r["world_size_reported_by_process_group"] == 8 and r["process_group"] == "gloo"
    assert r["curves_total"] == 8 * 4096
    assert [x["global_curves"] for x in r["per_rank"]] == [[4096 * k, 4096 * (k + 1)] for k in range(8)]
    assert [x["rank"] for x in r["per_rank"]] == list(range(8))
    assert r["found_record_allreduces_per_rank_timed_and_warmup"] == [3] * 8
    assert "gloo" in d["config"]["parallelism"]


def test_single_process_line_says_there_is_no_collective():
    p, lines = _run_bench("--gpus", "1", "--no-engine", "--steps", "1", "--warmup", "0")
    assert p.returncode == 0, p.stderr[-2000:]
    d = lines[0]
    assert "no collective" in d["config"]["parallelism"] and "RCCL" not in d["config"]["parallelism"]
    assert d["ranks"]["process_group"] is None and d["ranks"]["found_record_allreduces_per_rank_timed_and_warmup"] == [0]


def test_gpus_must_match_world_size():
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-engine"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_committed_pmc_summary_is_for_the_headline_kernel():
    """roofline.traffic comes from profiles/pmc_latest.json, and only when kernel, batch and B1 match the run: the
    committed summary must be the one of the headline configuration's kernel (415-bit N: 15 limbs, 32 lanes per curve)."""
    import json
    pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    assert pm["kernel"] == _bench().kernel_name(32, 15)
    assert pm["curves"] == 4096 and pm["B1"] == 1000000
    assert pm["hbm_bytes_per_launch_corrected"] > 0
    # and it names the build it was taken on: bench.py shows its counters only on that build (roofline.pmc says which)
    import re
    assert re.fullmatch(r"K:[0-9a-f]{16} R:[0-9a-f]{16} D:[0-9a-f]{16}", pm["build"]), pm.get("build")
