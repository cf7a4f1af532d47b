"""GPU test of the avx-ecm command line (host/avx_ecm_main.c): same positional arguments as the
reference; save_b1.txt, ecm_results.txt and checkpoint.txt byte-identical to the files the REFERENCE wrote
(tests/golden/stage1.json, batches.json, multirange.json — made by tests/golden/make_golden.py), for runs of one and
of several reference batches, one and several threads, one and several prime ranges."""
import hashlib
import json
import os
import subprocess
import tempfile

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "avx-ecm_amd", "avx-ecm")
S1 = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "stage1.json")))}
BATCHES = json.load(open(os.path.join(GOLDEN, "batches.json")))
MULTI = json.load(open(os.path.join(GOLDEN, "multirange.json"))) if os.path.exists(os.path.join(GOLDEN, "multirange.json")) else []


def _run(args, env=None):
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([EXE] + [str(a) for a in args], cwd=d, capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, **(env or {})))
        assert p.returncode == 0, p.stdout + p.stderr
        save = open(os.path.join(d, "save_b1.txt")).read() if os.path.exists(os.path.join(d, "save_b1.txt")) else ""
        res = open(os.path.join(d, "ecm_results.txt")).read() if os.path.exists(os.path.join(d, "ecm_results.txt")) else ""
    return p.stdout, save, [l for l in res.splitlines() if l.strip()]


@pytest.mark.parametrize("name", ["n415_b1_10000", "n831_b1_10000", "K1N_two_full_batches_b1_500", "n415_b1_10000_b2_1e6"])
def test_cli_save_file_and_results(name):
    c = S1[name]
    out, save, res = _run([c["N"], c["curves"], c["B1"], 1, c["B2"], c["sigma0"]])
    assert hashlib.sha256(save.encode()).hexdigest() == c["save_sha256"]
    assert "Stage 1 completed at prime" in out and "with %d point-adds and %d point-doubles" % (c["ptadds"], c["ptdups"]) in out
    assert "Choosing MAXBITS = %d, NWORDS = %d" % (c["maxbits"], c["nwords"]) in out
    assert save.splitlines() == c["save_lines"]
    assert res == c["results_lines"]
    if c["stage2_counts"]:
        assert "performed %d pt-adds, %d inversions, and %d pair-muls in stage 2" % tuple(c["stage2_counts"]) in out


def test_cli_expression_input_config1():
    """BASELINE configs[0]: the reference's own command line, input given as an expression"""
    c = S1["config1_fib791"]
    out, save, res = _run(["fib(791)/13/677/216416017", 8, 20000, 1, 20000, 1000])
    n = int(c["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
    assert "commencing parallel ecm on %d" % n in out
    assert "Choosing MAXBITS = 624, NWORDS = 12, NBLOCKS = 3 based on input size 508" in out
    assert len(save.splitlines()) == 8 and all("N=0x%x;" % n in l for l in save.splitlines())


def test_cli_usage():
    p = subprocess.run([EXE], capture_output=True, text=True)
    assert p.returncode == 1 and "usage: avx-ecm" in p.stdout


def test_cli_rounds_curves_up_to_a_multiple_of_8():
    """main.c:585-589 / ecm.c:1151: the reference runs whole 8-lane vectors; 10 curves -> 16 lines"""
    c = S1["K1N_two_full_batches_b1_500"]          # N without small factors, sigma 100.., 16 lines in the fixture
    out, save, res = _run([c["N"], 10, c["B1"], 1, c["B2"], c["sigma0"]])
    assert save.splitlines() == c["save_lines"] and len(c["save_lines"]) == 16


def test_cli_more_curves_than_one_pass():
    """more curves than one pass holds (131072 per GPU): the driver runs a second pass and the sigmas continue
    (main.c:761, ecm.c:1187); N without small factors so that no factor stops the run after pass one"""
    import pyecm
    c = S1["K1N_two_full_batches_b1_500"]
    n = int(c["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
    curves, b1, sigma0 = 131072 + 16, 30, 500000
    out, save, res = _run([c["N"], curves, b1, 1, b1, sigma0])
    lines = save.splitlines()
    assert len(lines) == curves and "Commencing curves 131072-131087 of 131088" in out
    sig = [int(l.split("SIGMA=")[1].split(";")[0]) for l in lines]
    assert sig == list(range(sigma0, sigma0 + curves))
    eng = pyecm.Engine(n)
    pick = [0, 1, 131071, 131072, 131087]
    eng.build_curves([sigma0 + k for k in pick])
    eng.stage1(b1)
    assert [l.rstrip("\n") for l in eng.save_lines()] == [lines[k] for k in pick]
    eng.close()


def test_cli_prints_the_reference_banner_lines():
    """main.c:529-533, 558, 591-593 and the stage-2 lines of ecm.c:2440-2442, 2568, 2904-2905, 1421, 1462, 1481-1483,
    verbatim (values as the reference printed them for this command: tests/golden/cli_stdout.json)"""
    want = json.load(open(os.path.join(GOLDEN, "cli_stdout.json")))
    out, save, res = _run(want["args"])
    import re
    norm = lambda l: re.sub(r"[0-9]+\.[0-9]+ seconds", "T seconds", l)
    got = [norm(l) for l in out.splitlines()]
    for l in want["lines"]:
        assert norm(l) in got, l


@pytest.mark.parametrize("case", BATCHES, ids=[c["name"] for c in BATCHES])
def test_cli_reference_batches(case):
    """More curves than one reference batch (8 x threads).  The reference works batch by batch and stops after the
    first one in which a curve found a factor (ecm.c:1531-1532); every thread of a batch runs the same eight sigmas
    when sigma is given (ecm.c:1187), and a factor is labelled curve threads*curve + j*8 + i, thread j, vec i
    (ecm.c:1356-1366).  One pass on the GPU holds all those batches; the files must be the reference's."""
    c = case
    out, save, res = _run([c["N"], c["curves"], c["B1"], c["threads"], c["B2"], c["sigma0"]])
    assert save.splitlines() == c["save_lines"]
    assert res == c["results_lines"]
    want_out = [l for l in c["stdout_lines"] if l.startswith(("found ", "Input has", "Choosing MAXBITS", "performed "))]
    got = out.splitlines()
    for l in want_out:
        assert l in got, l
    # and the same files when the batches are spread over several passes, pipelined or not
    for env in ({"GECM_PASS_CURVES": "8"}, {"GECM_PASS_CURVES": "16", "GECM_NO_PIPELINE": "1"},
                {"GECM_PASS_CURVES": "24", "GECM_CONTEXTS_PER_GPU": "2"}):
        out2, save2, res2 = _run([c["N"], c["curves"], c["B1"], c["threads"], c["B2"], c["sigma0"]], env=env)
        assert (save2, res2) == (save, res), env


def test_cli_fourth_argument_is_the_reference_s_thread_count():
    """argv[4] is `threads` as in the reference: it rounds the curve count (main.c:585-589: per thread, whole vectors
    of 8), sets the labels and the number of lines per batch, and is printed; it neither selects GPUs nor fails when
    it exceeds them."""
    c = S1["K1N_two_full_batches_b1_500"]
    out, save, res = _run([c["N"], 20, c["B1"], 16, c["B2"], c["sigma0"]])
    assert "using 16 threads (2 curves/thread)" in out
    lines = save.splitlines()
    assert len(lines) == 16 * 8                                       # 2 curves per thread -> one vector of 8 each
    assert lines == c["save_lines"][:8] * 16                          # every thread: the reference's eight sigmas


def _run_many(cmds, exes=None):
    """several driver processes at once (each a handful of curves: the GPU has room), one directory each"""
    import shutil
    dirs, procs = [], []
    for i, args in enumerate(cmds):
        d = tempfile.mkdtemp()
        dirs.append(d)
        procs.append(subprocess.Popen([exes[i] if exes else EXE] + [str(a) for a in args], cwd=d, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for d, p in zip(dirs, procs):
        out, _ = p.communicate(timeout=1100)
        assert p.returncode == 0, out
        rd = lambda f: open(os.path.join(d, f)).read().splitlines() if os.path.exists(os.path.join(d, f)) else None
        outs.append((out, rd("save_b1.txt"), rd("checkpoint.txt"), [l for l in (rd("ecm_results.txt") or []) if l.strip()]))
        shutil.rmtree(d)
    return outs


@pytest.mark.skipif(not MULTI, reason="tests/golden/multirange.json not generated")
def test_cli_multirange_b1_above_1e8():
    """B1 = 1.1e8: two prime ranges (ecm.c:1209-1312).  checkpoint.txt after the first (B1 field = 99999989), the
    doublings repeated and 100000007 skipped in the second, then save_b1.txt — and, in the second case, stage 2 from
    B1 to 1.3e8.  Byte for byte the files of the reference (three minutes of its time per case; about as long here:
    eight curves are one wavefront's worth of latency)."""
    cmds = [[c["N"], c["curves"], c["B1"], 1, c["B2"], c["sigma0"]] for c in MULTI]
    # and, at the same time, the REFERENCE's own program with its four work functions on the GPU (oracle/_ref/avx-ecm-52-l1,
    # tests/test_gpu_dropin.py): its vececm drives the two ranges and writes checkpoint.txt itself
    l1 = os.path.join(ROOT, "oracle", "_ref", "avx-ecm-52-l1")
    outs = _run_many(cmds + ([cmds[0]] if os.path.exists(l1) else []), exes=[EXE] * len(cmds) + [l1])
    if os.path.exists(l1):
        out, save, ckpt, res = outs[-1]
        assert ckpt == MULTI[0]["checkpoint_lines"] and save == MULTI[0]["save_lines"] and res == MULTI[0]["results_lines"]
        assert "Saving checkpoint after p=99999989" in out
    for c, (out, save, ckpt, res) in zip(MULTI, outs):
        assert ckpt == c["checkpoint_lines"], c["name"]
        assert save == c["save_lines"], c["name"]
        assert res == c["results_lines"], c["name"]
        got = out.replace("\r", "\n").splitlines()
        for l in c["stdout_lines"]:
            if l.startswith(("Found ", "Commencing Stage 1 @", "Stage 1 completed", "Saving checkpoint", "performed ", "found ")):
                assert l in got, (c["name"], l)


def test_cli_two_gpus_write_the_same_files():
    """one host thread and one context per GPU (GECM_GPUS caps how many are used): save_b1.txt and ecm_results.txt
    do not depend on the number of GPUs.  Skipped on a one-GPU box."""
    import pyecm
    if pyecm.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    c = S1["K1N_two_full_batches_b1_500"]
    one = _run([c["N"], 64, c["B1"], 1, 20000, c["sigma0"]], env={"GECM_GPUS": "1"})
    two = _run([c["N"], 64, c["B1"], 1, 20000, c["sigma0"]], env={"GECM_GPUS": "2"})
    assert "2 GPU(s)" in two[0] and "1 GPU(s)" in one[0]
    assert one[1] == two[1] and len(one[1].splitlines()) == 64
    assert one[2] == two[2]


@pytest.mark.parametrize("contexts", [2, 3])
def test_cli_several_contexts_write_the_same_files(contexts):
    """The same multi-context path on whatever the box has: GECM_CONTEXTS_PER_GPU puts several contexts, each with its
    host thread, on one device (curves split between them, the first stage-2 range's pair map shared, every context's
    tape prepared while its stage-1 kernel runs).  save_b1.txt and ecm_results.txt equal the one-context run's."""
    c = S1["K1N_two_full_batches_b1_500"]
    one = _run([c["N"], 72, c["B1"], 1, 20000, c["sigma0"]], env={"GECM_GPUS": "1"})
    many = _run([c["N"], 72, c["B1"], 1, 20000, c["sigma0"]], env={"GECM_GPUS": "1", "GECM_CONTEXTS_PER_GPU": str(contexts)})
    assert "%d GPU(s)" % contexts in many[0] and "1 GPU(s)" in one[0]
    assert one[1] == many[1] and len(one[1].splitlines()) == 72
    assert one[2] == many[2]
    assert one[1].splitlines()[:len(c["save_lines"])] == c["save_lines"]


def test_cli_pipelined_passes_write_the_same_file():
    """several passes: two sets of contexts alternate, the host work of a pass (curves, lines, files) overlaps the
    kernels of its neighbours; the file is the one a single pass writes"""
    c = S1["K1N_two_full_batches_b1_500"]
    one = _run([c["N"], 4096, 300, 1, 300, 7000])
    piped = _run([c["N"], 4096, 300, 1, 300, 7000], env={"GECM_PASS_CURVES": "512"})
    serial = _run([c["N"], 4096, 300, 1, 300, 7000], env={"GECM_PASS_CURVES": "512", "GECM_NO_PIPELINE": "1"})
    assert len(one[1].splitlines()) == 4096 and one[1] == piped[1] == serial[1]
    assert "Commencing curves 3584-4095 of 4096" in piped[0]


def test_cli_without_sigma_draws_one_per_lane_and_thread():
    """no sigma on the command line: every lane of every thread draws its own (ecm.c:1564-1570), so a batch is
    8 x threads distinct curves; two runs differ"""
    c = S1["K1N_two_full_batches_b1_500"]
    outs = [_run([c["N"], 40, 200, 3, 200]) for _ in range(2)]
    for out, save, res in outs:
        lines = save.splitlines()
        assert len(lines) == 48 and "using 3 threads (14 curves/thread)" in out      # 14 per thread -> two batches of 24 lines
        sig = [int(l.split("SIGMA=")[1].split(";")[0]) for l in lines]
        assert len(set(sig)) == 48 and min(sig) >= 6
    assert outs[0][1] != outs[1][1]


def test_batch_memory_figure_is_what_a_batch_takes():
    """gecm_batch_bytes (what the driver sizes its passes with) against what a batch really takes on the device"""
    import pyecm
    n = int(S1["K1N_two_full_batches_b1_500"]["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
    for curves in (4096, 32768):
        eng = pyecm.Engine(n)
        free0, total = eng.device_memory()
        assert total > 100 * 2 ** 30 and free0 <= total
        eng.build_curves(list(range(1000, 1000 + curves)))
        eng.stage1(300)
        eng.scan_factors(1)
        free1, _ = eng.device_memory()
        eng.stage2(30000)
        free2, _ = eng.device_memory()
        s1, s2 = eng.batch_bytes(curves, False), eng.batch_bytes(curves, True, 300)
        slack = 512 << 20                                            # the runtime's own pools, tape, constants, granularity
        assert 0 < (free0 - free1) <= s1 + slack
        assert s2 - slack <= (free0 - free2) <= s2 + slack, (curves, free0 - free2, s2)
        if curves == 32768:
            assert 0.9 * s2 <= (free0 - free2) <= 1.1 * s2, (free0 - free2, s2)
        eng.close()
    # BASELINE configs[3]'s slice and the largest class: what a full pass takes
    eng = pyecm.Engine(n)
    full = eng.batch_bytes(131072, True, 1000000)
    assert 80e9 < full < 95e9        # 60 GB of table (7683 entries x 15 limbs x 4 B x 131072 curves) + ring, chunk and block scratch
    assert eng.batch_bytes(131072, False) < 0.01 * full
    eng.close()


EXE32 = os.path.join(ROOT, "avx-ecm_amd", "avx-ecm-32")


@pytest.mark.parametrize("name", ["n1023_d32_b1_1000", "n1023_d32_b1_10000", "n415_d32_b1_10000", "K1N_d32_b1_2000"])
def test_cli_32_bit_flavour_writes_what_the_reference_s_32_bit_build_writes(name):
    """avx-ecm-32 = the reference built with DIGITBITS = 32 (avx_ecm.h:80-89): vectors of 16 curves — a batch is
    16 x threads lines, so a run that finds a factor stops after 16 — and the 128-bit NWORDS rule in the banner;
    fixtures from oracle/_ref/avx-ecm-32"""
    c = S1[name]
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([EXE32, c["N"], str(c["curves"]), str(c["B1"]), "1", str(c["B2"]), str(c["sigma0"])], cwd=d,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        save = open(os.path.join(d, "save_b1.txt")).read()
        res = [l for l in open(os.path.join(d, "ecm_results.txt")).read().splitlines() if l.strip()] \
            if os.path.exists(os.path.join(d, "ecm_results.txt")) else []
    assert hashlib.sha256(save.encode()).hexdigest() == c["save_sha256"] and len(save.splitlines()) == 16
    assert res == c["results_lines"]
    assert "ECM has been configured with DIGITBITS = 32, VECLEN = 16, GMP_LIMB_BITS = 64" in p.stdout
    assert "Choosing MAXBITS = %d, NWORDS = %d, NBLOCKS = %d" % (c["maxbits"], c["nwords"], c["nwords"] // 4) in p.stdout
    assert "with %d point-adds and %d point-doubles" % (c["ptadds"], c["ptdups"]) in p.stdout
