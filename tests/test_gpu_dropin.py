"""The drop-in claim at the operator seam, exercised with the reference itself.

oracle/_ref/avx-ecm-52-gecm is the REFERENCE program (its own main, vececm, prac, vec_add, vec_duplicate,
stage 2 — compiled from /root/reference in the build container by `make -C oracle refgpu`) with the five
operator pointers of avx_ecm.h:205-209 bound to libgecm's gecm_vec*mod entry points through
oracle/ref_gecm_binding.c (the INTEGRATION.md §2 binding).  Every field operation of the run is a GPU
kernel launch; the save file must be byte-identical to the one the pure AVX-512 reference wrote.

One launch per operator on 8 lanes is a PCIe round trip, so the cases are small."""
import hashlib
import json
import os
import re
import subprocess
import tempfile

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "oracle", "_ref", "avx-ecm-52-gecm")
S1 = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "stage1.json")))}


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/avx-ecm-52-gecm not built (needs /root/reference at build time)")
@pytest.mark.parametrize("name", ["n64_b1_500", "K1N_two_full_batches_b1_500", "n415_b1_1000"])
def test_reference_driver_with_gpu_operators_writes_the_reference_save_file(name):
    c = S1[name]
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([EXE, c["N"], str(c["curves"]), str(c["B1"]), "1", str(c["B2"]), str(c["sigma0"])],
                           cwd=d, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        save = open(os.path.join(d, "save_b1.txt")).read()
        res = [l for l in open(os.path.join(d, "ecm_results.txt")).read().splitlines() if l.strip()] \
            if os.path.exists(os.path.join(d, "ecm_results.txt")) else []
    m = re.search(r"gecm binding: (\d+) operator calls served on (.*)", p.stderr)
    assert m, p.stderr[-2000:]
    # every modular operation of the run went through the library: at least 6 mul/sqr per point-add
    assert int(m.group(1)) >= 6 * c["ptadds"] * (c["curves"] // 8)
    assert hashlib.sha256(save.encode()).hexdigest() == c["save_sha256"]
    assert save.splitlines() == c["save_lines"]
    if c["curves"] == 8:
        assert res == c["results_lines"]


# ---- the production seam: the reference's vececm with its four phase functions on the GPU ----
L1 = os.path.join(ROOT, "oracle", "_ref", "avx-ecm-52-l1")
BATCHES = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "batches.json")))}
S2ACC = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "stage2_acc.json")))}


def _run_l1(c, threads=1):
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([L1, str(c["N"]), str(c["curves"]), str(c["B1"]), str(threads), str(c["B2"]), str(c["sigma0"])],
                           cwd=d, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        save = open(os.path.join(d, "save_b1.txt")).read().splitlines()
        res = [l for l in open(os.path.join(d, "ecm_results.txt")).read().splitlines() if l.strip()] \
            if os.path.exists(os.path.join(d, "ecm_results.txt")) else []
    m = re.search(r"gecm L1 binding: (\d+) curve uploads, (\d+) stage-1 ranges, (\d+) stage-2 inits, (\d+) stage-2 ranges served on", p.stderr)
    assert m, p.stderr[-2000:]
    return p.stdout, save, res, [int(x) for x in m.groups()]


@pytest.mark.skipif(not os.path.exists(L1), reason="oracle/_ref/avx-ecm-52-l1 not built (needs /root/reference at build time)")
@pytest.mark.parametrize("name", ["n415_b1_10000_b2_1e6", "K1N_two_full_batches_b1_500", "K2", "config1_fib791"])
def test_reference_vececm_with_gpu_phases_writes_the_reference_files(name):
    """oracle/_ref/avx-ecm-52-l1 is the REFERENCE program — its main, parser, sieve, pair(), thread pool, curve
    construction, file writers and factor scan, compiled from /root/reference by `make -C oracle refl1` — whose four
    work functions (ecm.c:1130-1133) are served by libgecm's phase functions through oracle/ref_gecm_l1_binding.c:
    gecm_upload_points, gecm_stage1_range, gecm_stage2_init, gecm_stage2_pair with the reference's own pair map.  The
    files it writes are the pure reference's: save_b1.txt, the stage-1 and stage-2 factor lines, the counters."""
    c = S1[name]
    out, save, res, calls = _run_l1(c)
    batches = len(c["save_lines"]) // 8
    assert calls[0] == calls[1] == batches
    assert save == c["save_lines"]
    assert res == c["results_lines"]
    assert "with %d point-adds and %d point-doubles" % (c["ptadds"], c["ptdups"]) in out
    if c["stage2_counts"]:
        assert calls[2] == batches and calls[3] == batches * -(-(c["B2"] - c["B1"]) // 10 ** 8)
        assert "performed %d pt-adds, %d inversions, and %d pair-muls in stage 2" % tuple(c["stage2_counts"]) in out


@pytest.mark.skipif(not os.path.exists(L1), reason="oracle/_ref/avx-ecm-52-l1 not built (needs /root/reference at build time)")
def test_reference_vececm_with_gpu_phases_two_threads():
    """two threads of the reference's pool = two contexts; labels and line order are the reference's own code"""
    c = BATCHES["K1N_32_curves_2_threads_b1_300"]
    out, save, res, calls = _run_l1(c, threads=2)
    assert save == c["save_lines"] and res == c["results_lines"]
    assert calls[0] == calls[1] == 4                                   # two batches x two threads
    c = BATCHES["n415_64_curves_2_threads"]
    out, save, res, calls = _run_l1(c, threads=2)
    assert save == c["save_lines"] and res == c["results_lines"]


@pytest.mark.skipif(not os.path.exists(L1), reason="oracle/_ref/avx-ecm-52-l1 not built (needs /root/reference at build time)")
def test_reference_vececm_with_gpu_phases_at_baseline_size():
    """B1 = 1e6, B2 = 1e8 (BASELINE configs[3]'s parameters) on the reference's stage-2 KAT: its own scan finds the
    PRP31 of test_t35.csh line 46 in the accumulator the GPU left in work->stg2acc"""
    c = S2ACC["T35_46_b1_1e6_b2_1e8"]
    out, save, res, calls = _run_l1(c)
    assert res == c["results_lines"] and len(res) == 1
    assert "performed %d pt-adds, %d inversions, and %d pair-muls in stage 2" % tuple(c["stage2_counts"]) in out
    assert save == S1["T35_46"]["save_lines"]


SPECIAL = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "special.json")))}


@pytest.mark.skipif(not os.path.exists(L1), reason="oracle/_ref/avx-ecm-52-l1 not built (needs /root/reference at build time)")
@pytest.mark.parametrize("name", sorted(SPECIAL))
def test_reference_vececm_with_gpu_phases_special_form_inputs(name):
    """the reference's special-form mode (main.c:642-684: arithmetic modulo 2^k -/+ 1 or 2^k - c, vectors holding plain
    residues, factors checked against the input number) through the same four phase functions: the binding makes the
    context on mdata->n as always and converts the vectors at the seam (oracle/ref_gecm_l1_binding.c scale_vec)"""
    c = SPECIAL[name]
    out, save, res, calls = _run_l1(c)
    assert calls[0] == calls[1] == 1
    assert save == c["save_lines"]
    assert res == c["results_lines"]
    assert "with %d point-adds and %d point-doubles" % (c["ptadds"], c["ptdups"]) in out
    assert "performed %d pt-adds, %d inversions, and %d pair-muls in stage 2" % tuple(c["stage2_counts"]) in out
    for l in c["stdout_lines"]:
        if l.startswith("Using special") or l.startswith("commencing"):
            assert l in out
