"""Cunningham-type inputs (N | 2^k -/+ 1) on the GPU.

The reference strips algebraic factors and then either keeps REDC (small cofactor) or switches to
multiplication modulo 2^k -/+ 1 (main.c:405-527):
 * REDC case: save_b1.txt of the avx-ecm driver is byte-identical to the reference's, banner lines included;
 * special-reduction case: the reference works modulo Mw = 2^k -/+ 1 throughout and so does the driver (contexts on Mw,
   N as report modulus): byte-identical files (tests/test_gpu_special.py has nine such runs).  A context created on N
   itself computes modulo N, with the special-form multiply where it pays: its residues are the TRUE point [k]P modulo N
   on every lane (checked against an independent x-only ladder, tests/xladder.py) and equal the reference's modulo N on
   every lane but one — sigma 1006, where the reference's curve set-up inversion fails modulo Mw (Mw has the factors
   503 and 54217 that N has not) and it goes on with a stale operand: its point there is not [k]P."""
import json
import os
import subprocess
import tempfile

import pytest

from conftest import GOLDEN, ROOT
from xladder import true_stage1_point

pytestmark = pytest.mark.gpu
FIX = json.load(open(os.path.join(GOLDEN, "inputs.json")))
RUNS = {c["name"]: c for c in FIX["runs"]}
EXE = os.path.join(ROOT, "avx-ecm_amd", "avx-ecm")


def _field(line, key):
    return int(line.split(key + "=0x")[1].split(";")[0], 16)


def test_redc_case_driver_output_is_the_references():
    c = RUNS["redc_phi105_phi210"]
    banner = next(b for b in FIX["banner"] if b["expr"] == c["N"])
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([EXE, c["N"], "8", str(c["B1"]), "1", str(c["B2"]), str(c["sigma0"])], cwd=d,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        save = open(os.path.join(d, "save_b1.txt")).read().splitlines()
        res = [l for l in open(os.path.join(d, "ecm_results.txt")).read().splitlines() if l.strip()]
    assert save == c["save_lines"]
    assert res == c["results_lines"]
    out = p.stdout.splitlines()
    pos = [out.index(l) for l in banner["lines"]]          # every banner line, in the reference's order
    assert pos == sorted(pos)
    assert "Choosing MAXBITS = %d, NWORDS = %d, NBLOCKS = %d based on input size 97" % (c["maxbits"], c["nwords"], c["nwords"] // 4) in p.stdout


def test_special_reduction_case_residues_equal_the_references_modulo_n():
    import pyecm
    c = RUNS["special_m251_cofactor"]
    n = int(c["N"])
    sig = [int(l.split("SIGMA=")[1].split(";")[0]) for l in c["save_lines"]]
    good = c["reference_lane_is_the_true_point"]
    assert sum(good) == 7 and not good[6]
    for lanes in (1, 2):
        eng = pyecm.Engine(n, digitbits=52)
        eng.build_curves(sig)
        eng.set_lanes_per_curve(lanes)
        eng.stage1(c["B1"])
        assert eng.special_form_used()                     # N | 2^251 - 1: the F-form kernel ran
        mine = eng.save_lines()
        eng.close()
        for k, (l, r) in enumerate(zip(mine, c["save_lines"])):
            assert _field(l, "N") == _field(r, "N") == n
            X, Z = true_stage1_point(n, sig[k], c["B1"])
            assert (X * _field(l, "Z") - _field(l, "X") * Z) % n == 0
            if good[k]:
                assert _field(l, "X") == _field(r, "X") % n and _field(l, "Z") == _field(r, "Z") % n


def test_driver_on_a_mersenne_cofactor_writes_the_references_file():
    c = RUNS["special_m251_cofactor"]
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([EXE, c["N"], "8", str(c["B1"]), "1", str(c["B2"]), str(c["sigma0"])], cwd=d,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        save = open(os.path.join(d, "save_b1.txt")).read().splitlines()
    assert "removing algebraic C1 factor 0" in p.stdout                       # what the reference prints here
    assert "Using special Mersenne mod for factor of: 2^251-1" in p.stdout    # main.c:644-670
    assert "Choosing MAXBITS = 416, NWORDS = 8, NBLOCKS = 2 based on input size 251" in p.stdout
    assert save == c["save_lines"]                                             # all eight lanes, modulo 2^251 - 1
