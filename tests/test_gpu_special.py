"""Special-form inputs end to end, against the reference's own files (tests/golden/special.json, made by
tests/golden/make_golden.py --only special): N | 2^k - 1, N | 2^k + 1 and 2^k = c (mod N) for which the reference
leaves REDC (main.c:505-527, 642-684).  It then works modulo Mw = 2^k -/+ 1 or 2^k - c throughout — curve construction,
stage 1, stage 2 — and keeps the number given for "N=" and for its factor checks (ecm.c:1111-1118).  The driver makes
its contexts on Mw with N as report modulus: save_b1.txt and ecm_results.txt are the reference's byte for byte, on every
lane — also where a set-up inversion fails modulo Mw (the cofactor of M251: Mw has the factors 503 and 54217 that N has
not) and the reference goes on with a stale operand."""
import json
import os
import subprocess
import tempfile

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
CASES = json.load(open(os.path.join(GOLDEN, "special.json")))
EXE = os.path.join(ROOT, "avx-ecm_amd", "avx-ecm")


def _mw(c):
    sp = c["special"]
    return (1 << sp["k"]) + (sp["c"] if sp["sign"] == "+" else -sp["c"])


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_driver_writes_the_reference_files_for_special_form_inputs(case):
    c = case
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([EXE, c["N"], "8", str(c["B1"]), "1", str(c["B2"]), str(c["sigma0"])], cwd=d, capture_output=True,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        save = open(os.path.join(d, "save_b1.txt")).read().splitlines()
        res = [l for l in open(os.path.join(d, "ecm_results.txt")).read().splitlines() if l.strip()] \
            if os.path.exists(os.path.join(d, "ecm_results.txt")) else []
    assert save == c["save_lines"]
    assert res == c["results_lines"]
    got = p.stdout.splitlines()
    for l in c["stdout_lines"]:
        if l.startswith(("gen:", "removing", "commencing parallel", "Choosing MAXBITS", "Input has", "Using special", "performed ",
                         "found ", "Stage 1 completed")):
            assert l in got, l


@pytest.mark.parametrize("lanes", [1, 2, 8, 32])
def test_context_on_the_special_modulus_reports_against_n(lanes):
    """the library seam of the same: a context on Mw, N as report modulus; every lane layout (one and two lanes per curve
    multiply with the special reduction, the many-lane layouts by REDC modulo Mw); factors are those of N, never of Mw/N"""
    import pyecm
    c = [x for x in CASES if x["name"] == "M251_cofactor_b1_20000_stage2"][0]
    n = int(c["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
    mw = _mw(c)
    assert mw % n == 0 and mw // n == 503 * 54217
    eng = pyecm.Engine(mw)
    with pytest.raises(pyecm.GecmError):
        eng.set_report_modulus(n + 2)                       # must divide the modulus
    eng.set_report_modulus(n)
    eng.set_lanes_per_curve(lanes)
    sig = [int(l.split("SIGMA=")[1].split(";")[0]) for l in c["save_lines"]]
    eng.build_curves(sig + list(range(50000, 50064)))       # the eight reference curves and a wavefront more
    eng.stage1(c["B1"])
    assert eng.special_form_used() == (lanes in (1, 2))
    assert [eng.save_line(k).rstrip("\n") for k in range(8)] == c["save_lines"]
    nf, first = eng.scan_factors(1)
    for k in range(72):
        f = eng.stage1_factor(k)
        assert eng.curve_flag(1, k) == (f is not None)
        if f:
            assert n % f[0] == 0 and 1 < f[0] < n
    eng.stage2(c["B2"])
    eng.scan_factors(2)
    for k in range(72):
        f = eng.stage2_factor(k)
        assert eng.curve_flag(2, k) == (f is not None)
        if f:
            assert n % f[0] == 0 and 1 < f[0] < n
    eng.close()
