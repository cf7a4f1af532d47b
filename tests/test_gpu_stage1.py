"""GPU parity: stage-1 residues from the HIP path, through the C ABI, against the save_b1.txt
lines the REFERENCE wrote for the same (N, B1, sigma) — tests/golden/stage1.json (made by
tests/golden/make_golden.py from oracle/_ref).  Bar: byte-identical lines."""
import json
import os

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

CASES = json.load(open(os.path.join(GOLDEN, "stage1.json")))


def _n_of(case):
    # the fixture stores the command-line expression; the save line stores the evaluated N
    line = case["save_lines"][0]
    return int(line.split("N=0x")[1].split(";")[0], 16)


def _run(case, b1=None):
    import pyecm
    n = _n_of(case)
    eng = pyecm.Engine(n, digitbits=case["digitbits"])
    sig = [int(l.split("SIGMA=")[1].split(";")[0]) for l in case["save_lines"]]
    eng.build_curves(sig)
    eng.stage1(b1 or case["B1"])
    lines = [l.rstrip("\n") for l in eng.save_lines()]
    st = eng.stage1_stats()
    facs = [eng.stage1_factor(k) for k in range(len(sig))]
    cfg = eng.cfg
    eng.close()
    return lines, st, facs, cfg


SMALL = [c for c in CASES if c["B1"] <= 100000]
BIG = [c for c in CASES if c["B1"] > 100000]


@pytest.mark.parametrize("case", SMALL, ids=[c["name"] for c in SMALL])
def test_stage1_save_lines_small(case):
    lines, st, facs, cfg = _run(case)
    assert cfg.nwords == case["nwords"] and cfg.maxbits == case["maxbits"]
    assert st.ptadds == case["ptadds"] and st.ptdups == case["ptdups"]
    assert lines == case["save_lines"]


@pytest.mark.parametrize("case", BIG, ids=[c["name"] for c in BIG])
def test_stage1_save_lines_b1_1e6(case):
    lines, st, facs, cfg = _run(case)
    assert st.ptadds == 1980817 and st.ptdups == 217929
    assert lines == case["save_lines"]


def test_stage1_factors_match_reference_results():
    """factor lines of ecm_results.txt (ecm.c:1362-1366): same factor, same PRP/C tag, same lane."""
    import re
    for case in SMALL:
        lines, st, facs, cfg = _run(case)
        want = {}
        for l in case["results_lines"]:
            m = re.match(r"found (PRP|C)(\d+) factor (\d+) in stage 1 .*vec (\d+), sigma (\d+)", l)
            if m:
                want[int(m.group(4))] = (int(m.group(3)), m.group(1) == "PRP")
        got = {k: f for k, f in enumerate(facs) if f is not None}
        assert got == want, case["name"]
