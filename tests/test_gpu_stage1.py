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


def _run(case, b1=None, lanes=1):
    """lanes: 1 = one curve per lane (the throughput kernel), 2 = X and Z of a curve on adjacent lanes, 8 = X and Z
    on two quads of lanes with the limbs of each residue spread over the quad, 32 = X and Z on two DPP rows with
    the limbs over the 16 lanes of a row (what the library picks by itself for batches this small) —
    include/gecm.h gecm_set_lanes_per_curve"""
    import pyecm
    n = _n_of(case)
    eng = pyecm.Engine(n, digitbits=case["digitbits"])
    sig = [int(l.split("SIGMA=")[1].split(";")[0]) for l in case["save_lines"]]
    eng.build_curves(sig)
    eng.set_lanes_per_curve(lanes)
    eng.stage1(b1 or case["B1"])
    assert eng.lanes_per_curve() == lanes
    lines = [l.rstrip("\n") for l in eng.save_lines()]
    st = eng.stage1_stats()
    facs = [eng.stage1_factor(k) for k in range(len(sig))]
    cfg = eng.cfg
    eng.close()
    return lines, st, facs, cfg


SMALL = [c for c in CASES if c["B1"] <= 100000]
BIG = [c for c in CASES if c["B1"] > 100000]


@pytest.mark.parametrize("lanes", [1, 2, 8, 32])
@pytest.mark.parametrize("case", SMALL, ids=[c["name"] for c in SMALL])
def test_stage1_save_lines_small(case, lanes):
    lines, st, facs, cfg = _run(case, lanes=lanes)
    assert cfg.nwords == case["nwords"] and cfg.maxbits == case["maxbits"]
    assert st.ptadds == case["ptadds"] and st.ptdups == case["ptdups"]
    assert lines == case["save_lines"]


@pytest.mark.parametrize("case", BIG, ids=[c["name"] for c in BIG])
def test_stage1_save_lines_b1_1e6(case):
    lines, st, facs, cfg = _run(case)
    assert st.ptadds == 1980817 and st.ptdups == 217929
    assert lines == case["save_lines"]


BIG2 = [c for c in BIG if c["name"] in ("K1", "n623_b1_1000000", "config1_fib791")]


@pytest.mark.parametrize("case", BIG2, ids=[c["name"] for c in BIG2])
def test_stage1_save_lines_b1_1e6_two_lanes_per_curve(case):
    lines, st, facs, cfg = _run(case, lanes=2)
    assert lines == case["save_lines"]


BIG8 = [c for c in BIG if c["name"] in ("K2", "n831_b1_1000000", "n415_b1_1000000")]


@pytest.mark.parametrize("case", BIG8, ids=[c["name"] for c in BIG8])
def test_stage1_save_lines_b1_1e6_eight_lanes_per_curve(case):
    lines, st, facs, cfg = _run(case, lanes=8)
    assert lines == case["save_lines"]


BIG32 = [c for c in BIG if c["name"] in ("K1", "n415_b1_1000000", "n831_b1_1000000", "config1_fib791")]


@pytest.mark.parametrize("case", BIG32, ids=[c["name"] for c in BIG32])
def test_stage1_save_lines_b1_1e6_32_lanes_per_curve(case):
    lines, st, facs, cfg = _run(case, lanes=32)
    assert lines == case["save_lines"]


def test_lanes_per_curve_is_chosen_from_the_batch_size():
    """auto mode: 32 lanes per curve for small batches, then 8 or 2, one when the batch fills whole rounds of 2
    wavefronts per SIMD (256 CUs); the lines do not depend on the layout"""
    import pyecm
    case = next(c for c in CASES if c["name"] == "K1N_two_full_batches_b1_500")
    eng = pyecm.Engine(_n_of(case), digitbits=52)
    assert eng.lanes_per_curve() == 0
    eng.build_curves(list(range(100, 116)))
    eng.stage1(500)
    assert eng.lanes_per_curve() == 32                      # 16 curves: far below 32 curves per CU
    small = [l.rstrip("\n") for l in eng.save_lines()]
    eng.build_curves(list(range(100, 100 + 12000)))       # 47 curves per CU: still 32 lanes (gecm_dev_auto_lanes)
    eng.stage1(500)
    assert eng.lanes_per_curve() == 32
    assert [l.rstrip("\n") for l in eng.save_lines()[:16]] == small
    eng.build_curves(list(range(100, 100 + 15000)))       # 59 curves per CU: past the 32-lane layout's range
    eng.stage1(500)
    assert eng.lanes_per_curve() == (8 if eng.cfg.dev_limbs >= 19 else 2)
    assert [l.rstrip("\n") for l in eng.save_lines()[:16]] == small
    eng.build_curves(list(range(100, 100 + 20000)))       # too many for 8 lanes each, too few for whole rounds of 1
    eng.stage1(500)
    assert eng.lanes_per_curve() == 2
    assert [l.rstrip("\n") for l in eng.save_lines()[:16]] == small
    eng.build_curves(list(range(100, 100 + 131072)))     # a full round of 2 wavefronts on each of 1024 SIMDs
    eng.stage1(500)
    assert eng.lanes_per_curve() == 1
    assert [l.rstrip("\n") for l in eng.save_lines()[:16]] == small == case["save_lines"]
    for bad in (3, 4, 16, 64, -1):
        with pytest.raises(pyecm.GecmError):
            eng.set_lanes_per_curve(bad)
    eng.close()


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
