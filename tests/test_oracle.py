"""CPU tests: the oracle (oracle/ecm_oracle.c, the scalar-C restatement) pinned against the
REFERENCE's own outputs — tests/golden/l0.json and stage1.json were produced by running
oracle/_ref (the reference compiled from /root/reference).  No GPU needed."""
import ctypes
import json
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, ROOT

ORC_DIR = os.path.join(ROOT, "oracle")


@pytest.fixture(scope="module")
def orc():
    so = os.path.join(ORC_DIR, "libecm_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", ORC_DIR, "oracle"])
    L = ctypes.CDLL(so)
    L.orc_create.restype = ctypes.c_void_p
    L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.orc_destroy.argtypes = [ctypes.c_void_p]
    L.orc_nwords.argtypes = [ctypes.c_void_p]
    L.orc_maxbits.argtypes = [ctypes.c_void_p]
    L.orc_l0_hex.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    L.orc_stage1_line.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t,
                                  ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
    L.orc_stage2.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                             ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t,
                             ctypes.POINTER(ctypes.c_uint64)]
    L.orc_lucas_cost.restype = ctypes.c_double
    L.orc_lucas_cost.argtypes = [ctypes.c_uint64, ctypes.c_double]
    L.orc_prac_choice.argtypes = [ctypes.c_uint64]
    return L


L0 = json.load(open(os.path.join(GOLDEN, "l0.json")))
S1 = json.load(open(os.path.join(GOLDEN, "stage1.json")))


@pytest.mark.parametrize("s", L0, ids=["d%d_n%d_%d" % (s["digitbits"], s["nwords"], i) for i, s in enumerate(L0)])
def test_oracle_l0_against_reference_vectors(orc, s):
    c = orc.orc_create(("0x" + s["N"]).encode(), s["digitbits"])
    assert c
    if orc.orc_nwords(c) != s["nwords"]:
        orc.orc_destroy(c)
        pytest.skip("harness NWORDS differs from the main.c:465-483 rule for this N")
    out = ctypes.create_string_buffer(4096)
    for op, key in ((0, "mul"), (1, "sqr"), (2, "add"), (3, "sub")):
        for a, b, want in zip(s["a"], s["b"], s[key]):
            orc.orc_l0_hex(c, op, a.encode(), b.encode(), out)
            assert out.value.decode() == want, (key, a, b)
    # fused add+sub of the reference equals separate add and sub
    assert s["asum"] == s["add"] and s["adiff"] == s["sub"]
    orc.orc_destroy(c)


def _n_of(case):
    return int(case["save_lines"][0].split("N=0x")[1].split(";")[0], 16)


FAST = [c for c in S1 if c["B1"] <= 100000]
SLOW = [c for c in S1 if c["B1"] > 100000]


def _check_stage1(orc, case, lanes):
    c = orc.orc_create(str(_n_of(case)).encode(), case["digitbits"])
    assert orc.orc_nwords(c) == case["nwords"] and orc.orc_maxbits(c) == case["maxbits"]
    line = ctypes.create_string_buffer(8192)
    fac = ctypes.create_string_buffer(2048)
    cnt = (ctypes.c_uint64 * 2)()
    facs = {}
    for k in lanes:
        want = case["save_lines"][k]
        sigma = int(want.split("SIGMA=")[1].split(";")[0])
        orc.orc_stage1_line(c, sigma, case["B1"], line, len(line), fac, len(fac), cnt)
        assert line.value.decode().rstrip("\n") == want
        assert (cnt[0], cnt[1]) == (case["ptadds"], case["ptdups"])
        if fac.value:
            facs[k] = int(fac.value)
    orc.orc_destroy(c)
    want_f = {}
    for l in case["results_lines"]:
        m = re.match(r"found (?:PRP|C)\d+ factor (\d+) in stage 1 .*vec (\d+), sigma", l)
        if m and int(m.group(2)) in lanes:
            want_f[int(m.group(2))] = int(m.group(1))
    assert facs == want_f


@pytest.mark.parametrize("case", FAST, ids=[c["name"] for c in FAST])
def test_oracle_stage1_lines(orc, case):
    lanes = range(len(case["save_lines"])) if case["B1"] <= 10000 else range(2)
    _check_stage1(orc, case, lanes)


@pytest.mark.parametrize("case", [c for c in SLOW if c["name"] in ("n415_b1_1000000", "K1")], ids=lambda c: c["name"])
def test_oracle_stage1_b1_1e6_one_lane(orc, case):
    _check_stage1(orc, case, [0])


def test_oracle_prac_choice_is_deterministic(orc):
    # spot values; the full chain is pinned by the save lines above
    assert orc.orc_prac_choice(3) in range(10)
    assert orc.orc_lucas_cost(7, 0.61803398874989485) > 0
