"""Input preparation (gecm_prepare_input = main.c:393-527 of the reference): expression, Cunningham-form
detection, algebraic-factor removal.  The fixture holds what the reference itself printed for each
expression (tests/golden/make_golden.py --only inputs).  CPU only: no device is touched."""
import ctypes
import json
import os

import pytest

from conftest import GOLDEN, ROOT

FIX = json.load(open(os.path.join(GOLDEN, "inputs.json")))


class Info(ctypes.Structure):
    _fields_ = [("form", ctypes.c_int), ("k", ctypes.c_int), ("c", ctypes.c_uint64), ("nbits", ctypes.c_int),
                ("ref_special_reduction", ctypes.c_int)]


@pytest.fixture(scope="module")
def lib():
    L = ctypes.CDLL(os.path.join(ROOT, "avx-ecm_amd", "libgecm.so"))
    L.gecm_prepare_input.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                     ctypes.POINTER(Info), ctypes.c_char_p, ctypes.c_size_t]
    return L


def prepare(L, expr, digitbits=52):
    nd, log, inf = ctypes.create_string_buffer(4096), ctypes.create_string_buffer(1 << 16), Info()
    rc = L.gecm_prepare_input(expr.encode(), digitbits, nd, len(nd), ctypes.byref(inf), log, len(log))
    return rc, nd.value.decode(), log.value.decode().splitlines(), inf


@pytest.mark.parametrize("case", FIX["banner"], ids=[c["expr"][:24] for c in FIX["banner"]])
def test_prepared_input_and_printed_lines_equal_the_reference(lib, case):
    rc, n, lines, inf = prepare(lib, case["expr"])
    assert lines == case["lines"]
    if case["N"] is None:                       # the reference gave up on this input (exit(1) in find_primitive_factor)
        assert rc < 0
        return
    assert rc == 0 and n == case["N"] and inf.nbits == int(n).bit_length()
    assert bool(inf.ref_special_reduction) == case["special_reduction"]


def test_prepare_input_rejects_what_cannot_be_factored(lib):
    for bad in ("", "2^", "10", "1", "2^64", "7-9", "foo(3)"):
        assert prepare(lib, bad)[0] < 0
    assert prepare(lib, "15", digitbits=64)[0] < 0


def test_factor_size_label_is_gmp_sizeinbase(lib):
    """the reference labels factors with mpz_sizeinbase(f, 10): exact or one more (fixture: C13 for 657080389981)"""
    lib.gecm_sizeinbase10.argtypes = [ctypes.c_char_p]
    line = next(r for r in FIX["runs"] if r["name"] == "redc_phi105_phi210")["results_lines"][0]
    assert "found C13 factor 657080389981 " in line
    assert lib.gecm_sizeinbase10(b"657080389981") == 13
    for v, want in ((0, 1), (1, 1), (7, 1), (9, 2), (10, 2), (255, 3), (10 ** 12 - 1, 13), (2 ** 40, 13), (2 ** 39, 13), (2 ** 39 - 1, 12)):
        assert lib.gecm_sizeinbase10(str(v).encode()) == want, v
    assert lib.gecm_sizeinbase10(b"12x") < 0
