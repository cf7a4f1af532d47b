"""GPU parity for stage 1 above one prime range (B1 > 1e8; ecm.c:1209-1312): gecm_stage1_range is one ecm_stage1 call
of the reference's loop, with the 2-power doublings repeated and the first prime of every range skipped.

  * the path itself, cheaply: PRIME_RANGE shortened through the library's test hook, every range's checkpoint line
    and the final save line against the oracle walked the same way, in every lane layout, with the tape cut into
    several launches;
  * the real thing: tests/golden/multirange.json — checkpoint.txt and save_b1.txt the REFERENCE wrote for a 204-bit N
    at B1 = 1.1e8 (tests/golden/make_golden.py) — reproduced byte for byte through the command-line driver
    (tests/test_gpu_cli.py::test_cli_multirange_*)."""
import ctypes
import json
import os

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _oracle():
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
    L.orc_create.restype = ctypes.c_void_p
    L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.orc_destroy.argtypes = [ctypes.c_void_p]
    L.orc_stage1_ranges_line.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                         ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                         ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
    return L


def _oracle_line(L, c, sigma, b1, b2, prange, stop_after, b1_field):
    line = ctypes.create_string_buffer(8192)
    cnt = (ctypes.c_uint64 * 3)()
    ck = ctypes.c_int(0)
    L.orc_stage1_ranges_line(c, sigma, b1, b2, prange, stop_after, b1_field, line, len(line), None, 0, cnt, ctypes.byref(ck))
    return line.value.decode(), list(cnt), ck.value


@pytest.fixture
def short_ranges():
    import pyecm
    hook = pyecm.lib.gecm_plan_set_prime_range_for_tests
    hook.argtypes = [ctypes.c_uint64]
    hook.restype = None
    yield hook
    hook(0)
    os.environ.pop("GECM_TAPE_CHUNK", None)


@pytest.mark.parametrize("lanes", [1, 2, 8, 32])
@pytest.mark.parametrize("bits,b1,prange,chunk", [(415, 5000, 2000, 0), (415, 4001, 1000, 64), (831, 3000, 1024, 256),
                                                  (200, 2500, 1250, 0)])
def test_ranges_against_the_oracle(short_ranges, bits, b1, prange, chunk, lanes):
    import random
    import pyecm
    short_ranges(prange)
    if chunk:
        os.environ["GECM_TAPE_CHUNK"] = str(chunk)
    n = random.Random(bits).getrandbits(bits) | (1 << (bits - 1)) | 1
    sig = list(range(1000, 1000 + 70))
    nr = pyecm.stage1_ranges(b1)
    assert nr == -(-b1 // prange) and nr >= 2
    eng = pyecm.Engine(n)
    if lanes == 32 and eng.cfg.dev_limbs < 10:
        lanes = 8
    eng.set_lanes_per_curve(lanes)
    eng.build_curves(sig)
    L = _oracle()
    c = L.orc_create(str(n).encode(), 52)
    tot = [0, 0]
    for r in range(nr):
        d = pyecm.describe_range(b1, b1, r)
        eng.stage1_range(b1, r)
        st = eng.stage1_stats()
        for k in (0, 1, 63, 64, 69):
            want, cnt, ck = _oracle_line(L, c, sig[k], b1, b1, prange, r + 1, 0)
            assert eng.resume_line(k, d.last_prime) == want, (r, k)
            assert (st.ptadds, st.ptdups, st.last_prime) == tuple(cnt), (r, cnt)
            assert bool(d.checkpoint) == bool(ck)
    # the whole loop in one call gives the same points, and the save line carries B1
    lines = [eng.save_line(k) for k in (0, 1, 63, 64, 69)]
    eng.build_curves(sig)
    eng.stage1(b1)
    assert [eng.save_line(k) for k in (0, 1, 63, 64, 69)] == lines
    for i, k in enumerate((0, 1, 63, 64, 69)):
        assert lines[i] == _oracle_line(L, c, sig[k], b1, b1, prange, 0, b1)[0]
    # it is NOT the product of all prime powers below B1: the reference's ranges repeat the doublings and skip a prime
    short_ranges(0)
    eng.build_curves(sig)
    eng.stage1(b1)
    assert eng.save_line(0) != lines[0]
    L.orc_destroy(c)
    eng.close()


def test_cut_tape_is_the_same_stage1(short_ranges):
    """one range, the tape cut into many launches (how a 200 MB tape of a 1e8 range runs)"""
    import random
    import pyecm
    n = random.Random(623).getrandbits(623) | (1 << 622) | 1
    sig = list(range(2000, 2000 + 130))
    out = {}
    for chunk in (0, 4, 1000):
        if chunk:
            os.environ["GECM_TAPE_CHUNK"] = str(chunk)
        for lanes in (1, 2, 8, 32):
            eng = pyecm.Engine(n)
            eng.set_lanes_per_curve(lanes)
            eng.build_curves(sig)
            eng.stage1(3000)
            out[(chunk, lanes)] = eng.save_lines()
            eng.close()
    assert len({tuple(v) for v in out.values()}) == 1
