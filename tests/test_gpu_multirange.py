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
                                                  (200, 2500, 1250, 0),
                                                  # a PRIME range length (tools/soak_fuzz.py found the difference): 503 ends the
                                                  # first list — the lists hold both ends, GetPRIMESRange — and heads the second,
                                                  # where it is skipped; and a last range [2012, 2013) without any prime
                                                  (729, 2013, 503, 0)])
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


# ---- the command-line driver over several prime ranges AND several reference batches ----
# (the reference-made fixtures of tests/test_gpu_cli.py::test_cli_multirange_b1_above_1e8 are single batches: three
# minutes of reference time each.  Here PRIME_RANGE is shortened for the driver process — GECM_TEST_PRIME_RANGE — and the
# expected lines come from the oracle walked the same way; the ORDER expected is the reference's: it finishes a batch,
# all its ranges and checkpoints, before it starts the next, ecm.c:1151-1312.)
import subprocess
import tempfile

EXE = os.path.join(ROOT, "avx-ecm_amd", "avx-ecm")
S1 = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "stage1.json")))}


def _cli(args, env):
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run([EXE] + [str(a) for a in args], cwd=d, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stdout + p.stderr
        rd = lambda f: open(os.path.join(d, f)).read().splitlines() if os.path.exists(os.path.join(d, f)) else []
        return p.stdout, rd("save_b1.txt"), rd("checkpoint.txt"), [l for l in rd("ecm_results.txt") if l.strip()]


@pytest.mark.parametrize("threads,passes", [(1, {}), (2, {}), (1, {"GECM_PASS_CURVES": "8"}), (1, {"GECM_PASS_CURVES": "16"})])
def test_cli_checkpoints_of_several_batches_come_in_the_reference_s_order(threads, passes):
    c = S1["K1N_two_full_batches_b1_500"]                      # no small factors: nothing stops the run
    n = int(c["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
    b1, prange, sigma0, batches = 2500, 1000, 300, 3
    out, save, ckpt, res = _cli([c["N"], 8 * batches * threads, b1, threads, b1, sigma0], dict(passes, GECM_TEST_PRIME_RANGE=str(prange)))
    L = _oracle()
    o = L.orc_create(str(n).encode(), 52)
    want_ckpt, want_save = [], []
    for b in range(batches):
        sig = [sigma0 + 8 * b + i for i in range(8)]
        for r in (0, 1):                                        # ranges [0,1000) and [1000,2000) end below B1: checkpoints
            lines = [_oracle_line(L, o, s, b1, b1, prange, r + 1, 0)[0].rstrip("\n") for s in sig]
            want_ckpt += lines * threads                        # every thread of a batch: the same eight sigmas
        want_save += [_oracle_line(L, o, s, b1, b1, prange, 0, b1)[0].rstrip("\n") for s in sig] * threads
    L.orc_destroy(o)
    assert ckpt == want_ckpt
    assert save == want_save and res == []
    assert out.count("Saving checkpoint after p=997") >= 1 and "Saving checkpoint after p=1999" in out
    assert "Found 168 primes in range [0 : 1000]" in out and "Commencing Stage 1 @ prime 1009" in out


def test_cli_checkpoint_factor_stops_after_its_batch():
    """a modulus with small factors: the first batch finds one at the first checkpoint already; the reference reports
    it there (B1 = the range's last prime, ecm.c:1262-1291), goes on to the end of stage 1, and stops after the batch"""
    c = S1["n415_b1_1000"]
    b1, prange = 2500, 1000
    out, save, ckpt, res = _cli([c["N"], 24, b1, 1, b1, c["sigma0"]], {"GECM_TEST_PRIME_RANGE": str(prange)})
    assert len(save) == 8 and len(ckpt) == 16                  # batch 0 only: two checkpoints, one save
    assert [l.split("B1=")[1].split(";")[0] for l in ckpt] == ["997"] * 8 + ["1999"] * 8
    assert [int(l.split("SIGMA=")[1].split(";")[0]) for l in save] == list(range(c["sigma0"], c["sigma0"] + 8))
    labels = [l.split("in stage 1 (B1 = ")[1].split(")")[0] for l in res]
    assert labels == sorted(labels, key=lambda x: (0, 1, 2)[["997", "1999", "2500"].index(x)])
    assert set(labels) == {"997", "1999", "2500"} and all("curve " in l and "thread 0, vec " in l for l in res)
    # the factor lines of the last range are the ones a plain run at that B1 cannot give (other residues), but every
    # factor reported divides N
    n = int(save[0].split("N=0x")[1].split(";")[0], 16)
    assert all(n % int(l.split(" factor ")[1].split(" ")[0]) == 0 for l in res)


def test_ranges_with_the_special_form_multiply(short_ranges):
    """N | 2^k - 1 at a batch size that takes the F-form context: every range runs there (its own copy of the range's
    tape), the points come back modulo N between ranges, and the lines equal the generic path's and the oracle's"""
    import pyecm
    short_ranges(1500)
    n = (1 << 401) - 1
    b1 = 4000
    sig = list(range(1000, 1000 + 200))
    L = _oracle()
    o = L.orc_create(str(n).encode(), 52)
    out = {}
    for special in (True, False):
        eng = pyecm.Engine(n)
        eng.set_special_form(special)
        eng.set_lanes_per_curve(1)
        eng.build_curves(sig)
        per_range = []
        for r in range(pyecm.stage1_ranges(b1)):
            d = pyecm.describe_range(b1, b1, r)
            eng.stage1_range(b1, r)
            assert eng.special_form_used() == special
            per_range.append([eng.resume_line(k, d.last_prime) for k in (0, 63, 64, 199)])
        out[special] = per_range + [[eng.save_line(k) for k in (0, 63, 64, 199)]]
        eng.close()
    assert out[True] == out[False]
    for r in range(3):
        assert out[True][r][0] == _oracle_line(L, o, sig[0], b1, b1, 1500, r + 1, 0)[0]
    assert out[True][3][3] == _oracle_line(L, o, sig[199], b1, b1, 1500, 0, b1)[0]
    L.orc_destroy(o)
