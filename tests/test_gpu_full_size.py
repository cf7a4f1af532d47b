"""GPU parity at BASELINE.json's full sizes (the fixtures hold 8 or 16 curves per case; these run the whole batch).

  * configs[1] and configs[2]: 4096 curves, B1 = 1e6, 415- and 831-bit N, through the layout the library picks (32
    lanes per curve) and through one other layout: one sha256 over all 4096 save lines; lanes 0, 63, 64, 4095
    against the oracle; the first 8 lanes against the save_b1.txt lines the reference wrote (stage1.json);
  * the per-GPU slice of configs[3]: 4096 curves of the test_t35.csh modulus, B1 = 1e6, stage 2 to B2 = 1e8, the
    reference's KAT sigma (test_t35.csh line 46) in the middle of the batch: the factor is found at that index, the
    counters are the reference's 79,886 / 1,341 / 3,008,627, the accumulator of that lane equals the 8-curve run's.
"""
import ctypes
import hashlib
import json
import os
import re

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

S1 = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "stage1.json")))}


def _oracle_lines(n, sigmas, b1):
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
    L.orc_create.restype = ctypes.c_void_p
    L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.orc_destroy.argtypes = [ctypes.c_void_p]
    L.orc_stage1_line.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t,
                                  ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
    c = L.orc_create(str(n).encode(), 52)
    buf = ctypes.create_string_buffer(16384)
    out = []
    for s in sigmas:
        L.orc_stage1_line(c, s, b1, buf, len(buf), None, 0, None)
        out.append(buf.value.decode())
    L.orc_destroy(c)
    return out


@pytest.mark.parametrize("name,other", [("n415_b1_1000000", 8), ("n831_b1_1000000", 8)])
def test_config_4096_curves_b1_1e6(name, other):
    import pyecm
    case = S1[name]
    n = int(case["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
    b1, curves = 1000000, 4096
    sig = list(range(1000, 1000 + curves))
    eng = pyecm.Engine(n, digitbits=52)
    sha, lines = {}, None
    for lanes in (0, other):
        eng.set_lanes_per_curve(lanes)
        eng.build_curves(sig)
        eng.stage1(b1)
        used = eng.lanes_per_curve()
        got = eng.save_lines()
        sha[used] = hashlib.sha256("".join(got).encode()).hexdigest()
        if lanes == 0:
            assert used == 32                       # what bench.py's headline runs
            lines = got
            assert eng.stage1_stats().ptadds == 1980817 and eng.stage1_stats().ptdups == 217929
    eng.close()
    assert len(sha) == 2 and len(set(sha.values())) == 1, sha
    assert [l.rstrip("\n") for l in lines[:8]] == case["save_lines"]             # the reference's own lines
    check = [0, 63, 64, 4095]
    assert [lines[k] for k in check] == _oracle_lines(n, [sig[k] for k in check], b1)


def test_config4_4096_curves_1023_bits_32_bit_boundary_b1_1e5():
    """BASELINE configs[4]: the DIGITBITS = 32 boundary (NWORDS = 32, MAXBITS = 1024), 1023-bit N, B1 = 1e5, the whole
    4096-curve batch: one sha256 over all save lines through the layout the library picks (32 lanes per curve, three
    limbs per lane, operand limbs by DPP — two per 64-bit move) and through eight lanes per curve; the first 16 lanes against the lines the
    reference's 32-bit build wrote; lanes 0, 63, 64, 4095 against the oracle in the 32-bit format"""
    import pyecm
    case = S1["n1023_d32_b1_100000"]
    n = int(case["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
    b1, curves = 100000, 4096
    sig = list(range(1000, 1000 + curves))
    eng = pyecm.Engine(n, digitbits=32)
    assert (eng.cfg.nwords, eng.cfg.maxbits, eng.cfg.dev_limbs) == (32, 1024, 37)
    sha, lines = {}, None
    for lanes in (0, 8):
        eng.set_lanes_per_curve(lanes)
        eng.build_curves(sig)
        eng.stage1(b1)
        used = eng.lanes_per_curve()
        got = eng.save_lines()
        sha[used] = hashlib.sha256("".join(got).encode()).hexdigest()
        if lanes == 0:
            assert used == 32 and eng.last_kernel_name() == "k_stage1_row<3, 38, false>"
            lines = got
            assert (eng.stage1_stats().ptadds, eng.stage1_stats().ptdups) == (case["ptadds"], case["ptdups"]) == (195448, 23269)
    eng.close()
    assert len(sha) == 2 and len(set(sha.values())) == 1, sha
    assert [l.rstrip("\n") for l in lines[:16]] == case["save_lines"]
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
    L.orc_create.restype = ctypes.c_void_p
    L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.orc_destroy.argtypes = [ctypes.c_void_p]
    L.orc_stage1_line.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t,
                                  ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
    c = L.orc_create(str(n).encode(), 32)
    buf = ctypes.create_string_buffer(8192)
    for k in (0, 63, 64, 4095):
        L.orc_stage1_line(c, sig[k], b1, buf, len(buf), None, 0, None)
        assert buf.value.decode() == lines[k], k
    L.orc_destroy(c)


def test_config3_slice_stage2_4096_curves_b2_1e8():
    import pyecm
    case = S1["T35_46"]
    n = int(case["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
    kat = int(case["save_lines"][0].split("SIGMA=")[1].split(";")[0])
    want = int(re.match(r"found PRP\d+ factor (\d+) in stage 2", case["results_lines"][0]).group(1))
    curves, at = 4096, 2049
    sig = list(range(5000, 5000 + curves))
    sig[at] = kat
    eng = pyecm.Engine(n)
    eng.build_curves(sig)
    eng.stage1(case["B1"])
    assert eng.lanes_per_curve() == 32
    assert eng.save_line(at).rstrip("\n") == case["save_lines"][0]
    eng.stage2(case["B2"])
    st = eng.stage2_stats()
    assert [st.ptadds, st.numinv, st.paired] == case["stage2_counts"] == [79886, 1341, 3008627]
    nf, first = eng.scan_factors(2)
    assert eng.curve_flag(2, at) and eng.stage2_factor(at) == (want, True)
    assert nf >= 1 and first <= at
    acc_big = eng.download_acc()[at]
    # the same curve in an 8-curve batch (the reference's own shape): same accumulator
    eng.build_curves([kat] + list(range(9000, 9007)))
    eng.stage1(case["B1"])
    eng.stage2(case["B2"])
    assert eng.download_acc()[0] == acc_big
    assert eng.stage2_factor(0) == (want, True)
    eng.close()


S2ACC = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "stage2_acc.json")))}


@pytest.mark.parametrize("name", ["T35_46_b1_1e6_b2_1e8", "K1N_b1_1e6_b2_1e8"])
def test_config3_slice_accumulators_equal_the_reference_s_stg2acc(name):
    """BASELINE configs[3]'s per-GPU slice — 4096 curves, B1 = 1e6, B2 = 1e8, D = 2310, U = 16 — with the eight curves
    of a reference run (tests/golden/stage2_acc.json: work->stg2acc as the reference holds it at ecm.c:1489, tapped by
    oracle/ref_tap.c) placed at the ends and the middle of the batch, wavefront boundaries included: the accumulator
    of every one of them is the reference's, bit for bit, in the layouts a batch of this size runs in (32 lanes per
    curve in stage 1; 32 sub-sequences and 64 pair-walk slices per curve in stage 2)."""
    import pyecm
    case = S2ACC[name]
    n = int(case["N"])
    at = [0, 63, 64, 2049, 4095, 1, 2048, 4032]
    sig = list(range(700000, 700000 + 4096))
    for k, lane in enumerate(at):
        sig[lane] = case["sigma0"] + k
    eng = pyecm.Engine(n)
    assert eng.cfg.nwords == case["nwords"]
    eng.build_curves(sig)
    eng.stage1(case["B1"])
    assert eng.lanes_per_curve() == 32
    eng.stage2(case["B2"])
    st = eng.stage2_stats()
    assert (st.D, st.U, st.L) == (case["D"], case["U"], case["L"]) == (2310, 16, 32)
    assert [st.ptadds, st.numinv, st.paired] == case["stage2_counts"] == [79886, 1341, 3008627]
    acc = eng.download_acc()
    assert [acc[lane] for lane in at] == [int(h, 16) for h in case["acc_hex"]]
    found = {int(re.search(r"vec (\d+),", l).group(1)): int(re.search(r"factor (\d+) in stage 2", l).group(1))
             for l in case["results_lines"] if "in stage 2" in l}
    eng.scan_factors(2)
    for k, lane in enumerate(at):
        f = eng.stage2_factor(lane)
        assert (f[0] if f else None) == found.get(k), (k, lane)
        assert eng.curve_flag(2, lane) == (k in found)
    eng.close()
