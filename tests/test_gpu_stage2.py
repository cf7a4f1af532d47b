"""GPU parity for stage 2 (ecm_stage2_init / ecm_stage2_pair, ecm.c:2201-2540) through the C ABI.

  * accumulator (stg2acc) bit-identical to the REFERENCE's own work->stg2acc (tests/golden/stage2_acc.json, taken
    from the reference at ecm.c:1489 by oracle/ref_tap.c) for the D, U the reference chose;
  * accumulator bit-identical to the oracle for other (N, sigma, B1, B2, D, U): the oracle's stage 2 is pinned
    to the same fixture and to the reference's result lines and counters by tests/test_oracle.py;
  * factors found on the reference's own stage-2 KATs (test.csh / test_t35.csh via
    tests/golden/stage1.json);
  * counters (point adds, inversions, pair multiplications) equal to the reference's printout.
"""
import ctypes
import json
import os
import re

import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

S1 = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "stage1.json")))}
K1N = int(S1["K1"]["save_lines"][0].split("N=0x")[1].split(";")[0], 16)


@pytest.fixture(scope="module")
def orc():
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
    L.orc_create.restype = ctypes.c_void_p
    L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.orc_destroy.argtypes = [ctypes.c_void_p]
    L.orc_stage2.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                             ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t,
                             ctypes.POINTER(ctypes.c_uint64)]
    return L


def _oracle(orc, n, sig, b1, b2, D, U, digitbits=52):
    c = orc.orc_create(str(n).encode(), digitbits)
    out = []
    for s in sig:
        acc = ctypes.create_string_buffer(4096)
        fac = ctypes.create_string_buffer(2048)
        cnt = (ctypes.c_uint64 * 3)()
        orc.orc_stage2(c, s, b1, b2, D, U, acc, fac, len(fac), cnt)
        out.append((int(acc.value, 16), int(fac.value) if fac.value else None, tuple(cnt)))
    orc.orc_destroy(c)
    return out


@pytest.mark.parametrize("b1,b2,D,U,digitbits", [(2000, 100000, 385, 4, 52), (2000, 60000, 385, 16, 52),
                                                  (5000, 300000, 2310, 2, 52), (300, 20000, 210, 3, 32)])
def test_stage2_accumulator_equals_oracle(orc, b1, b2, D, U, digitbits):
    import pyecm
    sig = list(range(100, 100 + 66))           # ragged: 2 wavefronts
    eng = pyecm.Engine(K1N, digitbits=digitbits)
    eng.build_curves(sig)
    eng.stage1(b1)
    eng.stage2(b2, D, U)
    acc = eng.download_acc()
    st = eng.stage2_stats()
    facs = [eng.stage2_factor(k) for k in range(len(sig))]
    eng.close()
    check = [0, 1, 63, 64, 65]
    want = _oracle(orc, K1N, [sig[k] for k in check], b1, b2, D, U, digitbits)
    for k, (wacc, wfac, wcnt) in zip(check, want):
        assert acc[k] == wacc, "sigma %d" % sig[k]
        assert (facs[k][0] if facs[k] else None) == wfac
        assert (st.ptadds, st.numinv, st.paired) == wcnt


S2ACC = json.load(open(os.path.join(GOLDEN, "stage2_acc.json")))


@pytest.mark.parametrize("case", S2ACC, ids=[c["name"] for c in S2ACC])
def test_stage2_accumulator_equals_reference_stg2acc(case):
    """every lane of the reference's stg2acc (ecm.c:1489; CROSS_PRODUCT_INV ecm.c:1857-1859), with the library
    left to choose D and U as the reference does (main.c:838-872; U = 16), and the reference's counters"""
    import pyecm
    eng = pyecm.Engine(int(case["N"]), digitbits=case["digitbits"])
    assert eng.cfg.nwords == case["nwords"]
    eng.build_curves([case["sigma0"] + k for k in range(case["curves"])])
    eng.stage1(case["B1"])
    eng.stage2(case["B2"])
    st = eng.stage2_stats()
    assert (st.D, st.U, st.L) == (case["D"], case["U"], case["L"])
    assert [st.ptadds, st.numinv, st.paired] == case["stage2_counts"]
    assert eng.download_acc() == [int(h, 16) for h in case["acc_hex"]]
    found = {int(re.search(r"vec (\d+),", l).group(1)): int(re.search(r"factor (\d+) in stage 2", l).group(1))
             for l in case["results_lines"] if "in stage 2" in l}
    for k in range(case["curves"]):
        f = eng.stage2_factor(k)
        assert (f[0] if f else None) == found.get(k), k
    eng.close()


def test_stage2_phase_api_matches_convenience(orc):
    """gecm_stage2_init + gecm_pair_primes + gecm_stage2_pair (the reference's own call sequence)"""
    import pyecm
    sig = list(range(500, 508))
    b1, b2 = 3000, 150000
    eng = pyecm.Engine(K1N)
    eng.build_curves(sig)
    eng.stage1(b1)
    eng.stage2_init(0, 0)
    st = eng.stage2_stats()
    assert (st.D, st.U, st.L) == (1155, 16, 32)          # main.c:848-851, U observed in the reference
    pm = pyecm.pair_primes(b1, b2, st.D, st.U)
    eng.stage2_pair(pm)
    acc = eng.download_acc()
    eng.close()
    want = _oracle(orc, K1N, sig[:2], b1, b2, 1155, 16)
    assert acc[0] == want[0][0] and acc[1] == want[1][0]


def test_stage2_kept_tapes_are_found_again_and_never_mistaken(orc):
    """The launch tape of a pair map is kept from batch to batch (gecm_stage2_pair: by a fingerprint of the map;
    gecm_stage2: with the kept map; gecm_stage2_pair_prepare: ahead of time).  Same map again, another map, another
    (D, U), and the convenience call in between: every accumulator is the oracle's."""
    import pyecm
    sig = list(range(900, 966))
    b1 = 2500
    eng = pyecm.Engine(K1N)

    def fresh():
        eng.build_curves(sig)
        eng.stage1(b1)

    def check(b2, D, U):
        acc = eng.download_acc()
        want = _oracle(orc, K1N, [sig[0], sig[65]], b1, b2, D, U)
        assert acc[0] == want[0][0] and acc[65] == want[1][0], (b2, D, U)

    fresh()
    pm_a = pyecm.pair_primes(b1, 120000, 1155, 16)
    eng.stage2_pair_prepare(pm_a, 1155, 16)               # before any stage-2 state exists on the device
    eng.stage2_init(1155, 16)
    eng.stage2_pair(pm_a)
    check(120000, 1155, 16)
    fresh()
    eng.stage2_init(1155, 16)
    eng.stage2_pair(pm_a)                                 # the same map: tape and device copy kept
    check(120000, 1155, 16)
    fresh()
    pm_b = pyecm.pair_primes(b1, 90000, 1155, 16)         # another map of the same (D, U)
    eng.stage2_init(1155, 16)
    eng.stage2_pair(pm_b)
    check(90000, 1155, 16)
    fresh()
    eng.stage2(120000, 385, 4)                            # the convenience call with another plan in between
    check(120000, 385, 4)
    fresh()
    eng.stage2(120000, 385, 4)                            # ... and again: its own kept map and tape
    check(120000, 385, 4)
    fresh()
    eng.stage2_init(1155, 16)
    eng.stage2_pair(pm_a)                                 # back to the first map after the plan changed twice
    check(120000, 1155, 16)
    eng.close()


def _kat(name, lanes=1):
    import pyecm
    case = S1[name]
    n = int(case["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
    sig = [int(l.split("SIGMA=")[1].split(";")[0]) for l in case["save_lines"]][:lanes]
    eng = pyecm.Engine(n)
    eng.build_curves(sig)
    eng.stage1(case["B1"])
    eng.stage2(case["B2"])
    st = eng.stage2_stats()
    facs = [eng.stage2_factor(k) for k in range(len(sig))]
    eng.close()
    want = {}
    for l in case["results_lines"]:
        m = re.match(r"found (PRP|C)\d+ factor (\d+) in stage 2 .*vec (\d+), sigma (\d+)", l)
        if m and int(m.group(3)) < lanes:
            want[int(m.group(3))] = (int(m.group(2)), m.group(1) == "PRP")
    got = {k: f for k, f in enumerate(facs) if f}
    assert got == want
    assert [st.ptadds, st.numinv, st.paired] == case["stage2_counts"]


def test_stage2_kat_t35():
    """test_t35.csh line 46: PRP31 factor in stage 2 at B1=1e6, B2=1e8"""
    _kat("T35_46")


def test_stage2_kat_k2_two_ranges():
    """test.csh:7 (K2): PRP42 factor in stage 2, B2=1.5e8 spans two prime ranges"""
    _kat("K2")


def test_stage2_when_stage1_already_found_the_factor():
    """random N with small factors: Z == 0 mod p after stage 1, so the batch inversion of stage 2
    meets a non-invertible product (ecm.c:1927-1939); the factor is reported from that gcd"""
    _kat("n415_b1_10000_b2_1e6", lanes=8)


def test_stage2_config1_lane_with_stage1_factor():
    """BASELINE configs[0] (fib(791)/13/677/216416017, 8 curves, B2 = 1e8): sigma 1007 finds its PRP21 in stage 1
    and the reference reports it again after stage 2; no other lane reports anything"""
    _kat("config1_fib791", lanes=8)


@pytest.mark.parametrize("args", [(200, 200, 60000, 210, 4), (130, 2000, 100000, 385, 16), (70, 5000, 300000, 2310, 3)])
def test_stage2_sub_sequences_give_the_plain_chain_s_accumulators(args):
    """small batches build the table and make the giant steps with K interleaved sub-sequences per curve
    (gecm_dev_s2_subseq; GECM_S2_SUBSEQ overrides): same points, same X/Z, same accumulator as the plain chain, for
    every K; B1 < D (amin = 0, where the reference's first giant steps are degenerate) falls back to the plain chain"""
    import pyecm
    batch, b1, b2, D, U = args
    sig = list(range(5000, 5000 + batch))
    accs = {}
    for K in (1, 2, 8, 32):
        os.environ["GECM_S2_SUBSEQ"] = str(K)
        try:
            eng = pyecm.Engine(K1N)
            eng.build_curves(sig)
            eng.stage1(b1)
            eng.stage2(b2, D, U)
            accs[K] = eng.download_acc()
            eng.close()
        finally:
            os.environ.pop("GECM_S2_SUBSEQ", None)
    assert accs[2] == accs[1] and accs[8] == accs[1] and accs[32] == accs[1]


def test_stage2_rejects_table_height_beyond_the_ring():
    import pyecm
    eng = pyecm.Engine(K1N)
    eng.build_curves([100, 101])
    eng.stage1(300)
    with pytest.raises(pyecm.GecmError):
        eng.stage2_init(210, 129)              # window of 4U giant steps + one chunk would not fit the ring
    eng.stage2_init(210, 128)
    eng.close()


def test_device_factor_scan_equals_per_curve_host_gcd():
    """gecm_scan_factors (device gcd of the whole batch) flags exactly the curves whose host
    gcd(Z, N) / gcd(acc, N) reports a factor (check_factor, ecm.c:2542-2557)"""
    import pyecm
    n = 1000003 * K1N                 # one 20-bit prime factor: some curves find it at B1=300, most do not
    sig = list(range(1000, 1000 + 130))
    eng = pyecm.Engine(n)
    eng.build_curves(sig)
    eng.stage1(300)
    host = [eng.stage1_factor(k) is not None for k in range(len(sig))]
    nf, first = eng.scan_factors(1)
    dev = [eng.curve_flag(1, k) for k in range(len(sig))]
    assert dev == host and nf == sum(host) and 0 < nf < len(sig)
    assert first == host.index(True)
    eng.stage2(3000, 0, 2)
    host2 = [eng.stage2_factor(k) is not None for k in range(len(sig))]
    nf2, first2 = eng.scan_factors(2)
    assert [eng.curve_flag(2, k) for k in range(len(sig))] == host2 and nf2 == sum(host2)
    eng.close()


@pytest.mark.parametrize("batch", [40000, 9000])
def test_sliced_pair_walk_equals_single_accumulator(batch):
    """A batch that cannot fill the chip walks each run of pairs in several slices with separate
    accumulators (3 slices for 40000 curves, 14 for 9000 on 256 CUs) merged at the end of the range; the
    merged accumulator must be the one a single running accumulator gives (GECM_S2_SLICES=1), bit for bit,
    and the oracle's on the lanes checked."""
    import ctypes
    import os
    import pyecm
    from conftest import ROOT
    n = K1N
    sig = list(range(5000, 5000 + batch))
    b1, b2, D, U = 200, 60000, 210, 4
    accs = []
    for env in (None, "1"):
        if env is None:
            os.environ.pop("GECM_S2_SLICES", None)
        else:
            os.environ["GECM_S2_SLICES"] = env
        try:
            eng = pyecm.Engine(n)
            eng.build_curves(sig)
            eng.stage1(b1)
            eng.stage2(b2, D, U)
            accs.append(eng.download_acc())
            eng.close()
        finally:
            os.environ.pop("GECM_S2_SLICES", None)
    assert accs[0] == accs[1]
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
    L.orc_create.restype = ctypes.c_void_p
    L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.orc_stage2.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                             ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t,
                             ctypes.POINTER(ctypes.c_uint64)]
    c = L.orc_create(str(n).encode(), 52)
    acch = ctypes.create_string_buffer(8192)
    for k in (0, 63, 64, batch // 2, batch - 1):
        L.orc_stage2(c, sig[k], b1, b2, D, U, acch, None, 0, None)
        assert int(acch.value, 16) == accs[0][k]


def test_stage2_on_a_modulus_of_many_small_primes_reports_the_last_failing_batch():
    """tests/golden/degenerate.json (a reference run found through tools/soak_fuzz.py): a modulus of many small primes.
    Every stage-2 batch inversion of these curves fails, with different gcds, and the curves' tiny orders modulo those
    primes make differential additions degenerate — where, depends on the addition chain.  The plain chain (one
    sub-sequence: the reference's own chain) reports the reference's factors on all eight lanes; the sub-sequence chains
    of a small batch report proper factors of N that DIVIDE them.  Save lines, stage-1 factors and counters
    are the reference's in every configuration."""
    import pyecm
    case = json.load(open(os.path.join(GOLDEN, "degenerate.json")))[0]
    n = int(case["save_lines"][0].split("N=0x")[1].split(";")[0], 16)
    want1 = {int(re.search(r"vec (\d+),", l).group(1)): int(re.search(r"factor (\d+) in", l).group(1)) for l in case["results_lines"] if "in stage 1" in l}
    want2 = {int(re.search(r"vec (\d+),", l).group(1)): int(re.search(r"factor (\d+) in", l).group(1)) for l in case["results_lines"] if "in stage 2" in l}
    for env in ({}, {"GECM_S2_SUBSEQ": "1"}, {"GECM_S2_SUBSEQ": "4", "GECM_S2_SLICES": "1"}):
        for k in ("GECM_S2_SUBSEQ", "GECM_S2_SLICES"):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            eng = pyecm.Engine(n)
            eng.build_curves([case["sigma0"] + k for k in range(8)])
            eng.stage1(case["B1"])
            assert [l.rstrip("\n") for l in eng.save_lines()] == case["save_lines"]
            assert {k: eng.stage1_factor(k)[0] for k in range(8)} == want1
            eng.stage2(case["B2"])
            st = eng.stage2_stats()
            assert [st.ptadds, st.numinv, st.paired] == case["stage2_counts"]
            eng.scan_factors(2)
            for k in range(8):
                f = eng.stage2_factor(k)
                assert f and eng.curve_flag(2, k)
                assert 1 < f[0] < n and n % f[0] == 0 and want2[k] % f[0] == 0, (env, k, f[0], want2[k])
                if env.get("GECM_S2_SUBSEQ") == "1":
                    assert f[0] == want2[k], (k, f[0], want2[k])       # the reference's own chain: the reference's factor
            eng.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
