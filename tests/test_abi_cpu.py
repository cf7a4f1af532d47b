"""CPU tests (no GPU): the C-ABI library loads, exports every symbol include/gecm.h declares, fails
loudly without a device, and its host logic (tape compiler, sieve, PAIR, curve-setup arithmetic)
agrees with the reference-derived fixtures and with the oracle."""
import ctypes
import json
import os
import re
import subprocess

import pytest

from conftest import GOLDEN, ROOT

LIB = os.path.join(ROOT, "avx-ecm_amd", "libgecm.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "avx-ecm_amd"), "-j8"])
    return ctypes.CDLL(LIB)


def test_header_symbols_all_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "gecm.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(gecm_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 25
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    import pyecm
    assert set(pyecm.EXPORTS) == names


def test_no_device_is_a_loud_error_not_a_fallback(lib):
    import pyecm
    if pyecm.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pyecm.GecmError):
        pyecm.Engine((1 << 127) - 1)
    # and bad arguments are rejected before the device is touched
    h = ctypes.c_void_p()
    lib.gecm_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
    assert lib.gecm_create(ctypes.byref(h), 0, b"1000", 52) == -2      # even N
    assert lib.gecm_create(ctypes.byref(h), 0, b"1001", 48) == -2      # bad limb format


class Tape(ctypes.Structure):
    _fields_ = [("ops", ctypes.POINTER(ctypes.c_uint8)), ("len", ctypes.c_size_t), ("ptadds", ctypes.c_uint64),
                ("ptdups", ctypes.c_uint64), ("prac_calls", ctypes.c_uint64), ("last_prime", ctypes.c_uint64),
                ("rule_count", ctypes.c_uint64 * 4), ("swaps", ctypes.c_uint64)]


def test_tape_counts_match_reference_counters(lib):
    """the reference prints ptadds/ptdups after stage 1 (ecm.c:1849-1850); fixtures hold them"""
    cases = json.load(open(os.path.join(GOLDEN, "stage1.json")))
    seen = {}
    for c in cases:
        seen[c["B1"]] = (c["ptadds"], c["ptdups"])
    for b1, (adds, dups) in sorted(seen.items()):
        t = Tape()
        assert lib.gecm_tape_build_stage1(ctypes.byref(t), ctypes.c_uint64(b1)) == 0
        assert (t.ptadds, t.ptdups) == (adds, dups), b1
        assert t.len == t.prac_calls * 2 + sum(t.rule_count) + sum(1 for k in range(1, 64) if 2 ** k < b1)
        lib.gecm_tape_free(ctypes.byref(t))
    # op mix at B1=1e6
    t = Tape()
    lib.gecm_tape_build_stage1(ctypes.byref(t), ctypes.c_uint64(1000000))
    assert t.prac_calls == 78715 and list(t.rule_count) == [1762907, 103154, 35999, 42] and t.swaps == 1369236
    assert t.last_prime == 999983
    lib.gecm_tape_free(ctypes.byref(t))


def test_prac_multiplier_choice_equals_oracle(lib):
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
    orc.orc_prac_choice.argtypes = [ctypes.c_uint64]
    lib.gecm_prac_best_multiplier.argtypes = [ctypes.c_uint64]
    orc.orc_lucas_cost.restype = ctypes.c_double
    orc.orc_lucas_cost.argtypes = [ctypes.c_uint64, ctypes.c_double]
    lib.gecm_lucas_cost.restype = ctypes.c_double
    lib.gecm_lucas_cost.argtypes = [ctypes.c_uint64, ctypes.c_double]
    for c in list(range(3, 3000, 2)) + [999983, 99999989, 2 ** 31 - 1, 1000003]:
        assert lib.gecm_prac_best_multiplier(c) == orc.orc_prac_choice(c)
        assert lib.gecm_lucas_cost(c, 0.61803398874989485) == orc.orc_lucas_cost(c, 0.61803398874989485)


def test_sieve(lib):
    lib.gecm_primes_range.restype = ctypes.POINTER(ctypes.c_uint64)
    lib.gecm_primes_range.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(ctypes.c_size_t)]
    n = ctypes.c_size_t()
    p = lib.gecm_primes_range(0, 100, ctypes.byref(n))
    assert [p[i] for i in range(n.value)] == [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67,
                                              71, 73, 79, 83, 89, 97]
    p = lib.gecm_primes_range(0, 10 ** 7, ctypes.byref(n))
    assert n.value == 664579
    p = lib.gecm_primes_range(10 ** 8 - 100, 10 ** 8 + 100, ctypes.byref(n))
    assert [p[i] for i in range(n.value)] == [99999931, 99999941, 99999959, 99999971, 99999989, 100000007,
                                              100000037, 100000039, 100000049, 100000073, 100000081]
    p = lib.gecm_primes_range(0, 10 ** 8, ctypes.byref(n))
    assert n.value == 5761455                       # "cached 5761455 primes < 99999989" (main.c:583)


def test_pair_map_equals_oracle_and_reference_counts(lib):
    import pyecm
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
    orc.orc_pair.restype = ctypes.c_uint32
    PU = ctypes.POINTER(ctypes.c_uint32)
    orc.orc_pair.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(PU),
                             ctypes.POINTER(PU), PU, PU, PU]
    for b1, b2, D, U in [(1000, 100000, 385, 4), (10000, 1000000, 2310, 16), (500, 50000, 210, 2), (60, 6000, 30, 1)]:
        pm = pyecm.pair_primes(b1, b2, D, U)
        v, u, am, pr, nq = PU(), PU(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        n = orc.orc_pair(b1, b2, D, U, ctypes.byref(v), ctypes.byref(u), ctypes.byref(am), ctypes.byref(pr), ctypes.byref(nq))
        assert (pm.steps, pm.amin, pm.pairs, pm.primes) == (n, am.value, pr.value, nq.value)
        assert all(pm.pairmap_v[i] == v[i] and pm.pairmap_u[i] == u[i] for i in range(n))
        # coverage property (the reference's own `testcoverage` switch, ecm.c:2883-2900): every prime
        # in [B1, B2) is hit by exactly the pair entries, walking the window as the device does
        amin, covered = pm.amin, set()
        for i in range(n):
            if pm.pairmap_v[i] == 0 and pm.pairmap_u[i] == 0:
                amin += U
                continue
            pa = pm.pairmap_v[i] - amin
            assert 0 <= pa < 4 * U
            centre = (2 * amin + pa) * D
            covered.add(centre - pm.pairmap_u[i])
            covered.add(centre + pm.pairmap_u[i])
        sieve = [True] * (b2 + 1)
        for i in range(2, int(b2 ** 0.5) + 1):
            if sieve[i]:
                for j in range(i * i, b2 + 1, i):
                    sieve[j] = False
        primes = [p for p in range(max(b1, 2), b2) if sieve[p]]
        assert all(p in covered for p in primes)
        lib.gecm_pairmap_release(ctypes.byref(pm))
    # the reference's printout for config 1: "3008627 pairs found from 5682957 primes", amin = 216
    pm = pyecm.pair_primes(1000000, 100000000, 2310, 16)
    assert (pm.pairs, pm.primes, pm.amin) == (3008627, 5682957, 216)
    lib.gecm_pairmap_release(ctypes.byref(pm))


def test_mpl_bigint_kit(lib):
    import random
    MAXL = 136

    class M(ctypes.Structure):
        _fields_ = [("n", ctypes.c_int), ("d", ctypes.c_uint32 * MAXL)]

    def to(v):
        m = M()
        assert lib.mpl_set_str(ctypes.byref(m), hex(v).encode()) == 0
        return m

    def fr(m):
        return sum(m.d[i] << (32 * i) for i in range(m.n))

    rng = random.Random(7)
    for _ in range(300):
        a = rng.getrandbits(rng.choice([1, 32, 33, 415, 831, 1023, 2000]))
        b = rng.getrandbits(rng.choice([5, 32, 64, 65, 415, 1023])) | 1
        A, B, q, r, g, iv = to(a), to(b), M(), M(), M(), M()
        lib.mpl_divrem(ctypes.byref(q), ctypes.byref(r), ctypes.byref(A), ctypes.byref(B))
        assert (fr(q), fr(r)) == divmod(a, b)
        lib.mpl_gcd(ctypes.byref(g), ctypes.byref(A), ctypes.byref(B))
        import math
        assert fr(g) == math.gcd(a, b)
        ok = lib.mpl_invmod(ctypes.byref(iv), ctypes.byref(A), ctypes.byref(B))
        if b > 1:
            assert bool(ok) == (math.gcd(a, b) == 1)
            if ok:
                assert fr(iv) == pow(a, -1, b)
        buf = ctypes.create_string_buffer(MAXL * 10 + 2)
        lib.mpl_get_dec(buf, ctypes.byref(A))
        assert buf.value.decode() == str(a)
        lib.mpl_get_hex(buf, ctypes.byref(A))
        assert buf.value.decode() == "%x" % a


def test_cli_without_gpu_fails_loudly():
    import pyecm
    if pyecm.device_count() > 0:
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "avx-ecm_amd", "avx-ecm")
    p = subprocess.run([exe, "1000003*1000033", "8", "1000", "1", "1000", "7"], capture_output=True, text=True)
    assert p.returncode == 2 and "no HIP device" in p.stderr
    p = subprocess.run([exe, "2^64", "8", "1000"], capture_output=True, text=True)       # even input
    assert p.returncode == 1 and "odd integer" in p.stdout


def test_input_expression_evaluator(lib):
    """calc_lite (host/calc_lite.c): the command line's first argument is an expression, as in the
    reference (calc.c, README.md:30); config 1 of BASELINE.json uses one."""
    import math
    MAXL = 136

    class M(ctypes.Structure):
        _fields_ = [("n", ctypes.c_int), ("d", ctypes.c_uint32 * MAXL)]

    def ev(s):
        m = M()
        return None if lib.calc_lite(ctypes.byref(m), s.encode()) else sum(m.d[i] << (32 * i) for i in range(m.n))

    def fib(n):
        a, b = 0, 1
        for _ in range(n):
            a, b = b, a + b
        return a

    n1 = fib(791) // 13 // 677 // 216416017
    assert ev("fib(791)/13/677/216416017") == n1 and n1.bit_length() == 508
    assert ev("2^127-1") == 2 ** 127 - 1 and ev("2^3^2") == 512
    assert ev("20!+1") == math.factorial(20) + 1 and ev("31#") == 200560490130
    assert ev("luc(10)") == 123 and ev("0x10+1") == 17 and ev(" 7 * ( 3 + 4 ) ") == 49
    assert ev("(2^64+13)*3 % 1000") == (2 ** 64 + 13) * 3 % 1000
    assert ev("gcd(2^64-1, 2^32-1)") == 2 ** 32 - 1 and ev("sqrt(10^40+12345)") == math.isqrt(10 ** 40 + 12345)
    assert ev("modinv(3, 1000003)") == pow(3, -1, 1000003) and ev("modexp(2, 1000, 10^9+7)") == pow(2, 1000, 10 ** 9 + 7)
    assert ev("1<<100") == 1 << 100 and ev("(2^200+5)>>3") == (2 ** 200 + 5) >> 3
    for bad in ("3-5", "1/0", "foo(3)", "2^100000", "(1+2", "", "modinv(2,4)"):
        assert ev(bad) is None


def test_stage2_device_tape_follows_the_reference_s_batches(lib):
    """gecm_s2_tape_build (host/gecm_pair.c): the giant steps the tape asks for are exactly the reference's E = 2L + 2U
    per window shift (ecm.c:2425, 2499), no chunk exceeds 512, the LAST chunk is the reference's last inversion batch
    (its 2U newest steps) and is flagged single-chain, every pair lands on the giant step and table entry its (v,u)
    names, and a table too tall for the ring, or an entry outside the window, is refused"""
    import pyecm

    class Plan(ctypes.Structure):
        _fields_ = [("D", ctypes.c_uint32), ("U", ctypes.c_uint32), ("L", ctypes.c_uint32), ("R", ctypes.c_uint32),
                    ("umax", ctypes.c_uint32), ("map", ctypes.POINTER(ctypes.c_uint32)), ("npb", ctypes.c_uint32),
                    ("keep", ctypes.POINTER(ctypes.c_uint32)), ("keep_words", ctypes.c_size_t)]

    class Tape(ctypes.Structure):
        _fields_ = [("words", ctypes.POINTER(ctypes.c_uint32)), ("nwords", ctypes.c_size_t), ("adds", ctypes.c_uint64),
                    ("inv", ctypes.c_uint64), ("paired", ctypes.c_uint64), ("devinv", ctypes.c_uint64),
                    ("amin_last", ctypes.c_uint32)]

    PU = ctypes.POINTER(ctypes.c_uint32)
    lib.gecm_s2_plan_init.argtypes = [ctypes.POINTER(Plan), ctypes.c_uint32, ctypes.c_uint32]
    lib.gecm_s2_plan_free.argtypes = [ctypes.POINTER(Plan)]
    lib.gecm_s2_tape_build.argtypes = [ctypes.POINTER(Tape), ctypes.POINTER(Plan), ctypes.c_uint32, PU, PU, ctypes.c_uint32,
                                       ctypes.c_uint32, ctypes.c_uint32, PU]
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    GEN, chunk, ring = 0xffffffff, 512, 1024
    for b1, b2, D, U in [(10000, 3000000, 2310, 16), (2000, 100000, 385, 4), (300, 20000, 210, 3), (100, 9000, 210, 2)]:
        plan, tape, bad = Plan(), Tape(), ctypes.c_uint32()
        assert lib.gecm_s2_plan_init(ctypes.byref(plan), D, U) == 0
        pm = pyecm.pair_primes(b1, b2, D, U)
        assert lib.gecm_s2_tape_build(ctypes.byref(tape), ctypes.byref(plan), pm.steps, pm.pairmap_v, pm.pairmap_u, pm.amin,
                                      chunk, ring, ctypes.byref(bad)) == 0
        w = [tape.words[i] for i in range(tape.nwords)]
        shifts = sum(1 for i in range(pm.steps) if pm.pairmap_v[i] == 0 and pm.pairmap_u[i] == 0)
        E = 4 * U + 2 * U * shifts
        gens = [(w[i + 1] & 0x7fffffff, w[i + 1] >> 31) for i in range(0, len(w), 2) if w[i] == GEN]
        assert sum(n for n, f in gens) == E and all(0 < n <= chunk for n, f in gens)
        assert gens[-1] == ((2 * U if shifts else 4 * U), 1)                 # the reference's last batch, one chain
        assert all(f == (1 if pm.amin == 0 else 0) for n, f in gens[:-1])    # amin = 0: the plain chain throughout
        assert (tape.paired, tape.inv, tape.amin_last) == (pm.pairs, 2 + shifts, pm.amin + U * shifts)
        # replay: every pair must find its giant step generated and still inside the ring
        want, amin = [], pm.amin
        for i in range(pm.steps):
            if pm.pairmap_v[i] == 0 and pm.pairmap_u[i] == 0:
                amin += U
            else:
                want.append(((2 * amin - 2 * pm.amin + pm.pairmap_v[i] - amin) % ring, plan.map[pm.pairmap_u[i]]))
        got, generated = [], 0
        for i in range(0, len(w), 2):
            if w[i] == GEN:
                generated += w[i + 1] & 0x7fffffff
            else:
                got.append((w[i], w[i + 1]))
        assert sorted(got) == sorted(want) and generated == E
        libc.free(tape.words)
        # one entry outside the window: refused with its index
        v = (ctypes.c_uint32 * 3)(pm.pairmap_v[0], pm.amin + 4 * U, pm.pairmap_v[0])
        u = (ctypes.c_uint32 * 3)(pm.pairmap_u[0], pm.pairmap_u[0], pm.pairmap_u[0])
        assert lib.gecm_s2_tape_build(ctypes.byref(tape), ctypes.byref(plan), 3, v, u, pm.amin, chunk, ring, ctypes.byref(bad)) == -2
        assert bad.value == 1
        lib.gecm_pairmap_release(ctypes.byref(pm))
        lib.gecm_s2_plan_free(ctypes.byref(plan))
    # a table taller than the ring allows (4U + chunk > ring)
    plan = Plan()
    assert lib.gecm_s2_plan_init(ctypes.byref(plan), 210, 129) == 0
    z = (ctypes.c_uint32 * 4)(0, 0, 0, 0)
    tape, bad = Tape(), ctypes.c_uint32()
    assert lib.gecm_s2_tape_build(ctypes.byref(tape), ctypes.byref(plan), 4, z, z, 5, chunk, ring, ctypes.byref(bad)) == -2
    lib.gecm_s2_plan_free(ctypes.byref(plan))


def test_pair_walk_kernels_with_hand_placed_loads_keep_their_rows_in_registers():
    """gecm_stage2.hpp requests table rows with inline-asm loads up to GECM_S2_ASYNC_MAXNL limbs; that is only sound while
    the compiler never moves a pending row to scratch.  Read the code objects of the built library's kernel objects: the
    pair-walk kernels of those limb counts use no scratch and spill no register."""
    import glob
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    hdr = open(os.path.join(ROOT, "avx-ecm_amd", "csrc", "gecm_stage2.hpp")).read()
    maxnl = int(re.search(r"#define GECM_S2_ASYNC_MAXNL (\d+)", hdr).group(1))
    objs = glob.glob(os.path.join(ROOT, "avx-ecm_amd", "build", "gecm_kernels_*_p2.o"))
    if not objs:
        pytest.skip("no kernel objects (library built elsewhere)")
    seen = 0
    for o in objs:
        nl = int(re.search(r"_(\d+)_p2", o).group(1))
        if nl > maxnl:
            continue
        for name, vgpr, agpr, sgpr, scratch, spills in kernel_resources.kernels(o):
            if "k_s2_pairs" in name:
                seen += 1
                assert scratch == 0 and spills == 0, (nl, name, scratch, spills)
                assert vgpr <= 256
    assert seen >= 10


# ---- stage 1 above one prime range (ecm.c:1209-1312) ----
MULTI = json.load(open(os.path.join(GOLDEN, "multirange.json"))) if os.path.exists(os.path.join(GOLDEN, "multirange.json")) else []


@pytest.mark.skipif(not MULTI, reason="tests/golden/multirange.json not generated")
def test_range_tapes_and_range_descriptions_match_the_reference_s_stdout(lib):
    """per prime range: the counters after each ecm_stage1 call ("Stage 1 completed at prime p with a point-adds and
    d point-doubles", cumulative: the doublings run again in every range), the sieved interval and its prime count
    ("Found n primes in range [lo : hi]"), P_MIN, and whether a checkpoint follows — all from the reference's own
    output for B1 = 1.1e8 and for B1 = 1e8 (one range, checkpoint written all the same)"""
    import re
    import pyecm
    lib.gecm_tape_build_stage1_range.argtypes = [ctypes.POINTER(Tape), ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int]
    for c in MULTI:
        out = c["stdout_lines"]
        found = [tuple(map(int, m.groups())) for m in (re.match(r"Found (\d+) primes in range \[(\d+) : (\d+)\]", l) for l in out) if m]
        pmin = [int(m.group(1)) for m in (re.match(r"Commencing Stage 1 @ prime (\d+)", l) for l in out) if m]
        done = [tuple(map(int, m.groups())) for m in
                (re.match(r"Stage 1 completed at prime (\d+) with (\d+) point-adds and (\d+) point-doubles", l) for l in out) if m]
        ckpt = [int(m.group(1)) for m in (re.match(r"Saving checkpoint after p=(\d+)", l) for l in out) if m]
        nr = pyecm.stage1_ranges(c["B1"])
        assert nr == len(found) == len(pmin) == len(done)
        adds = dups = 0
        for r in range(nr):
            d = pyecm.describe_range(c["B1"], c["B2"], r)
            assert (d.nprimes, d.lo, d.hi) == found[r] and d.first_prime == pmin[r] and d.last_prime == done[r][0]
            assert bool(d.checkpoint) == (d.last_prime in ckpt)
            t = Tape()
            assert lib.gecm_tape_build_stage1_range(ctypes.byref(t), c["B1"], r, 8) == 0
            adds += t.ptadds
            dups += t.ptdups
            assert (t.last_prime, adds, dups) == done[r], (c["name"], r)
            lib.gecm_tape_free(ctypes.byref(t))


def test_range_tape_is_independent_of_the_worker_threads(lib):
    lib.gecm_tape_build_stage1_range.argtypes = [ctypes.POINTER(Tape), ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int]
    lib.gecm_plan_set_prime_range_for_tests.argtypes = [ctypes.c_uint64]
    lib.gecm_plan_set_prime_range_for_tests.restype = None
    try:
        lib.gecm_plan_set_prime_range_for_tests(300000)
        for r in (0, 1, 3):
            tapes = []
            for nt in (1, 3, 8):
                t = Tape()
                assert lib.gecm_tape_build_stage1_range(ctypes.byref(t), 1000000, r, nt) == 0
                tapes.append((bytes(t.ops[:t.len]), t.ptadds, t.ptdups, t.prac_calls, t.last_prime, list(t.rule_count), t.swaps))
                lib.gecm_tape_free(ctypes.byref(t))
            assert tapes[0] == tapes[1] == tapes[2]
        t = Tape()
        assert lib.gecm_tape_build_stage1_range(ctypes.byref(t), 1000000, 4, 1) == -2     # 4 ranges of 3e5 below 1e6: 0..3
        assert lib.gecm_tape_build_stage1(ctypes.byref(t), 1000000) == -2                # several ranges: no single tape
    finally:
        lib.gecm_plan_set_prime_range_for_tests(0)
    # one range = the whole of stage 1, as before
    a, b = Tape(), Tape()
    assert lib.gecm_tape_build_stage1(ctypes.byref(a), 100000) == 0
    assert lib.gecm_tape_build_stage1_range(ctypes.byref(b), 100000, 0, 4) == 0
    assert bytes(a.ops[:a.len]) == bytes(b.ops[:b.len]) and a.last_prime == b.last_prime == 99991
    lib.gecm_tape_free(ctypes.byref(a))
    lib.gecm_tape_free(ctypes.byref(b))


def test_the_library_was_built_from_this_tree(lib):
    """gecm_version() carries the hash of the sources every object inside libgecm.so was compiled from (kernel objects,
    32-lane kernels, device layer, host C; avx-ecm_amd/Makefile).  They must be the tree's: a stale object, a library
    left over from another commit or a DEV build fails here — and in smoke() on the GPU box — instead of producing
    numbers for code that is not the code."""
    import __graft_entry__ as ge
    lib.gecm_version.restype = ctypes.c_char_p
    v = lib.gecm_version().decode()
    assert "MIXED" not in v and "unset" not in v, v
    ge.check_manifest(v)
    with pytest.raises(RuntimeError):
        ge.check_manifest(v.replace("K:", "K:0"))
    # the command-line driver prints the same string
    exe = os.path.join(ROOT, "avx-ecm_amd", "avx-ecm")
    assert os.path.exists(exe)
