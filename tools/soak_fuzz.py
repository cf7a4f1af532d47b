#!/usr/bin/env python3
"""Differential soak on the GPU, outside the suite: random moduli (generic, 2^k -/+ 1, 2^k - c, with random small cofactors
removed), random B1 up to --b1max, random sigmas, ragged batches; every stage-1 kernel flavour (1, 2, 8, 32 lanes per
curve; generic and special-form multiply; the tape in one launch or cut into pieces; one prime range or several short ones)
must write the oracle's save lines.  Prints one line per case and a summary; exits 1 on the first difference.
usage: soak_fuzz.py [--seed S] [--minutes M] [--b1max B]"""
import argparse, ctypes, os, random, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd"))
import pyecm

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--minutes", type=float, default=8.0)
ap.add_argument("--b1max", type=int, default=20000)
a = ap.parse_args()
L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
L.orc_create.restype = ctypes.c_void_p
L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
L.orc_destroy.argtypes = [ctypes.c_void_p]
L.orc_stage1_ranges_line.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                     ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                                     ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]
L.orc_stage2.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                         ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
hook = pyecm.lib.gecm_plan_set_prime_range_for_tests
hook.argtypes = [ctypes.c_uint64]
hook.restype = None
rng = random.Random(a.seed)
t_end = time.time() + 60 * a.minutes
done = flavours = 0
while time.time() < t_end:
    kind = rng.randrange(4)
    if kind == 0:
        bits = rng.randrange(40, 1031)
        n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        name = "rand%d" % bits
    elif kind == 3:
        k = rng.randrange(200, 1025)
        cc = rng.getrandbits(rng.choice((3, 9, 20, 27, 28, 29, 40, 51))) | 1
        n = (1 << k) - cc
        name = "2^%d-%d" % (k, cc)
    else:
        k = rng.randrange(100, 1025)
        n = (1 << k) - 1 if kind == 1 else (1 << k) + 1
        name = "2^%d%s1" % (k, "-" if kind == 1 else "+")
    if kind:
        for p in (3, 5, 7, 11, 13, 17, 19, 23, 31, 127, 257):
            while n % p == 0 and n > p and rng.random() < 0.6:
                n //= p
    if n < 1000 or n % 2 == 0:
        continue
    b1 = rng.randrange(10, a.b1max)
    prange = rng.choice((0, 0, max(16, b1 // rng.randrange(2, 6))))
    chunk = rng.choice((0, 0, 64, 1000, 4096))
    digitbits = 32 if n.bit_length() >= 1000 else rng.choice((52, 52, 32))
    batch = rng.choice((1, 7, 9, 64, 65, 130))
    sig = [rng.randrange(6, 1 << rng.choice((10, 32, 63))) for _ in range(batch)]
    pick = sorted({0, batch // 2, batch - 1})
    o = L.orc_create(str(n).encode(), digitbits)
    buf = ctypes.create_string_buffer(16384)
    want = []
    for k in pick:
        L.orc_stage1_ranges_line(o, sig[k], b1, b1, prange if prange else 100000000, 0, b1, buf, len(buf), None, 0, None, None)
        want.append(buf.value.decode())
    L.orc_destroy(o)
    hook(prange)
    if chunk:
        os.environ["GECM_TAPE_CHUNK"] = str(chunk)
    else:
        os.environ.pop("GECM_TAPE_CHUNK", None)
    eng = pyecm.Engine(n, digitbits=digitbits)
    special_available = eng.special_form()[1] != 0
    tried = []
    for special in ((True, False) if special_available else (False,)):
        for lanes in ((1, 2) if special else (1, 2, 8, 32)):
            if lanes == 32 and eng.cfg.dev_limbs < 10 and batch > 64:
                pass
            eng.set_special_form(special)
            eng.set_lanes_per_curve(lanes)
            eng.build_curves(sig)
            try:
                eng.stage1(b1)
            except pyecm.GecmError as e:
                if "no 32-lane kernel" in str(e) or "no eight-lane kernel" in str(e):
                    continue
                raise
            got = [eng.save_line(k) for k in pick]
            tried.append("%s%d" % ("s" if special else "g", lanes))
            if got != want:
                print("DIFFERENCE: %s n=%d b1=%d prange=%d chunk=%d digitbits=%d batch=%d special=%s lanes=%d sigmas=%s" %
                      (name, n, b1, prange, chunk, digitbits, batch, special, lanes, [sig[k] for k in pick]), flush=True)
                sys.exit(1)
            flavours += 1
    # stage 2 now and then (one range of primes, the library's D and U) on a modulus WITHOUT small factors — a product of
    # Mersenne primes — so that no inversion fails and the accumulator is defined by the arithmetic alone: accumulator,
    # factor and counters against the oracle's.  (Where inversions fail, the reported factor is the gcd of the last failing
    # batch, and on moduli of many tiny primes that gcd depends on the addition chain — DESIGN.md §7; such lanes are pinned
    # by fixtures, tests/golden/degenerate.json among them.)
    s2 = ""
    if rng.random() < 0.3:
        eng.close()
        M = {e: (1 << e) - 1 for e in (61, 89, 107, 127, 521, 607)}
        n2 = rng.choice([M[127] * M[89], M[127] * M[107] * M[89] * M[61], M[521], M[521] * M[127], M[607] * M[127] * M[89],
                         M[607] * M[127] * M[107] * M[89] * M[61], M[521] * M[107] * M[89]])
        d2 = rng.choice((52, 32)) if n2.bit_length() < 1000 else 32
        b1s = rng.randrange(30, 5000)
        b2 = b1s + rng.randrange(500, 200000)
        hook(0)
        os.environ.pop("GECM_TAPE_CHUNK", None)
        eng = pyecm.Engine(n2, digitbits=d2)
        eng.build_curves(sig)
        eng.stage1(b1s)
        eng.stage2(b2)
        st = eng.stage2_stats()
        acc = eng.download_acc()
        o = L.orc_create(str(n2).encode(), d2)
        acch = ctypes.create_string_buffer(16384)
        fac = ctypes.create_string_buffer(4096)
        cnt = (ctypes.c_uint64 * 3)()
        for k in pick:
            L.orc_stage2(o, sig[k], b1s, b2, st.D, st.U, acch, fac, len(fac), cnt)
            f = eng.stage2_factor(k)
            wf = int(fac.value) if fac.value else None
            if (f[0] if f else None) != wf or int(acch.value, 16) != acc[k] or list(cnt) != [st.ptadds, st.numinv, st.paired]:
                print("STAGE-2 DIFFERENCE: n=%d b1=%d b2=%d D=%d U=%d digitbits=%d batch=%d sigma=%d: factor %s vs oracle %s, counters %s vs %s" %
                      (n2, b1s, b2, st.D, st.U, d2, batch, sig[k], f, wf, [st.ptadds, st.numinv, st.paired], list(cnt)), flush=True)
                sys.exit(1)
        L.orc_destroy(o)
        s2 = " stage2(%d bits, B1=%d, B2=%d, D=%d)" % (n2.bit_length(), b1s, b2, st.D)
    eng.close()
    done += 1
    print("%4d ok %-22s %4d bits b1=%-6d ranges=%-5s chunk=%-5d d%d batch=%-3d %s" % (done, name, n.bit_length(), b1, prange or "-", chunk, digitbits, batch, " ".join(tried) + s2), flush=True)
hook(0)
print("soak: %d moduli, %d kernel runs, all equal to the oracle (seed %d)" % (done, flavours, a.seed), flush=True)
