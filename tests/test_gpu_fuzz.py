"""Seeded differential fuzz on the GPU: random moduli (random odd N of 64..1030 bits, Cunningham forms
2^k -/+ 1 and pseudo-Mersenne forms 2^k - c with random small cofactors removed), random B1 and sigma; every kernel flavour of stage 1 — one
two, eight and 32 lanes per curve, generic and special-form multiply — must write the save lines of the oracle
(oracle/ecm_oracle.c, itself pinned to the reference's outputs)."""
import ctypes
import os
import random

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libecm_oracle.so"))
    L.orc_create.restype = ctypes.c_void_p
    L.orc_create.argtypes = [ctypes.c_char_p, ctypes.c_int]
    L.orc_destroy.argtypes = [ctypes.c_void_p]
    L.orc_stage1_line.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_size_t,
                                  ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
    return L


def _cases():
    rng = random.Random(20261004)
    out = []
    for i in range(48):
        kind = i % 4
        if kind == 3:
            # pseudo-Mersenne (main.c:432-441): N | 2^k - c with c odd below one reference limb
            k = rng.randrange(300, 1025)
            cc = rng.getrandbits(rng.choice((5, 20, 27, 28, 40, 50))) | 1
            n = (1 << k) - cc
            for p in (3, 5, 7, 11, 13, 17, 19, 23):
                while n % p == 0 and rng.random() < 0.7:
                    n //= p
            name = "2^%d-%d" % (k, cc)
        elif kind == 0:
            bits = rng.randrange(64, 1031)
            n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
            name = "rand%d" % bits
        else:
            k = rng.randrange(170, 1025)
            n = (1 << k) - 1 if kind == 1 else (1 << k) + 1
            for p in (3, 5, 7, 11, 13, 17, 31, 127, 257):       # divide out some small factors, as users do
                while n % p == 0 and rng.random() < 0.7:
                    n //= p
            name = "2^%d%s1" % (k, "-" if kind == 1 else "+")
        out.append((name, n, rng.randrange(10, 1500), rng.randrange(6, 1 << 62), rng.choice((52, 32)) if n.bit_length() < 1000 else 32))
    return out


CASES = _cases()


@pytest.mark.parametrize("name,n,b1,sigma0,digitbits", CASES, ids=["%02d_%s" % (i, c[0]) for i, c in enumerate(CASES)])
def test_all_stage1_kernel_flavours_write_the_oracles_lines(orc, name, n, b1, sigma0, digitbits):
    import pyecm
    sig = [sigma0 + 7 * j for j in range(9)]
    c = orc.orc_create(str(n).encode(), digitbits)
    line = ctypes.create_string_buffer(16384)
    want = []
    for s in sig:
        orc.orc_stage1_line(c, s, b1, line, len(line), None, 0, None)
        want.append(line.value.decode())
    orc.orc_destroy(c)
    eng = pyecm.Engine(n, digitbits=digitbits)
    special_available = eng.special_form()[1] != 0
    for special in ((True, False) if special_available else (False,)):
        for lanes in ((1, 2) if special else (1, 2, 8, 32)):
            eng.set_special_form(special)
            eng.set_lanes_per_curve(lanes)
            eng.build_curves(sig)
            eng.stage1(b1)
            assert eng.lanes_per_curve() == lanes and eng.special_form_used() == special
            assert eng.save_lines() == want, (name, special, lanes)
    eng.close()


def _s2_cases():
    """Stage 2: moduli without small factors (products of Mersenne primes, so that no inversion fails and the
    accumulator is defined by the reference's arithmetic alone), random B1 < B2, every wheel, table sizes U = 1 .. 16, the
    pair walk in one slice or in the library's choice of slices, K sub-sequences or the plain chain."""
    rng = random.Random(20261005)
    M = {e: (1 << e) - 1 for e in (61, 89, 107, 127, 521, 607)}
    mods = [M[127] * M[89], M[127] * M[107] * M[89] * M[61], M[521], M[521] * M[127], M[607] * M[127] * M[89],
            M[607] * M[127] * M[107] * M[89] * M[61], M[521] * M[107] * M[89]]
    out = []
    for i in range(28):
        n = mods[i % len(mods)]
        b1 = rng.randrange(50, 3000)
        D = rng.choice((210, 385, 2310))
        U = rng.randrange(1, 17)
        b2 = b1 + rng.randrange(1000, 40 * D * U + 20000)
        env = {}
        if rng.random() < 0.5:
            env["GECM_S2_SLICES"] = "1"
        if rng.random() < 0.3:
            env["GECM_S2_SUBSEQ"] = str(rng.choice((1, 2, 4)))
        out.append((n, b1, b2, D, U, rng.randrange(6, 1 << 62), rng.choice((1, 64, 65, 130)), env))
    return out


S2_CASES = _s2_cases()


@pytest.mark.parametrize("n,b1,b2,D,U,sigma0,batch,env", S2_CASES,
                         ids=["%02d_%dbit_B2_%d_D%d_U%d%s" % (i, c[0].bit_length(), c[2], c[3], c[4], "_" + "_".join(sorted(c[7])) if c[7] else "")
                              for i, c in enumerate(S2_CASES)])
def test_stage2_accumulators_on_random_parameters(orc, n, b1, b2, D, U, sigma0, batch, env, monkeypatch):
    import pyecm
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    orc.orc_stage2.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                               ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t,
                               ctypes.POINTER(ctypes.c_uint64)]
    sig = [sigma0 + k for k in range(batch)]
    eng = pyecm.Engine(n)
    eng.build_curves(sig)
    eng.stage1(b1)
    eng.stage2(b2, D, U)
    acc = eng.download_acc()
    eng.close()
    c = orc.orc_create(str(n).encode(), 52)
    acch = ctypes.create_string_buffer(8192)
    for k in sorted({0, batch // 2, batch - 1}):
        orc.orc_stage2(c, sig[k], b1, b2, D, U, acch, None, 0, None)
        assert int(acch.value, 16) == acc[k], (k, b1, b2, D, U, env)
    orc.orc_destroy(c)
