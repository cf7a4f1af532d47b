"""GPU debug helper: step-by-step comparison of the device path with a pure-Python restatement."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "avx-ecm_amd"))
import pyecm


def suyama(n, sigma):
    u = sigma * sigma - 5
    v = 4 * sigma
    x3 = pow(u, 3, n); z3 = pow(v, 3, n)
    num = pow(v - u, 3, n) * ((3 * u + v) % n) % n
    den = 16 * x3 * v % n
    from math import gcd
    s = num * (pow(den, -1, n) if gcd(den, n) == 1 else 16 * x3) % n
    x = x3 * (pow(z3, -1, n) if gcd(z3, n) == 1 else num) % n
    return x, 1, s


def dup(n, x, z, s):
    V = (x - z) ** 2 % n; U = (x + z) ** 2 % n
    X = U * V % n; w = (U - V) % n; t = (w * s + V) % n
    return X, t * w % n


def add(n, p1, p2, pd):
    (x1, z1), (x2, z2), (xd, zd) = p1, p2, pd
    U = (x1 - z1) * (x2 + z2) % n; V = (x1 + z1) * (x2 - z2) % n
    return zd * (U + V) ** 2 % n, xd * (U - V) ** 2 % n


n = random.Random(415).getrandbits(415) | (1 << 414) | 1
eng = pyecm.Engine(n)
print("cfg", eng.cfg.nwords, eng.cfg.maxbits, eng.cfg.dev_limbs, eng.device_name())
sig = list(range(1000, 1008))
R = 1 << eng.cfg.maxbits
rc = eng.build_curves(sig)
print("build rc", rc)
X, Z = eng.download_points()
exp = [suyama(n, s) for s in sig]
print("upload/download mont ok:", X == [e[0] * R % n for e in exp], Z == [R % n] * 8)
x, z = eng.download_points_plain()
print("plain ok:", x == [e[0] for e in exp], z == [1] * 8)
for b1 in (2, 3, 4, 5, 6, 8, 10):
    eng.build_curves(sig)
    eng.stage1(b1)
    x, z = eng.download_points_plain()
    st = eng.stage1_stats()
    # python: emulate via simple ladder equivalence? only compare x/z ratio: [k]P with k = prod prime powers < b1
    k = 1
    q = 2
    while q < b1: k *= 2; q *= 2
    for p in (3, 5, 7):
        if p < b1:
            c = 1
            while True:
                k *= p; c *= p
                if c * p >= b1: break
    ok = []
    for (x0, z0, s), xx, zz in zip(exp, x, z):
        # Montgomery ladder in python for [k]P
        def ladder(k):
            if k == 1: return (x0, z0)
            p1 = (x0, z0); p2 = dup(n, x0, z0, s)
            for bit in bin(k)[3:]:
                if bit == '1':
                    p1 = add(n, p2, p1, (x0, z0)); p2 = dup(n, p2[0], p2[1], s)
                else:
                    p2 = add(n, p1, p2, (x0, z0)); p1 = dup(n, p1[0], p1[1], s)
            return p1
        ex, ez = ladder(k)
        ok.append((xx * ez - ex * zz) % n == 0 and zz != 0)
    print("B1=%d k=%d tape_len=%d adds=%d dups=%d ratio-ok=%s z0=%x" % (b1, k, st.tape_len, st.ptadds, st.ptdups, all(ok), z[0] & 0xffffffff), "ms=%.3f" % eng.last_kernel_ms())
