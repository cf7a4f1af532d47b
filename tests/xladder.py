"""Independent check value for stage 1: [k]P by a plain x-only binary Montgomery ladder in Python integers
(no PRAC, no limbs, no Montgomery form).  Used by tests/golden/make_golden.py to tell on which lanes the
reference's special-reduction path is right, and by the GPU test of Cunningham-type inputs."""


def _primes(n):
    s = bytearray([1]) * (n + 1)
    s[0] = s[1] = 0
    for i in range(2, int(n ** 0.5) + 1):
        if s[i]:
            s[i * i::i] = bytearray(len(s[i * i::i]))
    return [i for i in range(n + 1) if s[i]]


def true_stage1_point(n, sigma, b1):
    """[k]P on the Suyama curve of sigma by a plain x-only binary ladder (independent of PRAC and of
    any limb arithmetic): the projective point every correct stage 1 must be proportional to."""
    u, v = (sigma * sigma - 5) % n, 4 * sigma % n
    x = pow(u, 3, n) * pow(pow(v, 3, n), -1, n) % n
    a24 = pow(v - u, 3, n) * (3 * u + v) % n * pow(16 * pow(u, 3, n) * v % n, -1, n) % n
    k = 1
    for p in _primes(b1):
        if p >= b1:
            break
        q = p
        while q * p < b1:
            q *= p
        k *= q

    def dbl(P):
        X, Z = P
        s, d = (X + Z) ** 2 % n, (X - Z) ** 2 % n
        t = (s - d) % n
        return s * d % n, t * (d + a24 * t) % n

    def add(P, Q):
        (X1, Z1), (X2, Z2) = P, Q
        a, b = (X1 - Z1) * (X2 + Z2) % n, (X1 + Z1) * (X2 - Z2) % n
        return (a + b) ** 2 % n, x * (a - b) ** 2 % n

    r0, r1 = (1, 0), (x, 1)
    for bit in bin(k)[2:]:
        if bit == "1":
            r0, r1 = add(r0, r1), dbl(r1)
        else:
            r0, r1 = dbl(r0), add(r0, r1)
    return r0
